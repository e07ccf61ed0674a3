"""Sharding of a batch of independent images over ranks, and the one exchange step of the path:
gathering the decoded rasters to rank 0 (RCCL over xGMI on GPUs; gloo in the CPU tests).

Images are independent units (one zlib stream + one raster each), so ranks
never talk while decoding."""
from __future__ import annotations


def shard(total: int, world: int, rank: int):
    """Contiguous block of ceil(total/world) images per rank: -> (lo, hi)."""
    per = -(-total // world)
    lo = min(total, rank * per)
    return lo, min(total, lo + per)


def gather_decoded(local, image_bytes: int, total: int, world: int, rank: int, out=None, dst: int = 0):
    """Gathers every rank's decoded rasters (a flat uint8 tensor holding its shard, image after
    image) to `dst`.  Returns, on `dst`, a list of per-rank tensors covering images in global order
    (rank r's tensor holds images shard(total, world, r)); None elsewhere.  Shards are padded to the
    common size ceil(total/world) so that one collective moves everything."""
    import torch
    import torch.distributed as dist
    per = -(-total // world)
    lo, hi = shard(total, world, rank)
    want = per * image_bytes
    send = local if local.numel() == want else torch.cat([local, local.new_zeros(want - local.numel())])
    if rank == dst:
        out = out if out is not None else [torch.empty(want, dtype=local.dtype, device=local.device)
                                           for _ in range(world)]
        dist.gather(send, out, dst=dst)
        return out
    dist.gather(send, None, dst=dst)
    return None
