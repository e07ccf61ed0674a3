"""Sharding of a batch of independent images over ranks, and the one exchange step of the path:
moving the decoded rasters to rank 0 (RCCL over xGMI on GPUs; gloo in the CPU tests).

Images are independent units (one zlib stream + one raster each), so ranks never
talk while decoding.  A rank decodes its shard in a few groups; as soon as a group is decoded its rasters
leave for rank 0 as one batch of point-to-point transfers (`exchange_plan` -> dist.batch_isend_irecv),
which the communication stream carries while the next group decodes.  xGMI is point-to-point: every
peer -> root transfer rides its own link, so the seven inbound slabs of a group arrive concurrently."""
from __future__ import annotations


def shard(total: int, world: int, rank: int):
    """Contiguous block of ceil(total/world) images per rank: -> (lo, hi)."""
    per = -(-total // world)
    lo = min(total, rank * per)
    return lo, min(total, lo + per)


def group_bounds(n: int, groups: int, g: int):
    """The g-th of `groups` consecutive pieces of a shard of n images: -> (lo, hi), local indices."""
    per = -(-n // max(1, groups)) if n else 0
    lo = min(n, g * per)
    return lo, min(n, lo + per)


def exchange_plan(dist, local, gathered, image_bytes: int, total: int, world: int, rank: int,
                  glo: int, ghi: int, groups: int, g: int, dst: int = 0, weak_offset: int = 0):
    """The transfers of group g.  `local`: this rank's rasters (flat uint8, image after image; the images
    of its shard start at slot `weak_offset`).  `gathered` (on dst): the whole batch in global order.
    A peer sends its group slab; dst copies its own slab and posts one receive per peer, for the slab that
    peer's group g covers.  -> list of dist.P2POp (empty when there is nothing to move)."""
    S = image_bytes
    ops = []
    if rank != dst:
        if ghi > glo:
            ops.append(dist.P2POp(dist.isend, local[(weak_offset + glo) * S:(weak_offset + ghi) * S], dst))
        return ops
    lo, hi = shard(total, world, rank)
    if ghi > glo:
        gathered[(lo + glo) * S:(lo + ghi) * S].copy_(local[(weak_offset + glo) * S:(weak_offset + ghi) * S])
    for r in range(world):
        if r == dst:
            continue
        rlo, rhi = shard(total, world, r)
        plo, phi = group_bounds(rhi - rlo, groups, g)
        if phi > plo:
            ops.append(dist.P2POp(dist.irecv, gathered[(rlo + plo) * S:(rlo + phi) * S], r))
    return ops


def gather_pipelined(dist, decode_group, local, gathered, image_bytes: int, total: int, world: int, rank: int,
                     groups: int, dst: int = 0, weak_offset: int = 0):
    """decode_group(lo, hi) for every group of this rank's shard, each followed by its transfers; waits
    for all of them at the end.  (What bench.py's step does; used by the CPU tests.)"""
    lo, hi = shard(total, world, rank)
    pending = []
    for g in range(groups):
        glo, ghi = group_bounds(hi - lo, groups, g)
        if ghi > glo:
            decode_group(glo, ghi)
        ops = exchange_plan(dist, local, gathered, image_bytes, total, world, rank, glo, ghi, groups, g, dst,
                            weak_offset)
        if ops:
            pending += dist.batch_isend_irecv(ops)
    for w in pending:
        w.wait()
