"""N > 1 path on CPU: world-size-2 gloo run of the sharding + gather logic bench.py uses
(swift_png_amd/distributed.py).  Each rank "decodes" its shard with the CPU oracle (the HIP path
needs a GPU), rank 0 gathers and checks global order and completeness."""
import os
import socket
import sys
import zlib

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pnghelp as ph


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_image(i, w=24, h=10):
    rng = np.random.default_rng(i)
    img = rng.integers(0, 256, (h, w * 4), dtype=np.uint8)
    return img


def _worker(rank, world, port, total, q, weak=False):
    sys.path.insert(0, os.path.dirname(__file__))
    sys.path.insert(0, os.path.dirname(os.path.dirname(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from swift_png_amd.distributed import gather_decoded, shard
    w, h = 24, 10
    S = w * h * 4
    lo, hi = shard(total, world, rank)
    local = []
    # weak scaling (bench.py's default): every rank decodes the whole batch and contributes its
    # shard of the result; strong: it only decodes its shard
    for i in range(0 if weak else lo, total if weak else hi):
        img = _make_image(i, w, h)
        rows = ph.orc_filter(img.reshape(-1), w, h, 8, 4, False)
        png = ph.Png(w, h, 8, 6, False, False, zlib.compress(rows, 6))
        st, storage, _ = ph.orc_decode(png)
        assert st == 0
        local.append(torch.from_numpy(storage.copy()))
    local = torch.cat(local) if local else torch.empty(0, dtype=torch.uint8)
    if weak:
        local = local[lo * S:hi * S]
    out = gather_decoded(local, S, total, world, rank)
    if rank == 0:
        ok = True
        for r in range(world):
            rlo, rhi = shard(total, world, r)
            for j, i in enumerate(range(rlo, rhi)):
                ok &= bool((out[r][j * S:(j + 1) * S].numpy() == _make_image(i, w, h).reshape(-1)).all())
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total,weak", [(7, False), (8, False), (8, True)])
def test_shard_and_gather_world2(total, weak):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q, weak)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True


def test_shard_covers_everything():
    from swift_png_amd.distributed import shard
    for total in (0, 1, 7, 1024):
        for world in (1, 2, 3, 8):
            spans = [shard(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert shard(1024, 8, 3) == (384, 512)
