"""N > 1 path on CPU: world-size-2 gloo run of the sharding + pipelined gather logic bench.py uses
(swift_png_amd/distributed.py): a rank decodes its shard group by group, each group's rasters travel
to rank 0 as a batch of point-to-point transfers while the next group decodes.  The CPU oracle stands in
for the HIP path (no GPU here); rank 0 checks global order and completeness."""
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


def _worker(rank, world, port, total, q, weak=False, groups=3):
    sys.path.insert(0, os.path.dirname(__file__))
    sys.path.insert(0, os.path.dirname(os.path.dirname(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from swift_png_amd.distributed import gather_pipelined, shard
    w, h = 24, 10
    S = w * h * 4
    lo, hi = shard(total, world, rank)
    # strong scaling (bench.py's default at N > 1, BASELINE configs[2]): a rank decodes only its shard;
    # weak: every rank decodes the whole batch and contributes its share of the result
    first = 0 if weak else lo
    count = total if weak else hi - lo
    local = torch.zeros(max(count, 1) * S, dtype=torch.uint8)
    off = lo if weak else 0

    def decode_group(glo, ghi):              # the CPU oracle stands in for the HIP path here (no GPU)
        for j in range(off + glo, off + ghi):
            img = _make_image(first + j, w, h)
            rows = ph.orc_filter(img.reshape(-1), w, h, 8, 4, False)
            png = ph.Png(w, h, 8, 6, False, False, zlib.compress(rows, 6))
            st, storage, _ = ph.orc_decode(png)
            assert st == 0
            local[j * S:(j + 1) * S] = torch.from_numpy(storage.copy())

    gathered = torch.zeros(total * S, dtype=torch.uint8) if rank == 0 else None
    gather_pipelined(dist, decode_group, local, gathered, S, total, world, rank, groups, weak_offset=off)
    if rank == 0:
        ok = all(bool((gathered[i * S:(i + 1) * S].numpy() == _make_image(i, w, h).reshape(-1)).all())
                 for i in range(total))
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


def test_group_bounds_cover_a_shard():
    from swift_png_amd.distributed import group_bounds
    for n in (0, 1, 5, 128):
        for groups in (1, 3, 4, 200):
            spans = [group_bounds(n, groups, g) for g in range(groups)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_bench_spawns_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher must start N ranks itself (the driver runs it plainly)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert "torch.distributed.run" in cmd and "--nproc-per-node=8" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "8", "--steps", "2"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
