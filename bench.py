#!/usr/bin/env python3
"""bench.py -- the north-star measurement: batched PNG decode (inflate + unfilter) of 4K RGBA8 images.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the decode hot path (spng_decode_batch: inflate -> unfilter, results left
in HBM) over the whole batch.  Workload = BASELINE.json configs[1]: 1024 synthetic 4096x4096 RGBA8
PNG streams, mixed filters chosen by the reference's own heuristic, DEFLATE level 6.  At N > 1 the
images are independent units and every GPU decodes its own 1024 (weak scaling: the global batch is
1024 x N, no communication while decoding); the one exchange step of the path, gathering decoded
rasters to rank 0 over RCCL, moves each rank's 1024/N-image shard, i.e. the 1024-image result of
configs[2].  `--scaling strong` runs the fixed 1024-image batch sharded 1024/N per GPU instead (a
stream is a serial chain that takes ~1.5 s however idle the GPU is, so that variant cannot speed
up by more than the occupancy effect).  Compressed inputs are resident in HBM before the timed
region; nothing is copied to the host inside it.  The CPU oracle is used only for the
`cpu_baseline` leg.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

W = H = 4096
DEPTH, CHANNELS = 8, 4
MPIX = W * H / 1e6
HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def build_inputs(spng, session, unique: int, threads: int, encoder: str):
    """-> (list of original rasters as numpy, list of zlib streams).  Product path only: synthetic
    rasters -> GPU filter-select (spng_filter, the reference heuristic) -> level-6 DEFLATE, either
    zlib on the host cores (default, seconds) or the device deflater (swift-png's own level-6
    bitstream: a dynamic block every <= 2047 tokens; about a minute)."""
    from swift_png_amd import synth
    with ThreadPoolExecutor(threads) as pool:
        images = list(pool.map(lambda s: synth.image(s, W, H, CHANNELS, DEPTH), range(unique)))
        rows = [session.filter(img.tobytes(), W, H, DEPTH, CHANNELS, False) for img in images]
        if encoder == "swiftpng":
            outs, res = session.deflate_batch([session.to_device(r) for r in rows], 6)
            assert all(r.status == 0 for r in res)
            streams = [bytes(o[:r.written].cpu().numpy()) for o, r in zip(outs, res)]
        else:
            streams = list(pool.map(lambda r: zlib.compress(r, 6), rows))
    for r, z in zip(rows, streams):
        assert zlib.decompress(z) == r
    return images, rows, streams


def cpu_baseline(streams, images, cores: int, sample: int):
    """Times the CPU oracle (restatement of swift-png's CPU path: inflate + defilter + assign) on
    `sample` images, one image per thread.  Test infrastructure used as a reported baseline only."""
    sys.path.insert(0, str(ROOT / "tests"))
    import numpy as np
    import pnghelp as ph
    lib = ph.oracle()
    work = [(streams[i % len(streams)], i % len(streams)) for i in range(sample)]

    def one(job):
        z, k = job
        src = np.frombuffer(z, dtype=np.uint8)
        storage = np.empty(W * H * 4, dtype=np.uint8)
        aux = (ctypes.c_uint64 * 2)()
        st = lib.orc_decode(ph._ptr(src), len(z), 0, W, H, DEPTH, CHANNELS, 0, ph._ptr(storage), aux)
        return st == 0 and bytes(storage[:4096]) == images[k].reshape(-1)[:4096].tobytes()

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as pool:
        ok = list(pool.map(one, work))
    dt = time.perf_counter() - t0
    assert all(ok)
    return {"value": round(sample * MPIX / dt, 1), "unit": "MPixels/s", "cores": cores, "kind": "port",
            "sample": f"{sample} of the same 4096x4096 RGBA8 level-6 streams, one image per thread, "
                      f"oracle inflate+defilter+assign, {dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--images", type=int, default=1024, help="batch size (BASELINE: 1024)")
    ap.add_argument("--unique", type=int, default=32, help="distinct images; slot i decodes image i mod unique")
    ap.add_argument("--streams", choices=("zlib", "swiftpng"), default="zlib",
                    help="level-6 encoder for the input streams: host zlib, or the device deflater (swift-png bitstream)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: every GPU decodes --images images (weak), or --images in total (strong)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import swift_png_amd as spng
    from swift_png_amd.distributed import gather_decoded, shard
    s = spng.load(local)
    cores = os.cpu_count() or 1
    images, rows, streams = build_inputs(spng, s, args.unique, min(cores, 32), args.streams)
    U = spng.inflated_size(W, H, DEPTH, CHANNELS, False)
    S = spng.storage_size(W, H, DEPTH, CHANNELS)
    C = [len(z) for z in streams]

    lo, hi = shard(args.images, world, rank)                 # this rank's share of a 1024-image result
    weak = args.scaling == "weak" or world == 1
    n = args.images if weak else hi - lo                     # images this rank decodes per step
    first = rank * args.images if weak else lo               # global index of its first image
    d_streams = [s.to_device(z) for z in streams]
    # one contiguous slab each for scanline scratch and decoded rasters
    rows_cap = (U + 4096 + 255) & ~255
    d_rows = torch.empty(n * rows_cap, dtype=torch.uint8, device=s.tdev)
    d_out = torch.empty(n * S, dtype=torch.uint8, device=s.tdev)
    descs = (spng.ImageDesc * n)()
    for j in range(n):
        g = first + j
        z = d_streams[g % args.unique]
        descs[j] = spng.ImageDesc(z.data_ptr(), z.numel(), d_rows.data_ptr() + j * rows_cap, rows_cap,
                                  d_out.data_ptr() + j * S, W, H, DEPTH, CHANNELS, 0, 0, 0)
    gathered = None
    do_gather = world > 1 and not args.no_gather
    if do_gather and rank == 0:
        per = -(-args.images // world)
        gathered = [torch.empty(per * S, dtype=torch.uint8, device=s.tdev) for _ in range(world)]

    def step():
        s.decode_batch(descs, wait=False)
        if do_gather:
            # the only exchange step of the path: decoded rasters -> rank 0 over xGMI (RCCL)
            gather_decoded(d_out[lo * S:hi * S] if weak else d_out, S, args.images, world, rank, out=gathered)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # The gather is the one piece that cannot be exercised on the single-GPU development box: if RCCL
    # refuses it on this node, keep measuring the (communication-free) sharded decode and say so.
    gather_error = None
    if do_gather:
        try:
            step()
            fence()
        except Exception as exc:                               # noqa: BLE001
            gather_error = repr(exc)[:200]
            do_gather = False
    for _ in range(args.warmup):
        step()
    fence()
    s.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = {k: s.profile_get(getattr(spng, "K_" + k.upper())) for k in ("inflate", "unfilter")}
    s.profile(False)

    # parity of the whole batch: every slot equals its source raster, every status is DONE
    res = s.fetch_results(n)
    assert all(r.status == 0 and r.written == U for r in res), [r.status for r in res if r.status][:8]
    ref = [s.to_device(img.reshape(-1)) for img in images]
    for j in range(n):
        assert torch.equal(d_out[j * S:(j + 1) * S], ref[(first + j) % args.unique]), f"slot {first + j} differs"

    t = torch.tensor([dt], dtype=torch.float64, device=s.tdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if rank == 0:
        ms = dt / args.steps * 1e3
        total_c = sum(C[(first + j) % args.unique] for j in range(n))
        infl_ms = prof["inflate"][0] / max(1, prof["inflate"][1])
        unf_ms = prof["unfilter"][0] / max(1, prof["unfilter"][1])
        infl_bytes = total_c + n * U                 # algorithmic: read C, write U
        unf_bytes = n * (U + S)                      # algorithmic: read U, write S
        # HBM traffic per launch from the committed PMC passes (FETCH_SIZE / WRITE_SIZE, corrected as
        # MI355X_MICROARCH.md prescribes; profiles/r01_pmc_traffic.json), scaled to this batch; null if absent
        traffic = {"inflate": None, "unfilter": None}
        try:
            pmc = json.loads((ROOT / "profiles" / "r01_pmc_traffic.json").read_text())
            for k in traffic:
                traffic[k] = pmc["kernels"][k]["hbm_bytes_per_image"] * n
        except (OSError, KeyError, ValueError):
            pass
        dominant = "inflate" if infl_ms >= unf_ms else "unfilter"
        dom_bytes, dom_ms = (infl_bytes, infl_ms) if dominant == "inflate" else (unf_bytes, unf_ms)
        out = {
            "metric": "decoded_mpixels_per_s",
            "value": round(args.images * (world if weak else 1) * MPIX / (dt / args.steps), 1),
            "unit": "MPixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak" if weak else "strong",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.images} x 4096x4096 RGBA8 PNG decode (inflate+unfilter), mixed "
                                   f"filters (reference heuristic), level 6 ({args.streams} encoder); BASELINE configs[1]"
                                   + ("" if world == 1 else
                                      f" per GPU (global batch {args.images * world}); each rank's {hi - lo}-image shard "
                                      f"gathered to rank 0 over RCCL (configs[2])" if weak else
                                      f" sharded {n}/GPU, RCCL gather to rank 0 (configs[2])"),
                       "unique_images": args.unique, "compressed_ratio": round(U * args.unique / sum(C), 3),
                       "gather": bool(do_gather), **({"gather_error": gather_error} if gather_error else {})},
            "inflate_gbps": round(n * U / (infl_ms * 1e-3) / 1e9, 2),
            "roofline": {"bound": "hbm", "kernel": f"{dominant}_kernel",
                         "achieved": round(dom_bytes / (dom_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(dom_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                         "traffic": traffic[dominant], "ms_per_launch": round(dom_ms, 3)},
            "kernels": {
                "inflate": {"ms_per_launch": round(infl_ms, 3), "algorithmic_bytes": infl_bytes,
                            "gbps": round(infl_bytes / (infl_ms * 1e-3) / 1e9, 2)},
                "unfilter": {"ms_per_launch": round(unf_ms, 3), "algorithmic_bytes": unf_bytes,
                             "gbps": round(unf_bytes / (unf_ms * 1e-3) / 1e9, 2),
                             "frac_of_hbm_peak": round(unf_bytes / (unf_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                             "traffic": traffic["unfilter"]},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(streams, images, cores, max(16, cores))
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
