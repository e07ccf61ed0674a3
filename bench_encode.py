"""bench.py --mode encode: BASELINE.json configs[3] -- random 4096x4096 RGBA8 rasters -> PNG at level 9 on one
MI355X (filter-select + LZ77 match search + shortest-path parse + Huffman emit), bit-exact with the CPU path.

A step = one spng_encode_batch over the whole batch (filter kernel, then the level-9 deflate kernel in groups
that fit the match-graph slab).  `value` = encoded MPixels/s.  Parity: every stream of the step is inflated by
zlib and compared with the filter kernel's scanlines, one stream is compared bit for bit with the oracle's
level-9 stream of the same scanlines inside the cpu_baseline leg (the oracle is pinned on swift-png's own
committed level-9 outputs)."""
from __future__ import annotations

import ctypes
import json
import os
import subprocess
import sys
import time
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
W = H = 4096
DEPTH, CHANNELS = 8, 4
MPIX = W * H / 1e6
HBM_PEAK_GBPS = 8000.0


def encode_cpu_all_cores(rows: bytes, level: int, cores: int):
    """all host cores, the oracle's deflate at `level` on 1 MiB slices of the scanlines (one worker process per core)"""
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix="spng_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        (Path(tmp) / "rows").write_bytes(rows[:64 << 20])
        out = subprocess.run([sys.executable, str(ROOT / "bench_cpu.py"), "deflate", str(Path(tmp) / "rows"), str(cores), str(level),
                              str(1 << 20)], capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-400:]
        r = json.loads(out.stdout.strip().splitlines()[-1])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {"value": round(r["tasks"] * r["slice"] / 4 / 1e6 / r["wall_s"], 2), "unit": "MPixels/s", "cores": cores, "kind": "port",
            "sample": f"{r['tasks']} slices of 1 MiB of the same scanlines, oracle deflate level {level}, one worker process per core, "
                      f"{r['wall_s']:.1f} s wall, {r['task_s']:.2f} s per slice (filter excluded)"}


def run_encode(args, torch, dist, spng, s, rank, world, rasters_kind="random"):
    """rasters_kind: "random" (BASELINE configs[3]: incompressible rasters) or "synthetic" (the structured images of the decode
    headline, swift_png_amd.synth: what the level-6 inputs of configs[1] are made from)"""
    from swift_png_amd.distributed import shard
    lo, hi = shard(args.images, world, rank)
    weak = args.scaling == "weak" or world == 1
    n = args.images if weak else hi - lo
    unique = min(args.unique, n)
    U = spng.inflated_size(W, H, DEPTH, CHANNELS, False)
    S = spng.storage_size(W, H, DEPTH, CHANNELS)
    cap = s.lib.spng_deflate_bound(U)
    gen = torch.Generator(device=s.tdev)
    gen.manual_seed(1234 + rank)
    if rasters_kind == "synthetic":
        from swift_png_amd import synth
        rasters = [s.to_device(synth.image(k, W, H).tobytes()) for k in range(unique)]
    else:
        rasters = [torch.randint(0, 256, (S,), dtype=torch.uint8, device=s.tdev, generator=gen) for _ in range(unique)]
    d_rows = torch.empty(n * U, dtype=torch.uint8, device=s.tdev)
    d_out = torch.empty(n * cap, dtype=torch.uint8, device=s.tdev)
    descs = (spng.ImageDesc * n)()
    for j in range(n):
        r = rasters[j % unique]
        descs[j] = spng.ImageDesc(d_out.data_ptr() + j * cap, cap, d_rows.data_ptr() + j * U, U, r.data_ptr(),
                                  W, H, DEPTH, CHANNELS, 0, 0, 0)
    dres = s.empty(n * ctypes.sizeof(spng.Result))

    def step():
        st = s.lib.spng_encode_batch(s.ctx, descs, args.level, n, ctypes.c_void_p(dres.data_ptr()), None)
        assert st == 0, st

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    s.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = {k: s.profile_get(getattr(spng, "K_" + k.upper()))[0] / args.steps for k in ("filter", "deflate", "dfl_search", "dfl_parse")}
    s.profile(False)
    res = list((spng.Result * n).from_buffer_copy(bytes(dres.cpu().numpy())))
    assert all(r.status == 0 for r in res), [r.status for r in res if r.status][:8]
    total_c = sum(r.written for r in res)
    # every distinct stream inflates (zlib) to the filter kernel's scanlines
    streams = {}
    for j in range(min(n, unique)):
        z = bytes(d_out[j * cap:j * cap + res[j].written].cpu().numpy())
        rows = bytes(d_rows[j * U:(j + 1) * U].cpu().numpy())
        assert zlib.decompress(z) == rows, f"stream {j} does not inflate to its scanlines"
        streams[j] = (z, rows)
    t = torch.tensor([dt], dtype=torch.float64, device=s.tdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if rank != 0:
        return None
    ms = dt / args.steps * 1e3
    kernels = {
        "filter": {"ms_per_step": round(prof["filter"], 3), "algorithmic_bytes": n * (S + U),
                   "gbps": round(n * (S + U) / (prof["filter"] * 1e-3) / 1e9, 2),
                   "frac_of_hbm_peak": round(n * (S + U) / (prof["filter"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
        "deflate": {"ms_per_step": round(prof["deflate"], 3), "algorithmic_bytes": n * U + total_c,
                    "gbps": round((n * U + total_c) / (prof["deflate"] * 1e-3) / 1e9, 3),
                    "frac_of_hbm_peak": round((n * U + total_c) / (prof["deflate"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6)},
    }
    dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
    # HBM traffic of the deflate kernels from the committed rocprofv3 --pmc passes of this workload's deflate step (1024 random
    # 64 MiB streams at level 9: profiles/r06_pmc_encode.json, tools/final_run.sh); FETCH_SIZE x 2 as on the decode side
    traffic, traffic_src = None, None
    try:
        pmc = json.loads((ROOT / "profiles" / "r06_pmc_encode.json").read_text())
        if rasters_kind == "random" and args.level == 9 and pmc.get("streams") == args.images and dom == "deflate":
            traffic = int(pmc["deflate_hbm_bytes_per_step"])
            traffic_src = ("profiles/r06_pmc_encode.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same 1024 x 64 MiB level-9 deflate, not measured "
                           f"in this run; taken on source digest {pmc.get('source_digest')}, running {spng.source_digest()})")
            kernels["deflate"]["traffic_by_kernel"] = pmc.get("kernels")
    except (OSError, KeyError, ValueError):
        pass
    # the two kernels of a round (both inside "deflate", at every level since round 5): the chip-wide match search reads U, the
    # one-wave-per-stream parse reads U (levels 0-7: and the search's 4 bytes per position) and writes C
    kernels["deflate"]["search_ms"] = round(prof["dfl_search"], 3)
    kernels["deflate"]["parse_ms"] = round(prof["dfl_parse"], 3)
    out = {
        "metric": "encoded_mpixels_per_s", "value": round(args.images * (world if weak else 1) * MPIX / (dt / args.steps), 2),
        "unit": "MPixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
        "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": f"{args.images} x 4096x4096 RGBA8 {rasters_kind} rasters -> filter-select + DEFLATE level {args.level} "
                               f"(spng_encode_batch)" + ("; BASELINE configs[3]" if rasters_kind == "random" and args.level == 9 else ""),
                   "unique_images": unique,
                   "compressed_ratio": round(n * U / total_c, 4)},
        "roofline": {"bound": "hbm", "kernel": dom + "_kernel", "achieved": kernels[dom]["gbps"], "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": kernels[dom]["frac_of_hbm_peak"], "traffic": traffic,
                     "ms_per_launch": kernels[dom]["ms_per_step"], **({"traffic_source": traffic_src} if traffic_src else {})},
        "kernels": kernels,
    }
    if world == 1 and not args.no_cpu_baseline:
        # the oracle on one core, on a bounded sample: the first 2 MiB of scanlines of stream 0 (level 9 on the
        # whole 64 MiB image takes minutes on a core); also the bit-exactness anchor of this run
        sys.path.insert(0, str(ROOT / "tests"))
        import pnghelp as ph
        import hashlib
        z, rows = streams[0]
        # the WHOLE stream of image 0 -- 64 MiB of scanlines, 32 blocks at the vertex cap -- against the oracle's (bit for bit,
        # by digest), timed as the one-core figure
        t0 = time.perf_counter()
        want = ph.orc_deflate(rows, args.level)
        dtc = time.perf_counter() - t0
        assert len(z) == len(want) and hashlib.sha256(z).digest() == hashlib.sha256(want).digest(), \
            "device level-%d stream of image 0 differs from the oracle's" % args.level
        out["parity"] = {"stream": 0, "bytes": len(rows), "stream_bytes": len(z), "sha256": hashlib.sha256(z).hexdigest()[:16],
                         "equals_oracle": True, "all_streams_inflate_to_their_scanlines": True}
        out["cpu_baseline"] = {"value": round(len(rows) / 4 / 1e6 / dtc, 3), "unit": "MPixels/s", "cores": 1, "kind": "port",
                               "sample": f"oracle deflate level {args.level} of all {len(rows)} scanline bytes of stream 0 "
                                         f"(filter excluded), {dtc:.1f} s; the device stream of the same bytes is identical"}
        try:
            from bench import host_cores
            allc = encode_cpu_all_cores(rows, args.level, host_cores())
            allc["one_core"] = out["cpu_baseline"]
            out["cpu_baseline"] = allc
        except Exception as exc:                                   # noqa: BLE001
            out["cpu_baseline"]["all_cores_error"] = repr(exc)[:200]
    return out


def run_encode_photographic(torch, spng, s, level=9, images=256, size=1024, cpu=True):
    """SURVEY 8d item 3's second encode input: images with structure (swift_png_amd.synth: gradients, edges, texture -- what a
    photograph gives a PNG encoder) instead of noise.  Compressible input is where the level >= 8 search has work to do
    (DESIGN 4.5), so the rasters are 1024 x 1024: `images` streams of 4 MiB, all resident at once."""
    from swift_png_amd import synth
    w = h = size
    unique = min(8, images)
    U = spng.inflated_size(w, h, DEPTH, CHANNELS, False)
    S = spng.storage_size(w, h, DEPTH, CHANNELS)
    cap = s.lib.spng_deflate_bound(U)
    rasters = [s.to_device(synth.image(100 + k, w, h).tobytes()) for k in range(unique)]
    d_rows = torch.empty(images * U, dtype=torch.uint8, device=s.tdev)
    d_out = torch.empty(images * cap, dtype=torch.uint8, device=s.tdev)
    descs = (spng.ImageDesc * images)()
    for j in range(images):
        r = rasters[j % unique]
        descs[j] = spng.ImageDesc(d_out.data_ptr() + j * cap, cap, d_rows.data_ptr() + j * U, U, r.data_ptr(), w, h, DEPTH, CHANNELS, 0, 0, 0)
    dres = s.empty(images * ctypes.sizeof(spng.Result))
    st = s.lib.spng_encode_batch(s.ctx, descs, level, images, ctypes.c_void_p(dres.data_ptr()), None)     # (warm-up: the first call allocates the slab)
    assert st == 0, st
    torch.cuda.synchronize()
    s.profile(True)
    t0 = time.perf_counter()
    st = s.lib.spng_encode_batch(s.ctx, descs, level, images, ctypes.c_void_p(dres.data_ptr()), None)
    assert st == 0, st
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = {k: s.profile_get(getattr(spng, "K_" + k.upper()))[0] for k in ("filter", "deflate")}
    s.profile(False)
    res = list((spng.Result * images).from_buffer_copy(bytes(dres.cpu().numpy())))
    assert all(r.status == 0 for r in res), [r.status for r in res if r.status][:8]
    total_c = sum(r.written for r in res)
    for j in range(unique):
        z = bytes(d_out[j * cap:j * cap + res[j].written].cpu().numpy())
        assert zlib.decompress(z) == bytes(d_rows[j * U:(j + 1) * U].cpu().numpy()), f"stream {j} does not inflate to its scanlines"
    out = {"workload": f"{images} x {w}x{h} RGBA8 synthetic photographs (swift_png_amd.synth) -> filter-select + DEFLATE level {level}",
           "ms": round(dt * 1e3, 1), "mpixels_per_s": round(images * w * h / 1e6 / dt, 2),
           "per_stream_mb_per_s": round(U / 1e6 / (prof["deflate"] * 1e-3), 3), "aggregate_mb_per_s": round(images * U / 1e6 / (prof["deflate"] * 1e-3), 1),
           "compressed_ratio": round(images * U / total_c, 3),
           "kernels_ms": {k: round(v, 2) for k, v in prof.items()}, "inflates_to_its_scanlines": True}
    if cpu:
        # the oracle on the same scanlines: one worker process per core, one whole 4 MiB stream each (the match search has work
        # to do here, unlike on the random rasters of configs[3])
        try:
            from bench import host_cores
            rows0 = bytes(d_rows[:U].cpu().numpy())
            cores = host_cores()
            import shutil
            import tempfile
            tmp = tempfile.mkdtemp(prefix="spng_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
            try:
                (Path(tmp) / "rows").write_bytes(rows0)
                r = subprocess.run([sys.executable, str(ROOT / "bench_cpu.py"), "deflate", str(Path(tmp) / "rows"), str(cores), str(level), str(U)],
                                   capture_output=True, text=True, timeout=900)
                assert r.returncode == 0, r.stderr[-400:]
                j = json.loads(r.stdout.strip().splitlines()[-1])
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
            out["cpu_baseline"] = {"value": round(j["tasks"] * w * h / 1e6 / j["wall_s"], 2), "unit": "MPixels/s", "cores": cores, "kind": "port",
                                   "per_core_mb_per_s": round(U / 1e6 / j["task_s"], 3),
                                   "sample": f"{j['tasks']} whole streams of the same scanlines ({U} bytes each), oracle deflate level {level}, one "
                                             f"worker process per core, {j['wall_s']:.1f} s wall, {j['task_s']:.2f} s per stream"}
        except Exception as exc:                                   # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(exc)[:200]}
    return out
