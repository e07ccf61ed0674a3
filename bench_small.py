"""bench_small.py -- the `small_images` leg of bench.py: the regime the reference itself publishes numbers for.

swift-png's decode benchmark (Benchmarks/Decompression/Swift/Main.swift:86-116, Benchmarks/README.md:57) times
`PNG.Image.decompress` + `unpack(as: PNG.RGBA<UInt8>.self)` of ONE small image per call -- 400 x 240 / 400 x 260 test
images in every colour format.  The same 28 files travel with this repository as test fixtures (tests/golden/encode/
*.baseline.png = the reference's Tests/Baselines).  The leg replicates them to `n` files per call, resident in HBM, and runs
file -> pixels on the device: spng_lex_batch (chunk walk, CRC-32, IDAT assembly; IHDR fields back to the host) ->
spng_decode_batch (inflate + defilter) -> spng_unpack_batch (RGBA<UInt8>).  Every distinct file's pixels are compared with
the CPU oracle's decode of the same file AFTER the clock; the oracle (bench_cpu.py `files`) is timed beside it on the host
cores as a reported baseline."""
from __future__ import annotations

import ctypes
import json
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
FIXTURES = ROOT / "tests" / "golden" / "encode"


def _palette_quads(png):
    q = np.full((len(png.palette) // 3, 4), 255, np.uint8)
    q[:, :3] = np.frombuffer(png.palette, np.uint8).reshape(-1, 3)
    if png.trns:
        t = np.frombuffer(png.trns, np.uint8)[:len(q)]
        q[:len(t), 3] = t
    return q.tobytes()


def run_small_images(torch, spng, s, n=8192, steps=3, cpu=True, cores=1):
    sys.path.insert(0, str(ROOT / "tests"))
    import pnghelp as ph
    paths = sorted(FIXTURES.glob("*.baseline.png"))
    files = [p.read_bytes() for p in paths]
    pngs = [ph.parse_png(f) for f in files]
    k = len(files)
    assert k == 28, k
    d_files = [s.to_device(f) for f in files]
    d_pal = [s.to_device(_palette_quads(p)) if p.color == 3 else None for p in pngs]
    U = [spng.inflated_size(p.width, p.height, p.depth, p.channels, p.interlaced) for p in pngs]
    S = [spng.storage_size(p.width, p.height, p.depth, p.channels) for p in pngs]
    P = [p.width * p.height * 4 for p in pngs]                   # RGBA<UInt8>
    al = lambda v: (v + 255) & ~255                              # noqa: E731
    # slot j holds file j mod 28: its IDAT bytes, scanlines, storage and pixels at offsets of their own
    offs = {"idat": [0], "rows": [0], "sto": [0], "px": [0]}
    for j in range(n):
        u = j % k
        offs["idat"].append(offs["idat"][-1] + al(len(files[u])))
        offs["rows"].append(offs["rows"][-1] + al(U[u] + 4096))
        offs["sto"].append(offs["sto"][-1] + al(S[u]))
        offs["px"].append(offs["px"][-1] + al(P[u]))
    d_idat = torch.empty(offs["idat"][-1], dtype=torch.uint8, device=s.tdev)
    d_rows = torch.empty(offs["rows"][-1], dtype=torch.uint8, device=s.tdev)
    d_sto = torch.empty(offs["sto"][-1], dtype=torch.uint8, device=s.tdev)
    d_px = torch.empty(offs["px"][-1], dtype=torch.uint8, device=s.tdev)
    fdescs = (spng.FileDesc * n)()
    for j in range(n):
        f = d_files[j % k]
        fdescs[j] = spng.FileDesc(f.data_ptr(), f.numel(), d_idat.data_ptr() + offs["idat"][j], al(len(files[j % k])))
    infos = (spng.Lexed * n)()
    idescs = (spng.ImageDesc * n)()
    udescs = (spng.UnpackDesc * n)()
    dres = s.empty(n * ctypes.sizeof(spng.Result))

    def step(first):
        assert s.lib.spng_lex_batch(s.ctx, fdescs, n, None, infos) == 0          # (IHDR fields and IDAT lengths come back to the host)
        if first:
            for j in range(n):
                r, u = infos[j], j % k
                p = pngs[u]
                assert r.status == 0 and (r.width, r.height, r.depth, r.color, r.interlace) == (p.width, p.height, p.depth, p.color, int(p.interlaced)), (j, r.status)
                assert r.idat_len == len(p.idat)
                idescs[j] = spng.ImageDesc(d_idat.data_ptr() + offs["idat"][j], r.idat_len, d_rows.data_ptr() + offs["rows"][j], U[u] + 4096,
                                           d_sto.data_ptr() + offs["sto"][j], p.width, p.height, p.depth, p.channels, int(p.interlaced), 0, 0)
                udescs[j] = spng.UnpackDesc(d_sto.data_ptr() + offs["sto"][j], d_px.data_ptr() + offs["px"][j],
                                            d_pal[u].data_ptr() if d_pal[u] is not None else None, p.width, p.height,
                                            len(p.palette) // 3 if p.color == 3 else 0, (ctypes.c_uint16 * 3)(), p.depth, p.channels,
                                            1 if p.color == 3 else 0, 0, 0, 8)
        assert s.lib.spng_decode_batch(s.ctx, idescs, n, ctypes.c_void_p(dres.data_ptr()), None) == 0
        assert s.lib.spng_unpack_batch(s.ctx, udescs, n) == 0

    torch.cuda.synchronize()
    step(True)
    step(False)                                                   # (the pipeline's pool is sized by the call before)
    torch.cuda.synchronize()
    names = ("lex", "pinf_find", "pinf_decode", "pinf_resolve", "inflate", "unfilter", "unpack")
    s.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step(False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = {kk: round(s.profile_get(getattr(spng, "K_" + kk.upper()))[0] / steps, 3) for kk in names}
    s.profile(False)
    # parity, after the clock: every status DONE, and the pixels of the first and the last replica of every distinct file equal
    # the oracle's decode of that file unpacked as RGBA<UInt16> and narrowed to UInt8 (PNG.RGBA.swift:259-365)
    res = list((spng.Result * n).from_buffer_copy(bytes(dres.cpu().numpy())))
    assert all(r.status == 0 for r in res), [(j, r.status) for j, r in enumerate(res) if r.status][:6]
    pipeline = sum(r.reserved == 1 for r in res)
    for u, p in enumerate(pngs):
        st, storage, _ = ph.orc_decode(p)
        assert st == 0
        want = (ph.unpack_rgba16(np.asarray(storage), p) >> 8).astype(np.uint8).reshape(-1)
        for j in (u, u + (n - 1 - u) // k * k):
            got = d_px[offs["px"][j]:offs["px"][j] + P[u]].cpu().numpy()
            assert np.array_equal(got, want), f"file {paths[u].name} (slot {j}): pixels differ from the oracle's"
    mpix = sum(p.width * p.height for p in pngs) / 1e6 * (n / k)
    file_bytes = sum(len(files[j % k]) for j in range(n))
    out = {"workload": f"{n} PNG files per call = the 28 images of the reference's Tests/Baselines (400x240 / 400x260, every colour format: "
                       f"indexed8, v8/16, va8/16, rgb8/16, rgba8/16) x {n // k}, resident in HBM: lex + CRC-32 -> inflate -> defilter -> "
                       f"unpack(as: RGBA<UInt8>); what Benchmarks/Decompression/Swift/Main.swift:97-109 times per image",
           "files": n, "file_bytes": file_bytes, "ms_per_step": round(dt * 1e3, 2), "us_per_image": round(dt * 1e6 / n, 2),
           "images_per_s": round(n / dt, 0), "mpixels_per_s": round(mpix / dt, 1), "kernels_ms": prof,
           "pipeline_streams": pipeline, "serial_streams": n - pipeline, "bit_exact": True,
           "checked": "both the first and the last replica of each of the 28 files against the oracle's pixels, every status DONE"}
    if cpu:
        run = subprocess.run([sys.executable, str(ROOT / "bench_cpu.py"), "files", str(FIXTURES), str(cores), str(max(28 * 8, 28 * cores))],
                             capture_output=True, text=True, timeout=600)
        assert run.returncode == 0, run.stderr[-400:]
        c = json.loads(run.stdout.strip().splitlines()[-1])
        rate = c["tasks"] / c["wall_s"]
        out["cpu_baseline"] = {"value": round(rate, 1), "unit": "images/s", "mpixels_per_s": round(rate * mpix / n, 2), "cores": cores, "kind": "port",
                               "us_per_image_one_core": round(c["task_s"] * 1e6, 1),
                               "sample": f"{c['tasks']} decodes (inflate + defilter + assign; no unpack) of the same 28 files on {cores} worker "
                                         f"processes, {c['wall_s']:.2f} s wall"}
        out["speedup_vs_cpu_baseline"] = round(n / dt / rate, 1)
    del d_idat, d_rows, d_sto, d_px
    torch.cuda.empty_cache()
    return out
