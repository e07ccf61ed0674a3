#!/usr/bin/env python3
"""Timing probe: N x 4K zlib level-6 streams through spng_inflate_batch (parallel pipeline)."""
import sys, time, zlib
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import swift_png_amd as spng
from swift_png_amd import synth
s = spng.load(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
kind = sys.argv[2] if len(sys.argv) > 2 else "zlib"
uniq = 4
rows, zs = [], []
for k in range(uniq):
    if kind == "zlib":
        img = synth.image(k, 4096, 4096)
        r = s.filter(img.tobytes(), 4096, 4096, 8, 4, False)
        z = zlib.compress(r, 6)
    else:
        img = synth.image(k, 1024, 1024)
        r = s.filter(img.tobytes(), 1024, 1024, 8, 4, False)
        z = s.deflate(r, 6)
    rows.append(r); zs.append(z)
d = [s.to_device(z) for z in zs]
ref = [s.to_device(r) for r in rows]
streams = [d[i % uniq] for i in range(N)]
caps = [len(rows[i % uniq]) + 4096 for i in range(N)]
for it in range(3):
    s.profile(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs, res = s.inflate_batch(streams, caps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    par = s.profile_get(spng.K_PINFLATE)[0]; ser = s.profile_get(spng.K_INFLATE)[0]
    s.profile(False)
    ok = sum(r.status == 0 and r.written == len(rows[i % uniq]) and torch.equal(o[:r.written], ref[i % uniq]) for i, (o, r) in enumerate(zip(outs, res)))
    fast = sum(r.reserved == 1 for r in res)
    tot = sum(len(rows[i % uniq]) for i in range(N))
    print(f"{kind} N={N} ok={ok} fast={fast} parallel {par:.2f} ms serial {ser:.2f} ms wall {dt*1e3:.1f} ms -> {tot/par/1e6:.1f} GB/s out", flush=True)
    del outs
