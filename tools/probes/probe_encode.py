#!/usr/bin/env python3
"""How long does one 4K stream take at level 9 (random / photographic-like)?"""
import sys, time, zlib
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import torch, numpy as np
import swift_png_amd as spng
from swift_png_amd import synth
s = spng.load(0)
rng = np.random.default_rng(1)
cases = [("random 4MB", rng.integers(0, 256, 4 << 20, dtype=np.uint8).tobytes()),
         ("synth 1024^2 rows", s.filter(synth.image(3, 1024, 1024).tobytes(), 1024, 1024, 8, 4, False)),
         ("random 64MB", rng.integers(0, 256, 64 << 20, dtype=np.uint8).tobytes())]
per_mb = {}
for name, rows in cases:
    for level in (9, 6):
        if len(rows) > (8 << 20) and per_mb.get(level, 0) * 64 > 30:
            print(f"{name} level {level}: skipped (would take ~{per_mb[level] * 64:.0f} s)", flush=True)
            continue
        d = s.to_device(rows)
        for n in (1, 64):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            outs, res = s.deflate_batch([d] * n, level)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            z = bytes(outs[0][:res[0].written].cpu().numpy())
            ok = zlib.decompress(z) == rows
            if name.startswith("random 4MB") and n == 1:
                per_mb[level] = dt / 4
            print(f"{name} level {level} x{n}: {dt:.2f} s, {len(rows)/dt/1e6*n:.2f} MB/s total, ratio {len(rows)/len(z):.3f}, ok={ok}", flush=True)
            del outs
