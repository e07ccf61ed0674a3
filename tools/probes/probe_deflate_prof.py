#!/usr/bin/env python3
"""Phase cycle counts of the deflate kernels (library built with -DSPNG_DEFLATE_PROF, passed as SPNG_LIB)."""
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
         ("synth 1024^2 rows", s.filter(synth.image(3, 1024, 1024).tobytes(), 1024, 1024, 8, 4, False))]
for name, rows in cases:
    for level in (9, 6):
        d = s.to_device(rows)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs, res = s.deflate_batch([d], level)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name} level {level}: {dt:.2f} s, {len(rows)/dt/1e6:.2f} MB/s", flush=True)
