#!/usr/bin/env python3
"""Why does a stream leave the parallel pipeline?  (SPNG_TRACE_PINFLATE=1)"""
import sys, zlib
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import swift_png_amd as spng
from swift_png_amd import synth
s = spng.load(0)
img = synth.image(3, 1024, 1024)
rows = s.filter(img.tobytes(), 1024, 1024, 8, 4, False)
small = rows[:200000]
for name, data in (("small", small), ("full", rows)):
    z = s.deflate(data, 6)
    print(name, len(data), "->", len(z), flush=True)
    for seg in (0, 65536):
        s.configure(spng.CFG_SEGMENT_BYTES, seg)
        outs, res = s.inflate_batch([s.to_device(z)], [len(data) + 16])
        r = res[0]
        print("  seg", seg, "status", r.status, "path", r.reserved, "written", r.written, flush=True)
