#!/usr/bin/env python3
"""BASELINE configs[4]: one 8192x8192 RGBA16 Adam7 image; per-stage times."""
import sys, time, zlib
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import torch
import swift_png_amd as spng
from swift_png_amd import synth
s = spng.load(0)
w = h = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
img = synth.image(11, w, h, 4, 16)
raw = img.tobytes()
t0 = time.perf_counter(); rows = s.filter(raw, w, h, 16, 4, True); print("filter (host buffers)", time.perf_counter() - t0)
t0 = time.perf_counter(); z = zlib.compress(rows, 6); print("zlib", time.perf_counter() - t0, len(rows), len(z))
d_idat, d_rows, d_out = s.to_device(z), s.empty(len(rows) + 4096), s.empty(len(raw))
desc = s.image_desc(d_idat, d_rows, d_out, w, h, 16, 4, True, 0, rows_cap=len(rows) + 4096)
for it in range(3):
    s.profile(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = s.decode_batch([desc])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    names = ["pinf_find", "pinf_decode", "pinf_resolve", "inflate", "unfilter", "scatter"]
    prof = {k: round(s.profile_get(getattr(spng, "K_" + k.upper()))[0], 2) for k in names}
    s.profile(False)
    print(f"decode {dt*1e3:.1f} ms status {res[0].status} path {res[0].reserved}", prof, flush=True)
print("equal", bytes(d_out.cpu().numpy()) == raw)
