"""GPU probe: deflate level 6 throughput on the benchmark's filtered rows."""
import sys; sys.path.insert(0, ".")
import sys, time, zlib
import numpy as np, torch
import swift_png_amd as spng
from swift_png_amd import synth
s = spng.load(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
W = H = 4096
rows = [s.to_device(s.filter(synth.image(i, W, H).tobytes(), W, H, 8, 4, False)) for i in range(min(N, 4))]
streams = [rows[i % len(rows)] for i in range(N)]
s.profile(True)
t0 = time.time()
outs, res = s.deflate_batch(streams, level)
dt = time.time() - t0
ms, n = s.profile_get(spng.K_DEFLATE); s.profile(False)
U = streams[0].numel()
print(f"deflate L{level} N={N}: kernel {ms/n:.0f} ms, per-stream {U/(ms/n*1e-3)/1e6:.2f} MB/s in, batch {N*U/(ms/n*1e-3)/1e9:.3f} GB/s, ratio {U/res[0].written:.2f}, status {res[0].status}")
z = bytes(outs[0][:res[0].written].cpu().numpy())
assert zlib.decompress(z) == bytes(rows[0].cpu().numpy())
print("zlib round trip ok")
