"""GPU probe: inflate throughput per stream on the benchmark's 4K streams."""
import sys; sys.path.insert(0, ".")
import sys, time, zlib
import numpy as np, torch
import swift_png_amd as spng
from swift_png_amd import synth

s = spng.load(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W = H = 4096
U = spng.inflated_size(W, H, 8, 4, False)
img = synth.image(0, W, H)
rows = s.filter(img.tobytes(), W, H, 8, 4, False)
cases = {"synth-l6": zlib.compress(rows, 6), "synth-l1": zlib.compress(rows, 1)}
noise = np.random.default_rng(1).integers(0, 256, U, dtype=np.uint8).tobytes()
cases["noise-l6"] = zlib.compress(noise, 6)
flat = bytes(U)
cases["zeros-l6"] = zlib.compress(flat, 6)
truth = {"synth-l6": rows, "synth-l1": rows, "noise-l6": noise, "zeros-l6": flat}
out = torch.empty(N * (U + 4096), dtype=torch.uint8, device=s.tdev)
import os
only = os.environ.get("PROBE_CASES")
for name, z in cases.items():
    if only and name not in only.split(","): continue
    K = int(os.environ.get("PROBE_DISTINCT", "1"))      # copies at distinct addresses: no sharing in L2
    dzs = [s.to_device(z) for _ in range(K)]
    sd = (spng.StreamDesc * N)(*[spng.StreamDesc(dzs[i % K].data_ptr(), dzs[i % K].numel(), out.data_ptr() + i * (U + 4096), U + 4096, 0, 0) for i in range(N)])
    res = (spng.Result * N)()
    s.lib.spng_inflate_batch(s.ctx, sd, N, None, res)
    for _ in range(int(os.environ.get("PROBE_REPS", "1"))):
        s.profile(True)
        s.lib.spng_inflate_batch(s.ctx, sd, N, None, res)
        ms, n = s.profile_get(spng.K_INFLATE); s.profile(False)
        if os.environ.get("PROBE_REPS"): print(f"  launch: {ms/n:.1f} ms")
    ok = bytes(out[(N - 1) * (U + 4096):(N - 1) * (U + 4096) + U].cpu().numpy()) == truth[name]
    print(f"inflate {name} N={N}: ratio {U/len(z):.2f} kernel {ms/n:.1f} ms status {res[0].status} ok={ok} -> per-stream {U/(ms/n*1e-3)/1e6:.1f} MB/s out, batch {N*U/(ms/n*1e-3)/1e9:.2f} GB/s")
