"""GPU probe: unfilter throughput on distinct 4K inputs (real HBM traffic)."""
import sys; sys.path.insert(0, ".")
import sys, time, zlib
import numpy as np, torch
import swift_png_amd as spng
from swift_png_amd import synth

s = spng.load(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = H = 4096
U = spng.inflated_size(W, H, 8, 4, False); S = W * H * 4
img = synth.image(0, W, H)
rows = s.filter(img.tobytes(), W, H, 8, 4, False)
base = np.frombuffer(rows, np.uint8).reshape(H, W * 4 + 1)
print("filter hist", np.bincount(base[:, 0], minlength=5))
src = torch.empty(N * U, dtype=torch.uint8, device=s.tdev)
out = torch.empty(N * S, dtype=torch.uint8, device=s.tdev)
want = s.to_device(img.reshape(-1))
import os
for forced in [None if x == "m" else int(x) for x in os.environ.get("FORCED", "m,0,1,2,3,4").split(",")]:
    r = base.copy()
    if forced is not None: r[:, 0] = forced
    one = s.to_device(r.reshape(-1))
    for i in range(N): src[i * U:(i + 1) * U] = one
    descs = [s.image_desc(None, src[i * U:(i + 1) * U], out[i * S:(i + 1) * S], W, H, 8, 4, False, rows_cap=U) for i in range(N)]
    s.unfilter_batch(descs); torch.cuda.synchronize()
    if forced is None:
        assert torch.equal(out[:S], want) and torch.equal(out[(N - 1) * S:], want), "unfilter mismatch"
    s.profile(True)
    for _ in range(3): s.unfilter_batch(descs)
    torch.cuda.synchronize()
    ms, n = s.profile_get(spng.K_UNFILTER); s.profile(False)
    print(f"unfilter forced={forced} N={N}: kernel {ms/n:.2f} ms -> {N*(U+S)/(ms/n*1e-3)/1e9:.1f} GB/s ({N*(U+S)/(ms/n*1e-3)/8e12*100:.1f}% of 8 TB/s)")
