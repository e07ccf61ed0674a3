"""GPU probe: filter-select kernel throughput on distinct 4K rasters."""
import sys; sys.path.insert(0, ".")
import sys, numpy as np, torch
import swift_png_amd as spng
from swift_png_amd import synth
s = spng.load(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = H = 4096
U = spng.inflated_size(W, H, 8, 4, False); S = W * H * 4
img = s.to_device(synth.image(0, W, H).reshape(-1))
src = torch.empty(N * S, dtype=torch.uint8, device=s.tdev); out = torch.empty(N * U, dtype=torch.uint8, device=s.tdev)
for i in range(N): src[i * S:(i + 1) * S] = img
descs = [s.image_desc(None, out[i * U:(i + 1) * U], src[i * S:(i + 1) * S], W, H, 8, 4, False, rows_cap=U) for i in range(N)]
s.filter_batch(descs); torch.cuda.synchronize()
s.profile(True)
for _ in range(3): s.filter_batch(descs)
torch.cuda.synchronize()
ms, n = s.profile_get(spng.K_FILTER); s.profile(False)
print(f"filter N={N}: kernel {ms/n:.2f} ms -> {N*(U+S)/(ms/n*1e-3)/1e9:.1f} GB/s ({N*(U+S)/(ms/n*1e-3)/8e12*100:.1f}% of 8 TB/s)")
