"""GPU probe: unfilter kernel time of one library build (SPNG_LIB=...) on 256 x 4K RGBA8, mixed filters, distinct buffers.
PROBE_STRIDE=n: a tuning build compiled with -DSPNG_PROBE_ALIGNED_ROWS=n reads rows n bytes apart with the filter byte at
offset 15 (data 16-byte aligned) -- what the decoder's private scanline scratch could look like."""
import os, sys; sys.path.insert(0, ".")
import numpy as np, torch
import swift_png_amd as spng
from swift_png_amd import synth

s = spng.load(0)
N, W, H = int(os.environ.get("PROBE_N", "256")), 4096, 4096
stride = int(os.environ.get("PROBE_STRIDE", "0"))
U = spng.inflated_size(W, H, 8, 4, False); S = W * H * 4
per = stride * H + 64 if stride else U
src = torch.zeros(N * per, dtype=torch.uint8, device=s.tdev)
out = torch.empty(N * S, dtype=torch.uint8, device=s.tdev)
imgs = [synth.image(k, W, H) for k in range(4)]
rows = [np.frombuffer(s.filter(im.tobytes(), W, H, 8, 4, False), np.uint8).reshape(H, W * 4 + 1) for im in imgs]
if os.environ.get("PROBE_HIST"): print("filter histogram of image 0:", np.bincount(rows[0][:, 0], minlength=5))
dev = []
for r in rows:
    if stride:
        a = np.zeros((H, stride), np.uint8); a[:, 15:15 + W * 4 + 1] = r
        a = np.concatenate([a.reshape(-1), np.zeros(64, np.uint8)])
    else:
        a = r.reshape(-1)
    dev.append(s.to_device(a))
for i in range(N): src[i * per:(i + 1) * per] = dev[i % 4]
want = [s.to_device(im.reshape(-1)) for im in imgs]
descs = [s.image_desc(None, src[i * per:(i + 1) * per], out[i * S:(i + 1) * S], W, H, 8, 4, False, rows_cap=per) for i in range(N)]
s.unfilter_batch(descs); torch.cuda.synchronize()
ok = all(torch.equal(out[i * S:(i + 1) * S], want[i % 4]) for i in (0, 1, 2, 3, N - 1))
s.profile(True)
for _ in range(5): s.unfilter_batch(descs)
torch.cuda.synchronize()
ms, n = s.profile_get(spng.K_UNFILTER); s.profile(False)
print(f"{os.environ.get('SPNG_LIB', 'default').split('/')[-1]:28s} stride={stride or W*4+1}: {ms/n:.3f} ms per {N} images -> {N*(U+S)/(ms/n*1e-3)/1e9:.0f} GB/s = "
      f"{N*(U+S)/(ms/n*1e-3)/8e12*100:.2f}% of 8 TB/s  bit-exact={ok}", flush=True)
