"""GPU probe: unfilter kernel time over the piece-rows knob (SPNG_CFG_UNFILTER_PIECE_ROWS), 256 x 4K RGBA8, mixed filters.
Run once per library build (SPNG_LIB=...)."""
import os, sys; sys.path.insert(0, ".")
import numpy as np, torch
import swift_png_amd as spng
from swift_png_amd import synth

s = spng.load(0)
N, W, H = int(os.environ.get("PROBE_N", "256")), 4096, 4096
U = spng.inflated_size(W, H, 8, 4, False); S = W * H * 4
src = torch.empty(N * U, dtype=torch.uint8, device=s.tdev)
out = torch.empty(N * S, dtype=torch.uint8, device=s.tdev)
imgs = [synth.image(k, W, H) for k in range(8)]
rows = [s.to_device(s.filter(im.tobytes(), W, H, 8, 4, False)) for im in imgs]
for i in range(N): src[i * U:(i + 1) * U] = rows[i % 8]
want0 = s.to_device(imgs[0].reshape(-1))
descs = [s.image_desc(None, src[i * U:(i + 1) * U], out[i * S:(i + 1) * S], W, H, 8, 4, False, rows_cap=U) for i in range(N)]
for pr in (0, 128, 256, 512, 1024, 2048, 4096):
    s.configure(spng.CFG_UNFILTER_PIECE_ROWS, pr)
    s.unfilter_batch(descs); torch.cuda.synchronize()
    assert torch.equal(out[:S], want0)
    s.profile(True)
    for _ in range(5): s.unfilter_batch(descs)
    torch.cuda.synchronize()
    ms, n = s.profile_get(spng.K_UNFILTER); s.profile(False)
    print(f"{os.environ.get('SPNG_LIB', 'default').split('/')[-1]} piece_rows={pr}: {ms/n:.3f} ms -> {N*(U+S)/(ms/n*1e-3)/8e12*100:.2f}% of 8 TB/s", flush=True)
