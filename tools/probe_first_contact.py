"""First-contact GPU probe: timing of unfilter and inflate on a few 4K images."""
import sys; sys.path.insert(0, ".")
import sys, time, zlib, ctypes
import numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import swift_png_amd as spng
from swift_png_amd import synth

s = spng.load(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W = H = 4096
U = spng.inflated_size(W, H, 8, 4, False)
t0 = time.time()
img = synth.image(0, W, H)
print("synth s", time.time() - t0)
t0 = time.time(); rows = s.filter(img.tobytes(), W, H, 8, 4, False); print("gpu filter (host roundtrip) s", time.time() - t0)
hist = np.bincount(np.frombuffer(rows, np.uint8).reshape(H, W * 4 + 1)[:, 0], minlength=5); print("filter hist", hist)
t0 = time.time(); z = zlib.compress(rows, 6); print("zlib6 s", time.time() - t0, "ratio", len(rows) / len(z))
drows = s.to_device(rows)
dz = s.to_device(z)
# unfilter batch: N images sharing the same input rows (read-only), own outputs
outs = [s.empty(W * H * 4) for _ in range(N)]
for forced in (None, 0, 1, 2, 3, 4):
    if forced is None:
        src = drows
    else:
        r = np.frombuffer(rows, np.uint8).reshape(H, W * 4 + 1).copy(); r[:, 0] = forced
        src = s.to_device(r.tobytes())
    descs = [s.image_desc(None, src, o, W, H, 8, 4, False, rows_cap=U) for o in outs]
    s.unfilter_batch(descs)
    torch.cuda.synchronize()
    s.profile(True)
    t0 = time.time()
    for _ in range(3): s.unfilter_batch(descs)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    ms, n = s.profile_get(spng.K_UNFILTER); s.profile(False)
    print(f"unfilter forced={forced} N={N}: wall {dt*1e3:.2f} ms  kernel {ms/n:.2f} ms  -> {N*(U+W*H*4)/(ms/n*1e-3)/1e9:.1f} GB/s")
if forced is not None:
    pass
ok = bytes(outs[0].cpu().numpy()) != b""
# inflate batch
rowsb = [s.empty(U + 4096) for _ in range(N)]
sd = (spng.StreamDesc * N)(*[spng.StreamDesc(dz.data_ptr(), dz.numel(), r.data_ptr(), U + 4096, 0, 0) for r in rowsb])
res = (spng.Result * N)()
s.lib.spng_inflate_batch(s.ctx, sd, N, None, res)
s.profile(True)
t0 = time.time()
s.lib.spng_inflate_batch(s.ctx, sd, N, None, res)
dt = time.time() - t0
ms, n = s.profile_get(spng.K_INFLATE); s.profile(False)
print(f"inflate N={N}: wall {dt*1e3:.1f} ms kernel {ms/n:.1f} ms status {res[0].status} written {res[0].written} -> {N*U/(ms/n*1e-3)/1e9:.2f} GB/s out, per-stream {U/(ms/n*1e-3)/1e6:.1f} MB/s")
assert bytes(rowsb[N-1][:U].cpu().numpy()) == rows
print("inflate output matches")
