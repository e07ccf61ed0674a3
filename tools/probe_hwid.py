import sys; sys.path.insert(0, ".")
import zlib, numpy as np, torch
import swift_png_amd as spng
s = spng.load(0)
W = H = 1024
U = spng.inflated_size(W, H, 8, 4, False)
S = spng.storage_size(W, H, 8, 4)
rng = np.random.default_rng(0)
rows = (rng.integers(0, 8, U, dtype=np.uint8)).tobytes()
rows = bytearray(rows)
for y in range(H): rows[y * (W * 4 + 1)] = 1
z = zlib.compress(bytes(rows), 6)
N = 1024
dz = s.to_device(z)
out = torch.empty(N * (U + 4096), dtype=torch.uint8, device=s.tdev)
d_out = torch.empty(N * S, dtype=torch.uint8, device=s.tdev)
sd = (spng.StreamDesc * N)(*[spng.StreamDesc(dz.data_ptr(), dz.numel(), out.data_ptr() + i * (U + 4096), U + 4096, 0, 0) for i in range(N)])
res = (spng.Result * N)()
print("--- inflate before any unfilter"); sys.stdout.flush()
s.lib.spng_inflate_batch(s.ctx, sd, N, None, res); torch.cuda.synchronize()
descs = (spng.ImageDesc * N)()
for j in range(N):
    descs[j] = spng.ImageDesc(dz.data_ptr(), dz.numel(), out.data_ptr() + j * (U + 4096), U + 4096, d_out.data_ptr() + j * S, W, H, 8, 4, 0, 0, 0)
print("--- decode_batch"); sys.stdout.flush()
s.decode_batch(descs, wait=True)
print("--- inflate after"); sys.stdout.flush()
s.lib.spng_inflate_batch(s.ctx, sd, N, None, res); torch.cuda.synchronize()
