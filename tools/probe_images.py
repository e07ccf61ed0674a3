"""GPU probe: inflate time of each of the benchmark's distinct images (the batch runs as long as its slowest stream)."""
import sys; sys.path.insert(0, ".")
import zlib
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
import swift_png_amd as spng
from swift_png_amd import synth

s = spng.load(0)
U0 = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = 8
W = H = 4096
U = spng.inflated_size(W, H, 8, 4, False)
import os, pickle
CACHE = "/tmp/probe_images_%d.pkl" % U0
if os.path.exists(CACHE):
    rows, streams = pickle.load(open(CACHE, "rb"))
else:
    with ThreadPoolExecutor(32) as pool:
        images = list(pool.map(lambda k: synth.image(k, W, H, 4, 8), range(U0)))
        rows = [s.filter(img.tobytes(), W, H, 8, 4, False) for img in images]
        streams = list(pool.map(lambda r: zlib.compress(r, 6), rows))
    pickle.dump((rows, streams), open(CACHE, "wb"))
out = torch.empty(N * (U + 4096), dtype=torch.uint8, device=s.tdev)
for k, z in enumerate(streams if len(sys.argv) <= 3 else []):
    dz = s.to_device(z)
    sd = (spng.StreamDesc * N)(*[spng.StreamDesc(dz.data_ptr(), dz.numel(), out.data_ptr() + i * (U + 4096), U + 4096, 0, 0) for i in range(N)])
    res = (spng.Result * N)()
    s.profile(True)
    s.lib.spng_inflate_batch(s.ctx, sd, N, None, res)
    ms, n = s.profile_get(spng.K_INFLATE); s.profile(False)
    filt = np.bincount(np.frombuffer(rows[k], dtype=np.uint8)[::W * 4 + 1], minlength=5)
    print(f"image {k:2d}: ratio {U/len(z):5.2f} kernel {ms/n:7.1f} ms status {res[0].status} filters {filt.tolist()}")

# all of them together, the way bench.py runs them: slot i decodes image i mod U0
NM = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if NM:
    outm = torch.empty(NM * (U + 4096), dtype=torch.uint8, device=s.tdev)
    dzs = [s.to_device(z) for z in streams]
    for order in ("interleaved", "grouped"):
        pick = (lambda i: i % U0) if order == "interleaved" else (lambda i: i * U0 // NM)
        sd = (spng.StreamDesc * NM)(*[spng.StreamDesc(dzs[pick(i)].data_ptr(), dzs[pick(i)].numel(), outm.data_ptr() + i * (U + 4096), U + 4096, 0, 0) for i in range(NM)])
        res = (spng.Result * NM)()
        for rep in range(2):
            s.profile(True)
            s.lib.spng_inflate_batch(s.ctx, sd, NM, None, res)
            ms, n = s.profile_get(spng.K_INFLATE); s.profile(False)
        print(f"mixed {order} N={NM}: kernel {ms/n:.1f} ms, statuses ok={all(r.status == 0 for r in res)}")

if NM and len(sys.argv) > 4:
    # the same through spng_decode_batch, with the raster slab allocated as bench.py does
    S = spng.storage_size(W, H, 8, 4)
    d_out = torch.empty(NM * S, dtype=torch.uint8, device=s.tdev)
    descs = (spng.ImageDesc * NM)()
    for j in range(NM):
        z = dzs[j % U0]
        descs[j] = spng.ImageDesc(z.data_ptr(), z.numel(), outm.data_ptr() + j * (U + 4096), U + 4096, d_out.data_ptr() + j * S, W, H, 8, 4, 0, 0, 0)
    for rep in range(3):
        s.profile(True)
        s.decode_batch(descs, wait=False)
        torch.cuda.synchronize()
        a = s.profile_get(spng.K_INFLATE); b = s.profile_get(spng.K_UNFILTER); s.profile(False)
        print(f"decode_batch rep {rep}: inflate {a[0]/max(1,a[1]):.1f} ms unfilter {b[0]/max(1,b[1]):.1f} ms")
    sd = (spng.StreamDesc * NM)(*[spng.StreamDesc(dzs[i % U0].data_ptr(), dzs[i % U0].numel(), outm.data_ptr() + i * (U + 4096), U + 4096, 0, 0) for i in range(NM)])
    s.profile(True); s.lib.spng_inflate_batch(s.ctx, sd, NM, None, res); ms, n = s.profile_get(spng.K_INFLATE); s.profile(False)
    print(f"inflate_batch after: {ms/n:.1f} ms")
