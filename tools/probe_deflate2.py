"""GPU probe: the two kernels of the level >= 8 rounds -- time of a batch, of its search launches and of its parse launches."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (runs from any directory: rocprofv3 wants /tmp)
import numpy as np, torch
import swift_png_amd as spng
from swift_png_amd import synth

s = spng.load(0)
level = int(os.environ.get("PROBE_LEVEL", "9"))
def run(name, tensors):
    outs, res = s.deflate_batch(tensors, level)          # warm-up: slab allocation
    del outs
    torch.cuda.synchronize(); s.profile(True); t0 = time.perf_counter()
    outs, res = s.deflate_batch(tensors, level)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    a = s.profile_get(spng.K_DFL_SEARCH)[0]; b = s.profile_get(spng.K_DFL_PARSE)[0]; d = s.profile_get(spng.K_DEFLATE)[0]; s.profile(False)
    n = sum(t.numel() for t in tensors)
    print(f"{name}: {len(tensors)} streams, {n/1e6:.0f} MB in {dt:.3f} s = {n/dt/1e6:.1f} MB/s ({n/dt/1e6/len(tensors):.2f} MB/s per stream); "
          f"search {a:.0f} ms, parse {b:.0f} ms, deflate span {d:.0f} ms; ratio {n/sum(r.written for r in res):.3f}", flush=True)
    del outs
gen = torch.Generator(device=s.tdev); gen.manual_seed(7)
N = int(os.environ.get("PROBE_N", "256"))
which = os.environ.get("PROBE_WHICH", "random,photo,one")
if "distinct" in which:
    # 1024 streams in 1024 buffers of their own (what bench.py's encode leg hands over) instead of 8 buffers read 128 times each
    big = torch.randint(0, 256, (N * (64 << 20),), dtype=torch.uint8, device=s.tdev, generator=gen)
    run("random 64 MiB, a buffer per stream", [big[i * (64 << 20):(i + 1) * (64 << 20)] for i in range(N)])
    if os.environ.get("PROBE_SLAB"):
        s.configure(spng.CFG_DEFLATE_BYTES, int(os.environ["PROBE_SLAB"]) << 30)
        run(f"... with a slab of {os.environ['PROBE_SLAB']} GiB", [big[i * (64 << 20):(i + 1) * (64 << 20)] for i in range(N)])
        s.configure(spng.CFG_DEFLATE_BYTES, 0)
    del big
elif "random" in which:
    rnd = [torch.randint(0, 256, (64 << 20,), dtype=torch.uint8, device=s.tdev, generator=gen) for _ in range(8)]
    run("random 64 MiB", [rnd[i % 8] for i in range(N)])
    del rnd
if "synth4k" in which:
    # the filtered scanlines of the bench's synthetic 4096^2 RGBA8 images (BASELINE configs[1]'s rasters): what an encode at level 6 deflates
    U = 4096 * (4096 * 4 + 1)
    imgs = [s.to_device(synth.image(k, 4096, 4096).tobytes()) for k in range(4)]
    rows4k = []
    for im in imgs:
        r = s.empty(U)
        res = s.filter_batch([s.image_desc(None, r, im, 4096, 4096, 8, 4, False, rows_cap=U)])
        rows4k.append(r)
    torch.cuda.synchronize()
    run("synthetic 4096^2 rows", [rows4k[i % 4] for i in range(N)])
    del rows4k, imgs
if "bench_photo" in which:
    # the rasters of bench.py's encode_photographic leg (synth.image(100 + k)): heavier chains than the eight above
    ph = [s.to_device(s.filter(synth.image(100 + k, 1024, 1024).tobytes(), 1024, 1024, 8, 4, False)) for k in range(8)]
    run("bench photographic 1024^2 rows", [ph[i % 8] for i in range(N)])
    for k in range(8): run(f"bench photographic image {100 + k}", [ph[k]] * 32)
elif "photo" in which:
    ph = [s.to_device(s.filter(synth.image(k, 1024, 1024).tobytes(), 1024, 1024, 8, 4, False)) for k in range(8)]
    run("photographic 1024^2 rows", [ph[i % 8] for i in range(N)])
    if "one" in which: run("photographic, one stream", ph[:1])
