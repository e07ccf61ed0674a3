"""Probe: batches of very different sizes through one context, back to back (1 image -> 1024 -> 8 -> 1024 -> 1): the scratch of
the multi-workgroup resolve is allocated, released by the large batch and allocated again; every batch bit-exact."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    import torch
    import swift_png_amd as spng
    s = spng.load(0)
    images, rows, streams = bench.build_inputs(s, 4, 32, "zlib")
    d_streams = [s.to_device(z) for z in streams]
    ref = [s.to_device(img.reshape(-1)) for img in images]
    for n in (1, 1024, 8, 1024, 1, 400, 3):
        job = bench.DecodeJob(spng, s, torch, d_streams, n, 0, 4, 1)
        t0 = time.perf_counter()
        job.decode_group(0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res = job.results()
        assert all(r.status == 0 and r.written == job.U and r.reserved == 1 for r in res), n
        assert all(torch.equal(job.d_out[j * job.S:(j + 1) * job.S], ref[job.src[j]]) for j in range(0, n, max(1, n // 16))), n
        print(f"{n} images: {dt * 1e3:.1f} ms, bit-exact", flush=True)
        del job
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
