"""Probe: one call of 128 swift-png-made 4K images (the N = 8 shard of the headline batch) -- wall time against the stage times;
run under `rocprofv3 --kernel-trace --stats` it shows which kernels the difference is.

    python tools/probe_shard128.py [--kind swiftpng] [--n 128]
"""
import argparse, json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="swiftpng")
ap.add_argument("--n", type=int, default=128)
ap.add_argument("--steps", type=int, default=3)
args = ap.parse_args()
import torch
import swift_png_amd as spng
s = spng.load(0)
images, rows, streams = bench.build_inputs(s, 8, 32, args.kind)
d_streams = [s.to_device(z) for z in streams]
s.trim()
job = bench.DecodeJob(spng, s, torch, d_streams, args.n, 0, 8, 1)
for _ in range(2):
    job.decode_group(0)
torch.cuda.synchronize()
s.profile(True)
ts = []
for _ in range(args.steps):
    t0 = time.perf_counter(); job.decode_group(0); torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
prof = {k: round(s.profile_get(getattr(spng, "K_" + k.upper()))[0] / args.steps, 2) for k in bench.STAGES}
res = job.results()
print(json.dumps({"n": args.n, "kind": args.kind, "ms_per_call": ts, "stages_ms": prof, "pipeline_streams": sum(r.reserved == 1 for r in res),
                  "ok": all(r.status == 0 and r.written == job.U for r in res)}))
