"""Probe: the inflate pipeline (pinflate2.hip: find / decode / resolve) on the bench workload, per stage, for zlib-made
and swift-png-made level-6 streams, over segment sizes; with a -DSPNG_D_PROF build (SPNG_LIB=...) the kernels print their
phase cycle counters.

    python tools/probe_v2.py [--images 1024] [--unique 8] [--steps 3] [--kinds zlib,swiftpng] [--segments 0,900000]
"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=1024)
    ap.add_argument("--unique", type=int, default=8)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--kinds", default="zlib,swiftpng")
    ap.add_argument("--modes", default="auto")
    ap.add_argument("--overlap", default="auto", help="comma list of auto,always,never (SPNG_CFG_INFLATE_OVERLAP)")
    ap.add_argument("--segments", default="0", help="comma list of SPNG_CFG_SEGMENT_BYTES values to try (auto mode)")
    args = ap.parse_args()
    import torch
    import swift_png_amd as spng
    s = spng.load(0)
    out = {}
    for kind in args.kinds.split(","):
        t0 = time.time()
        images, rows, streams = bench.build_inputs(s, args.unique, 32, kind)
        d_streams = [s.to_device(z) for z in streams]
        print(f"[{kind}] inputs in {time.time() - t0:.1f} s, ratio {sum(len(r) for r in rows) / sum(len(z) for z in streams):.3f}", flush=True)
        ref = [s.to_device(img.reshape(-1)) for img in images]
        for mode, segb, ovl in [(m, int(sb), o) for m in args.modes.split(",") for sb in (args.segments.split(",") if m == "auto" else ["0"])
                                for o in args.overlap.split(",")]:
            s.configure(spng.CFG_INFLATE_MODE, {"auto": spng.INFLATE_AUTO, "serial": spng.INFLATE_SERIAL}[mode])
            s.configure(spng.CFG_INFLATE_OVERLAP, {"auto": spng.OVERLAP_AUTO, "always": spng.OVERLAP_ALWAYS, "never": spng.OVERLAP_NEVER}[ovl])
            s.configure(spng.CFG_SEGMENT_BYTES, segb)
            job = bench.DecodeJob(spng, s, torch, d_streams, args.images, 0, args.unique, 1)
            for _ in range(2):
                job.decode_group(0)
            torch.cuda.synchronize()
            s.profile(True)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                job.decode_group(0)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            prof = {k: round(s.profile_get(getattr(spng, "K_" + k.upper()))[0] / args.steps, 2) for k in bench.STAGES}
            s.profile(False)
            res = job.results()
            fast = sum(r.reserved == 1 for r in res)
            bad = [r.status for r in res if r.status or r.written != job.U][:4]
            okay = all(torch.equal(job.d_out[j * job.S:(j + 1) * job.S], ref[job.src[j]]) for j in range(0, args.images, max(1, args.images // 64)))
            line = {"ms_per_step": round(dt * 1e3, 2), "stages": prof, "pipeline_streams": fast, "bad": bad, "bit_exact": okay}
            out[f"{kind}/{mode}/{segb}/{ovl}"] = line
            print(kind, mode, segb, ovl, json.dumps(line), flush=True)
            del job
            torch.cuda.empty_cache()
        s.configure(spng.CFG_INFLATE_MODE, spng.INFLATE_AUTO)
        del d_streams, ref
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
