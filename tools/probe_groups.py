"""Probe: the per-GPU shape of BASELINE configs[2] at N = 8 ("strong": 1024 images / 8 GPUs = 128 per rank) on ONE GPU --
128 images decoded as 1 x 128, 2 x 64 and 4 x 32 calls of spng_decode_batch (bench.py's --groups, the pipeline depth of the
decode / gather overlap), plus 8 and 32 images in one call: what a call costs when it holds few streams.

    python tools/probe_groups.py [--kind zlib]
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
    ap.add_argument("--kind", default="zlib")
    ap.add_argument("--unique", type=int, default=8)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--parts-total", default="0", help="comma list: workgroups that resolve a small batch side by side, all streams "
                    "together (SPNG_CFG_RESOLVE_PARTS = total / images per call); 0 = the planner's own choice")
    ap.add_argument("--shapes", default="128x1,128x2,128x4,32x1,8x1", help="images x calls")
    ap.add_argument("--segment-bytes", default="0", help="comma list of SPNG_CFG_SEGMENT_BYTES values (0 = the planner's own)")
    args = ap.parse_args()
    import torch
    import swift_png_amd as spng
    s = spng.load(0)
    images, rows, streams = bench.build_inputs(s, args.unique, 32, args.kind)
    d_streams = [s.to_device(z) for z in streams]
    s.trim()
    torch.cuda.empty_cache()
    out = {}
    shapes = [tuple(int(v) for v in sh.split("x")) for sh in args.shapes.split(",")]
    for segb, total, (n, groups) in [(int(sb), int(t), sh) for sb in args.segment_bytes.split(",") for t in args.parts_total.split(",") for sh in shapes]:
        s.configure(spng.CFG_SEGMENT_BYTES, segb)
        s.configure(spng.CFG_RESOLVE_PARTS, max(2, min(128, total // (n // groups))) if total else 0)
        job = bench.DecodeJob(spng, s, torch, d_streams, n, 0, args.unique, groups)
        for _ in range(2):
            for g in range(groups):
                job.decode_group(g)
        torch.cuda.synchronize()
        s.profile(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            for g in range(groups):
                job.decode_group(g)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        prof = {k: round(s.profile_get(getattr(spng, "K_" + k.upper()))[0] / args.steps, 2) for k in bench.STAGES}
        s.profile(False)
        res = job.results()
        assert all(r.status == 0 and r.written == job.U for r in res)
        line = {"ms": round(dt * 1e3, 2), "mpixels_per_s": round(n * bench.MPIX / dt, 1), "stages_ms": prof}
        tag = f"{n} images in {groups} call(s)" + (f", {total} parts in all" if total else "") + (f", segments of {segb}" if segb else "")
        out[tag] = line
        print(tag + ":", json.dumps(line), flush=True)
        del job
        torch.cuda.empty_cache()
    s.configure(spng.CFG_RESOLVE_PARTS, 0)
    s.configure(spng.CFG_SEGMENT_BYTES, 0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
