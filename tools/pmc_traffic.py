"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of one bench.py step into
profiles/r06_pmc_traffic.json (the file bench.py reads `roofline.traffic` / `kernels.*.traffic` from).

    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <kind> <images> <unique> <out.json> [<decode steps in the run>]

The passes run `python bench.py --steps 1 --warmup 0 --no-swiftpng --no-cpu-baseline --no-extras [--streams <kind>]`, i.e. the very
workload of the headline line, one decode step; every launch of a kernel inside that step is summed ("per launch" =
per step; the parallel inflate stages launch once per token-buffer pass).
Corrections as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: both counters are in KiB; FETCH_SIZE reports
half of the bytes of wide (16 B/lane) coalesced reads, so corrected fetch = 2 x raw for the kernels whose reads are
16 B/lane streams: every kernel of the decode step now is (decode stages its chunks and writes its tokens in 16-byte
units, resolve reads tokens and writes bytes in 16-byte units, unfilter as before); raw figures are kept next to them."""
import csv, glob, hashlib, json, os, sys
from pathlib import Path

NAMES = {"pinf2_find_kernel": "pinf_find", "pinf2_decode_kernel": "pinf_decode", "pinf2_resolve_kernel": "pinf_resolve",
         "::inflate_kernel": "inflate", "unfilter_kernel": "unfilter", "unfilter_pk_kernel": "unfilter"}


def collect(d, counter):
    out = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            key = next((v for k, v in NAMES.items() if k in row["Kernel_Name"]), None)
            if key is None:
                continue
            e = out.setdefault(key, {"value": 0.0, "launches": 0})
            e["value"] += float(row["Counter_Value"])
            e["launches"] += 1
    return out


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
kind, images, unique, dst = sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
# (round 5: the passes run a warm-up step and a timed one -- the first call of a context sizes its token pool differently --, so
#  the sums are over `nsteps` decode steps and the figures per step are the sums divided by it; `launches` likewise)
nsteps = int(sys.argv[7]) if len(sys.argv) > 7 else 1
for tab in (fetch, write):
    for e in tab.values():
        e["value"] /= nsteps; e["launches"] = round(e["launches"] / nsteps)
doc = json.load(open(dst)) if os.path.exists(dst) else {
    "note": "rocprofv3 --kernel-trace --pmc <counter> -- python bench.py --steps 1 --warmup 0 --no-swiftpng --no-cpu-baseline "
            "[--streams <kind>]; one counter per pass; units KiB; FETCH_SIZE x 2 (MI355X_MICROARCH.md, gfx950); summed over "
            "the launches of one decode step", "configs": {}}
cfg = {"images": images, "unique": unique, "kernels": {}}
for k in NAMES.values():
    f, w = fetch.get(k, {}).get("value"), write.get(k, {}).get("value")
    if f is None or w is None:
        continue
    cfg["kernels"][k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "launches": fetch[k]["launches"],
                         "hbm_bytes_raw": int((f + w) * 1024), "hbm_bytes_per_launch": int((2 * f + w) * 1024),
                         "hbm_bytes_per_step": int((2 * f + w) * 1024)}        # (per_launch: the older name of the same figure)
def source_digest():
    h = hashlib.sha256()
    root = Path(__file__).resolve().parent.parent / "swift_png_amd"
    for f in sorted(list((root / "csrc").glob("*.hip")) + list((root / "csrc").glob("*.hpp")) + list((root.parent / "include").glob("*.h"))):
        h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


cfg["source_digest"] = source_digest()        # (swift_png_amd.source_digest(): the build these counters were taken on)
doc["configs"][kind] = cfg
json.dump(doc, open(dst, "w"), indent=1)
print(json.dumps(cfg, indent=1))
