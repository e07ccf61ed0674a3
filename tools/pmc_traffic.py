"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/r01_pmc_traffic.json.

    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <images> <out.json>

Corrections as MI355X_MICROARCH.md prescribes for gfx950: both counters are in KiB; FETCH_SIZE reports
half of the bytes of wide (16 B/lane) coalesced reads, so corrected fetch = 2 x raw."""
import csv, glob, json, sys

def collect(d, counter):
    out = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter: continue
            k = row["Kernel_Name"]
            key = "inflate" if "inflate_kernel" in k else "unfilter" if "unfilter_kernel" in k else None
            if key is None: continue
            e = out.setdefault(key, {"value": 0.0, "VGPR_Count": row.get("VGPR_Count") or row.get("Arch_VGPR_Count"), "LDS_Block_Size": row.get("LDS_Block_Size")})
            e["value"] += float(row["Counter_Value"])
    return out

fetch, write, images, dst = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3]), sys.argv[4]
W = H = 4096
U, S = H * (W * 4 + 1), W * H * 4
doc = {"note": "rocprofv3 --kernel-trace --pmc <counter> -- python bench.py --steps 1 --warmup 0 --images %d --unique 8 "
               "--no-cpu-baseline; one counter per pass. Units KiB. Per MI355X_MICROARCH.md FETCH_SIZE reports 1/2 of the "
               "bytes of wide (16 B/lane) coalesced reads on gfx950: corrected = 2 x raw; WRITE_SIZE is exact "
               "(the inflate kernel's aligned 16 B/lane flushes write images x %d B)." % (images, U),
       "images": images, "kernels": {}}
for k in ("inflate", "unfilter"):
    f, w = fetch.get(k, {}).get("value"), write.get(k, {}).get("value")
    if f is None or w is None: continue
    total = int(2 * f * 1024 + w * 1024)
    e = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "VGPR_Count": fetch[k]["VGPR_Count"], "LDS_Block_Size": fetch[k]["LDS_Block_Size"],
         "hbm_bytes_corrected": total, "hbm_bytes_per_image": total // images}
    if k == "unfilter":
        e["algorithmic_bytes"] = images * (U + S)
        e["traffic_over_algorithmic"] = round(total / (images * (U + S)), 3)
    doc["kernels"][k] = e
json.dump(doc, open(dst, "w"), indent=1)
print(json.dumps(doc["kernels"], indent=1))
