"""Sums rocprofv3 --pmc counter rows per (kernel, counter): python tools/pmc_kernels.py <dir> [<dir> ...] > out.json
(each directory = one `rocprofv3 --kernel-trace --pmc ... --output-format csv -d <dir>` pass; value = [sum, launches])"""
import csv, glob, json, re, sys

out = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void spng::", "")[:48]
            e = out.setdefault(f"{name} {row['Counter_Name']}", [0.0, 0])
            e[0] += float(row["Counter_Value"]); e[1] += 1
print(json.dumps(out, indent=1, sort_keys=True))
