"""Picks the library of a tuning call's A/B legs (tools/final_run.sh, AB="name name ..."): the shipped build is the LAST name's; an
earlier one takes its place -- copied over swift_png_amd/libspng_mi355.so for the rest of the run -- only when it decodes the
headline workload more than 0.5 % faster, bit-exact.  Prints one JSON line.

    python tools/ab_pick.py <round tag> <name> [<name> ...]      (reads gpurun_out/<tag>_ab_<name>.log of tools/probe_v2.py)
"""
import json
import shutil
import sys

tag, names = sys.argv[1], sys.argv[2:]
ms = {}
for v in names:
    try:
        for ln in open(f"gpurun_out/{tag}_ab_{v}.log"):
            if ln.startswith("swiftpng auto"):
                d = json.loads(ln[ln.index("{"):])
                if d["bit_exact"] and not d["bad"]:
                    ms[v] = d["ms_per_step"]
    except OSError:
        pass
shipped = names[-1]
best = min(ms, key=ms.get) if ms else shipped
win = best if best != shipped and shipped in ms and ms[best] < 0.995 * ms[shipped] else shipped
print(json.dumps({"ms_per_step": ms, "shipped": shipped, "winner": win}))
if win != shipped:
    shutil.copy(f"variants/libspng_{win}.so", "swift_png_amd/libspng_mi355.so")
