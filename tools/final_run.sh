#!/bin/bash
# Round-end measurement on the GPU box: parity suite, smoke, headline bench (encode, configs[4] and file -> pixels legs included), kernel-trace stats and
# (PMC=1) the two PMC passes of the same workload + the counter calibration on the copy probe.  Everything lands in gpurun_out/ (copied to profiles/ afterwards).
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
R=${R:-r04}
[ -n "$SKIP_TESTS" ] || timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${R}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${R}_pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${R}_smoke.log 2>&1; tail -1 gpurun_out/${R}_smoke.log
[ -n "$SKIP_PROF" ] || { timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python bench.py --steps 3 --warmup 1 --no-swiftpng --no-cpu-baseline --no-extras > gpurun_out/${R}_bench_under_rocprof.json 2> gpurun_out/prof_stats.err
find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${R}_rocprof_kernel_stats.csv; head -8 gpurun_out/${R}_rocprof_kernel_stats.csv | cut -c1-160; }
if [ -n "$PMC" ]; then
P="python bench.py --steps 1 --warmup 0 --no-swiftpng --no-cpu-baseline --no-extras"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- $P > /dev/null 2> gpurun_out/prof_fetch.err
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- $P > /dev/null 2> gpurun_out/prof_write.err
python tools/pmc_traffic.py gpurun_out/prof_fetch gpurun_out/prof_write swiftpng 1024 32 gpurun_out/${R}_pmc_traffic.json > gpurun_out/pmc.log 2>&1; tail -30 gpurun_out/pmc.log
cp gpurun_out/${R}_pmc_traffic.json profiles/${R}_pmc_traffic.json   # (the bench line below reads it: traffic of this very build)
# counter calibration: known byte counts in the copy probe's patterns (plain 16-byte copy; 64-row tiles with and without skewed stores)
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/cal_fetch -- ./variants/probe_copy 128 pmc > gpurun_out/${R}_probe_copy_pmc.log 2> /dev/null
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/cal_write -- ./variants/probe_copy 128 pmc > /dev/null 2> /dev/null
python - <<'PY' > gpurun_out/${R}_pmc_calibration.txt 2>&1
import csv, glob
for d, c in (("gpurun_out/cal_fetch", "FETCH_SIZE"), ("gpurun_out/cal_write", "WRITE_SIZE")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == c:
                print(c, row["Kernel_Name"][:60], "grid", row.get("Grid_Size"), "KiB", row["Counter_Value"])
print("bytes moved per launch, each direction: 128 x 4096 x 16384 = 8589934592 (rows kernels: the source rows are pitch + 1 apart)")
PY
cat gpurun_out/${R}_pmc_calibration.txt | head -20
fi
# the headline line last: it carries the traffic measured above
[ -n "$SKIP_BENCH" ] || { timeout 1200 python bench.py > gpurun_out/${R}_bench_n1.json 2> gpurun_out/${R}_bench.err; head -c 1500 gpurun_out/${R}_bench_n1.json; echo; }
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/cal_fetch gpurun_out/cal_write
