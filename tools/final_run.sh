#!/bin/bash
# Round-end measurement on the GPU box: parity suite, smoke, headline bench (encode, configs[4] and file -> pixels legs included), kernel-trace stats and
# (PMC=1) the two PMC passes of the same workload.  Everything lands in gpurun_out/ (copied to profiles/ afterwards).
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
R=${R:-r03}
[ -n "$SKIP_TESTS" ] || timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/${R}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${R}_pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${R}_smoke.log 2>&1; tail -1 gpurun_out/${R}_smoke.log
timeout 900 python bench.py > gpurun_out/${R}_bench_n1.json 2> gpurun_out/${R}_bench.err; head -c 1500 gpurun_out/${R}_bench_n1.json; echo
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python bench.py --steps 3 --warmup 1 --no-swiftpng --no-cpu-baseline --no-extras > gpurun_out/${R}_bench_under_rocprof.json 2> gpurun_out/prof_stats.err
find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${R}_rocprof_kernel_stats.csv; head -8 gpurun_out/${R}_rocprof_kernel_stats.csv | cut -c1-160
if [ -n "$PMC" ]; then
P="python bench.py --steps 1 --warmup 0 --no-swiftpng --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- $P > /dev/null 2> gpurun_out/prof_fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- $P > /dev/null 2> gpurun_out/prof_write.err
python tools/pmc_traffic.py gpurun_out/prof_fetch gpurun_out/prof_write zlib 1024 32 gpurun_out/${R}_pmc_traffic.json > gpurun_out/pmc.log 2>&1; tail -30 gpurun_out/pmc.log
fi
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
