#!/bin/bash
# Round-end measurement on the GPU box: parity suite, PMC traffic passes, headline bench, kernel-trace stats.
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -q > gpurun_out/r01_pytest_gpu.log 2>&1; tail -2 gpurun_out/r01_pytest_gpu.log
P="python bench.py --steps 1 --warmup 0 --images 64 --unique 8 --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- $P > /dev/null 2> gpurun_out/prof_fetch.err
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- $P > /dev/null 2> gpurun_out/prof_write.err
python tools/pmc_traffic.py gpurun_out/prof_fetch gpurun_out/prof_write 64 profiles/r01_pmc_traffic.json > gpurun_out/pmc.log 2>&1; cp profiles/r01_pmc_traffic.json gpurun_out/; tail -5 gpurun_out/pmc.log
timeout 400 python bench.py > gpurun_out/r01_bench_n1.json 2> gpurun_out/bench.err; cat gpurun_out/r01_bench_n1.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python bench.py --steps 1 --warmup 0 --images 256 --unique 8 --no-cpu-baseline > gpurun_out/r01_bench_under_rocprof_256img.json 2> gpurun_out/prof_stats.err
find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r01_rocprof_kernel_stats_256img.csv; head -6 gpurun_out/r01_rocprof_kernel_stats_256img.csv
rm -rf gpurun_out/prof_fetch/*/*.db gpurun_out/prof_write/*/*.db 2>/dev/null; du -sh gpurun_out | tail -1
