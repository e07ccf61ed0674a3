#!/bin/bash
# round 5, second GPU call: the whole parity suite (new search at every level, pack, overdraw, multi), deflate probes at levels 9 and 6,
# kernel-trace stats of the level-6 probe.
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r05b_pytest_gpu.log 2>&1; tail -15 gpurun_out/r05b_pytest_gpu.log
PROBE_WHICH=random,photo PROBE_N=256 timeout 300 python tools/probe_deflate2.py > gpurun_out/r05b_probe_l9_256.log 2>&1; tail -2 gpurun_out/r05b_probe_l9_256.log
PROBE_WHICH=random PROBE_N=1024 timeout 300 python tools/probe_deflate2.py > gpurun_out/r05b_probe_l9_1024.log 2>&1; tail -1 gpurun_out/r05b_probe_l9_1024.log
PROBE_LEVEL=6 PROBE_WHICH=synth4k,photo PROBE_N=256 timeout 400 python tools/probe_deflate2.py > gpurun_out/r05b_probe_l6_256.log 2>&1; tail -2 gpurun_out/r05b_probe_l6_256.log
PROBE_LEVEL=6 PROBE_WHICH=synth4k,random PROBE_N=1024 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_l6 -- python tools/probe_deflate2.py > gpurun_out/r05b_probe_l6_1024.log 2>&1; tail -2 gpurun_out/r05b_probe_l6_1024.log
find gpurun_out/prof_l6 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05b_rocprof_l6_kernel_stats.csv; head -8 gpurun_out/r05b_rocprof_l6_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/prof_l6
