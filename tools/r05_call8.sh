#!/bin/bash
# round 5, eighth GPU call: the walk that only picks (terms of 128 positions worked out at once, literal runs as one store), leaner LDS
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_resume.py tests/test_gpu_gzip.py -m gpu -q -k "deflate or encode or mirror or push or gzip or Deflator or level" > gpurun_out/r05h_pytest_gpu.log 2>&1; tail -2 gpurun_out/r05h_pytest_gpu.log
PROBE_LEVEL=6 PROBE_WHICH=synth4k,photo PROBE_N=256 timeout 400 python tools/probe_deflate2.py > gpurun_out/r05h_probe_l6_256.log 2>&1; tail -2 gpurun_out/r05h_probe_l6_256.log
PROBE_LEVEL=6 PROBE_WHICH=synth4k,random PROBE_N=1024 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_l6 -- python tools/probe_deflate2.py > gpurun_out/r05h_probe_l6_1024.log 2>&1; grep -E "streams," gpurun_out/r05h_probe_l6_1024.log
find gpurun_out/prof_l6 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05h_rocprof_l6_kernel_stats.csv; head -6 gpurun_out/r05h_rocprof_l6_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/prof_l6
PROBE_LEVEL=6 PROBE_WHICH=synth4k PROBE_N=32 timeout 300 python tools/probe_deflate2.py > gpurun_out/r05h_probe_l6_32.log 2>&1; tail -1 gpurun_out/r05h_probe_l6_32.log
PROBE_WHICH=random,photo PROBE_N=256 timeout 300 python tools/probe_deflate2.py > gpurun_out/r05h_probe_l9_256.log 2>&1; tail -2 gpurun_out/r05h_probe_l9_256.log
