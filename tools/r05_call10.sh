#!/bin/bash
# round 5, tenth GPU call: 2 KB output ring (15 block waves per CU); the levels 0-7 search with 8 K and with 2 K of lead beside them
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_resume.py tests/test_gpu_gzip.py -m gpu -q -k "deflate or encode or mirror or push or gzip or Deflator or level" > gpurun_out/r05j_pytest_gpu.log 2>&1; tail -2 gpurun_out/r05j_pytest_gpu.log
PROBE_LEVEL=6 PROBE_WHICH=synth4k PROBE_N=256 timeout 400 python tools/probe_deflate2.py > gpurun_out/r05j_probe_l6_256.log 2>&1; tail -1 gpurun_out/r05j_probe_l6_256.log
PROBE_LEVEL=6 PROBE_WHICH=synth4k,random PROBE_N=1024 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_l6 -- python tools/probe_deflate2.py > gpurun_out/r05j_probe_l6_1024.log 2>&1; grep -E "streams," gpurun_out/r05j_probe_l6_1024.log
find gpurun_out/prof_l6 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05j_rocprof_l6_kernel_stats.csv; head -5 gpurun_out/r05j_rocprof_l6_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/prof_l6
SPNG_LIB=/root/repo/variants/libspng_r34.so PROBE_LEVEL=6 PROBE_WHICH=synth4k PROBE_N=1024 timeout 600 python tools/probe_deflate2.py > gpurun_out/r05j_probe_l6_1024_r34.log 2>&1; grep -E "streams," gpurun_out/r05j_probe_l6_1024_r34.log
PROBE_WHICH=random,photo PROBE_N=256 timeout 300 python tools/probe_deflate2.py > gpurun_out/r05j_probe_l9_256.log 2>&1; tail -2 gpurun_out/r05j_probe_l9_256.log
PROBE_WHICH=random PROBE_N=1024 timeout 300 python tools/probe_deflate2.py > gpurun_out/r05j_probe_l9_1024.log 2>&1; tail -1 gpurun_out/r05j_probe_l9_1024.log
