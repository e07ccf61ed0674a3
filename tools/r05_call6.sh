#!/bin/bash
# round 5, sixth GPU call: after the sweep of stale heads -- the profile build on compressible input, the deflate tests, the probes
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
SPNG_LIB=/root/repo/variants/libspng_d3prof.so PROBE_LEVEL=6 PROBE_WHICH=synth4k PROBE_N=4 timeout 300 python tools/probe_deflate2.py > gpurun_out/r05f_d3prof_l6.log 2>&1; grep -E "d3 prof" gpurun_out/r05f_d3prof_l6.log | sort -t: -k2 -n | awk '{print $13, $19, $21, $27, $29, $35, $38}' | sort -n | awk 'NR%40==0' | tail -12; grep -E "d3 parse prof" gpurun_out/r05f_d3prof_l6.log | awk 'NR%9<2' | head -12; grep "streams," gpurun_out/r05f_d3prof_l6.log
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_resume.py tests/test_gpu_gzip.py -m gpu -q -k "deflate or encode or mirror or push or gzip or Deflator or level" > gpurun_out/r05f_pytest_gpu.log 2>&1; tail -2 gpurun_out/r05f_pytest_gpu.log
PROBE_WHICH=random,photo PROBE_N=256 timeout 300 python tools/probe_deflate2.py > gpurun_out/r05f_probe_l9_256.log 2>&1; tail -2 gpurun_out/r05f_probe_l9_256.log
PROBE_WHICH=random PROBE_N=1024 timeout 300 python tools/probe_deflate2.py > gpurun_out/r05f_probe_l9_1024.log 2>&1; tail -1 gpurun_out/r05f_probe_l9_1024.log
PROBE_LEVEL=6 PROBE_WHICH=synth4k,photo PROBE_N=256 timeout 400 python tools/probe_deflate2.py > gpurun_out/r05f_probe_l6_256.log 2>&1; tail -2 gpurun_out/r05f_probe_l6_256.log
PROBE_LEVEL=6 PROBE_WHICH=synth4k PROBE_N=1024 timeout 600 python tools/probe_deflate2.py > gpurun_out/r05f_probe_l6_1024.log 2>&1; grep -E "streams," gpurun_out/r05f_probe_l6_1024.log
