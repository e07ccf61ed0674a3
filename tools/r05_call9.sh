#!/bin/bash
# round 5, ninth GPU call: 8 K positions of lead in the levels 0-7 search
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_decode.py -m gpu -q -k "deflate or encode" > gpurun_out/r05i_pytest_gpu.log 2>&1; tail -2 gpurun_out/r05i_pytest_gpu.log
PROBE_LEVEL=6 PROBE_WHICH=synth4k,photo PROBE_N=256 timeout 400 python tools/probe_deflate2.py > gpurun_out/r05i_probe_l6_256.log 2>&1; tail -2 gpurun_out/r05i_probe_l6_256.log
PROBE_LEVEL=6 PROBE_WHICH=synth4k,random PROBE_N=1024 timeout 600 python tools/probe_deflate2.py > gpurun_out/r05i_probe_l6_1024.log 2>&1; grep -E "streams," gpurun_out/r05i_probe_l6_1024.log
PROBE_LEVEL=1 PROBE_WHICH=synth4k PROBE_N=1024 timeout 600 python tools/probe_deflate2.py > gpurun_out/r05i_probe_l1_1024.log 2>&1; grep -E "streams," gpurun_out/r05i_probe_l1_1024.log
