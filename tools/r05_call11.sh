#!/bin/bash
# round 5, eleventh GPU call: the chain walk without per-lane branches (predicates + wave-uniform branches): parity subset, levels 6 and 9
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_resume.py tests/test_gpu_gzip.py -m gpu -q -x -k "deflate or encode or mirror or push or gzip or Deflator or level" > gpurun_out/r05m_pytest_gpu.log 2>&1; tail -2 gpurun_out/r05m_pytest_gpu.log
PROBE_LEVEL=6 PROBE_WHICH=synth4k,photo PROBE_N=256 timeout 400 python tools/probe_deflate2.py > gpurun_out/r05m_probe_l6_256.log 2>&1; tail -2 gpurun_out/r05m_probe_l6_256.log
PROBE_LEVEL=6 PROBE_WHICH=synth4k,random PROBE_N=1024 timeout 600 python tools/probe_deflate2.py > gpurun_out/r05m_probe_l6_1024.log 2>&1; grep -E "streams," gpurun_out/r05m_probe_l6_1024.log
PROBE_WHICH=random,photo PROBE_N=256 timeout 300 python tools/probe_deflate2.py > gpurun_out/r05m_probe_l9_256.log 2>&1; tail -2 gpurun_out/r05m_probe_l9_256.log
PROBE_WHICH=random PROBE_N=1024 timeout 300 python tools/probe_deflate2.py > gpurun_out/r05m_probe_l9_1024.log 2>&1; tail -1 gpurun_out/r05m_probe_l9_1024.log
