#!/bin/bash
# round 5, third GPU call: parity suite; deflate probes at levels 9 and 6 after the byte-at-the-best-length test, the 121 KB search
# workgroup and the two-wave parse of levels 0-7.
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r05c_pytest_gpu.log 2>&1; tail -4 gpurun_out/r05c_pytest_gpu.log
PROBE_WHICH=random,photo PROBE_N=256 timeout 300 python tools/probe_deflate2.py > gpurun_out/r05c_probe_l9_256.log 2>&1; tail -2 gpurun_out/r05c_probe_l9_256.log
PROBE_WHICH=random PROBE_N=1024 timeout 300 python tools/probe_deflate2.py > gpurun_out/r05c_probe_l9_1024.log 2>&1; tail -1 gpurun_out/r05c_probe_l9_1024.log
PROBE_LEVEL=6 PROBE_WHICH=synth4k,photo PROBE_N=256 timeout 400 python tools/probe_deflate2.py > gpurun_out/r05c_probe_l6_256.log 2>&1; tail -2 gpurun_out/r05c_probe_l6_256.log
PROBE_LEVEL=6 PROBE_WHICH=synth4k,random PROBE_N=1024 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_l6 -- python tools/probe_deflate2.py > gpurun_out/r05c_probe_l6_1024.log 2>&1; grep -E "streams," gpurun_out/r05c_probe_l6_1024.log
find gpurun_out/prof_l6 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05c_rocprof_l6_kernel_stats.csv; head -4 gpurun_out/r05c_rocprof_l6_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/prof_l6
