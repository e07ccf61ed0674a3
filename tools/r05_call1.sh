#!/bin/bash
# round 5, first GPU call: the parity suite with the new search kernel / pack / overdraw, the deflate probes, PMC passes of the
# search, the 128-image shard shape with the new piece rows.
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r05a_pytest_gpu.log 2>&1; tail -5 gpurun_out/r05a_pytest_gpu.log
PROBE_WHICH=random,photo PROBE_N=256 timeout 300 python tools/probe_deflate2.py > gpurun_out/r05a_probe_deflate2_256.log 2>&1; cat gpurun_out/r05a_probe_deflate2_256.log | tail -4
PROBE_WHICH=random PROBE_N=1024 timeout 300 python tools/probe_deflate2.py > gpurun_out/r05a_probe_deflate2_1024.log 2>&1; cat gpurun_out/r05a_probe_deflate2_1024.log | tail -2
PROBE_WHICH=random,photo PROBE_N=64 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_a -- python tools/probe_deflate2.py > gpurun_out/r05a_pmc_a.log 2>&1
PROBE_WHICH=random,photo PROBE_N=64 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_b -- python tools/probe_deflate2.py > gpurun_out/r05a_pmc_b.log 2>&1
python tools/pmc_kernels.py gpurun_out/pmc_a gpurun_out/pmc_b > gpurun_out/r05a_pmc_encode.json 2> gpurun_out/pmc_kernels.err; grep -A2 "dfl" gpurun_out/r05a_pmc_encode.json | head -80
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b
timeout 400 python tools/probe_groups.py --kind zlib --unique 4 > gpurun_out/r05a_probe_groups.log 2>&1; head -5 gpurun_out/r05a_probe_groups.log
