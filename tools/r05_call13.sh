#!/bin/bash
# round 5, thirteenth GPU call: the marker parts of a small batch on 4 KiB tiles (79 KB of LDS: two workgroups per CU) against 8 KiB
# tiles (93 KB: one), over the number of parts a batch is cut into; parity subset on the shipped build (4 KiB) first
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 700 python -m pytest tests/test_gpu_pinflate.py tests/test_gpu_decode.py tests/test_gpu_resume.py tests/test_gpu_gzip.py -m gpu -q -x > gpurun_out/r05o_pytest_gpu.log 2>&1; tail -2 gpurun_out/r05o_pytest_gpu.log
SPNG_LIB=/root/repo/variants/libspng_f_mark8k.so timeout 300 python tools/probe_groups.py --kind swiftpng --unique 4 --shapes 128x1,128x2,32x1,8x1,1x1 --parts-total 0,512 > gpurun_out/r05o_probe_groups_mark8k.log 2>&1
echo "== 8 KiB tiles"; grep -E "^[0-9]+ images" gpurun_out/r05o_probe_groups_mark8k.log | cut -c1-200
SPNG_LIB=/root/repo/variants/libspng_f_mark4k.so timeout 300 python tools/probe_groups.py --kind swiftpng --unique 4 --shapes 128x1,128x2,32x1,8x1,1x1 --parts-total 0,512,1024 > gpurun_out/r05o_probe_groups_mark4k.log 2>&1
echo "== 4 KiB tiles"; grep -E "^[0-9]+ images" gpurun_out/r05o_probe_groups_mark4k.log | cut -c1-200
