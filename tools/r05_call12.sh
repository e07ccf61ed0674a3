#!/bin/bash
# round 5, twelfth GPU call: the inflate pipeline with fewer instructions per token / per byte -- A/B of three builds in one call
# (variants/libspng_a_head.so = 5d9ad04, b_marks = 2f62886: marks + replay addressing, d_all = the working tree: + alignbit,
# code-length LUT fields, biased positions, resolve with reciprocal table and v_mbcnt), parity subset on the working tree first
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 700 python -m pytest tests/test_gpu_pinflate.py tests/test_gpu_decode.py tests/test_gpu_resume.py tests/test_gpu_gzip.py -m gpu -q -x > gpurun_out/r05n_pytest_gpu.log 2>&1; tail -2 gpurun_out/r05n_pytest_gpu.log
for v in a_head b_marks d_all; do
  SPNG_LIB=/root/repo/variants/libspng_$v.so timeout 300 python tools/probe_v2.py --kinds swiftpng,zlib --steps 3 > gpurun_out/r05n_probe_v2_$v.log 2>&1
  echo "== $v"; grep -E "^(swiftpng|zlib) auto" gpurun_out/r05n_probe_v2_$v.log | cut -c1-260
done
# the shipped library once more (= d_all), with the 128-image shard shape
timeout 300 python tools/probe_groups.py --kind swiftpng --unique 4 > gpurun_out/r05n_probe_groups.log 2>&1; head -4 gpurun_out/r05n_probe_groups.log | cut -c1-240
