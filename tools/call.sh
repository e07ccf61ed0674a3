cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pinflate.py tests/test_gpu_multi.py -q -x 2>&1 | tail -2
timeout 600 python tools/probe_groups.py --kind swiftpng --unique 4 --shapes 1x1,8x1,32x1,128x1,128x2 > gpurun_out/r06w_probe_groups_swiftpng.log 2>&1; grep "call(s)" gpurun_out/r06w_probe_groups_swiftpng.log | cut -c1-200
timeout 600 python tools/probe_groups.py --kind zlib --unique 4 --shapes 1x1,32x1,128x1 > gpurun_out/r06w_probe_groups_zlib.log 2>&1; grep "call(s)" gpurun_out/r06w_probe_groups_zlib.log | cut -c1-200
