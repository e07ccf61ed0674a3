cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/probe_v2.py --kinds swiftpng,zlib --steps 3 > gpurun_out/r06l_probe_find.log 2>&1; grep -E "^(swiftpng|zlib) auto" gpurun_out/r06l_probe_find.log | cut -c1-200
timeout 400 python tools/probe_groups.py --kind zlib --unique 4 --shapes 128x1,32x1,1x1 > gpurun_out/r06l_probe_groups_zlib.log 2>&1; head -4 gpurun_out/r06l_probe_groups_zlib.log | cut -c1-220
timeout 900 python -m pytest tests/test_gpu_pinflate.py tests/test_gpu_resume.py tests/test_gpu_gzip.py -q -x 2>&1 | tail -2
