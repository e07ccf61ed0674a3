cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/probe_v2.py --kinds swiftpng,zlib --steps 3 2>&1 | grep -E "^(swiftpng|zlib) " | cut -c1-400
timeout 300 python tools/probe_groups.py --kind swiftpng --shapes 128x1,8x1,1x1 2>&1 | grep "images in" | head -3 | cut -c1-300
SPNG_LIB=/root/repo/variants/libspng_prof.so timeout 200 python tools/probe_v2.py --kinds swiftpng --steps 1 > gpurun_out/r06_dprof.log 2>&1; grep -E "^resolve" gpurun_out/r06_dprof.log | sort | uniq -c | sort -rn | head -2 | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_pinflate.py -q -x 2>&1 | tail -2
