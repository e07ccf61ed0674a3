cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_decode.py -q -x -k "deflate or encode" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_resume.py -q -x -k "deflate" 2>&1 | tail -2
PROBE_LEVEL=6 PROBE_WHICH=synth4k,photo PROBE_N=256 timeout 300 python tools/probe_deflate2.py 2>&1 | tail -2 | cut -c1-250
PROBE_LEVEL=6 PROBE_WHICH=synth4k,random PROBE_N=1024 timeout 400 python tools/probe_deflate2.py 2>&1 | tail -2 | cut -c1-250
PROBE_LEVEL=1 PROBE_WHICH=synth4k PROBE_N=1024 timeout 400 python tools/probe_deflate2.py 2>&1 | tail -1 | cut -c1-250
SPNG_LIB=/root/repo/variants/libspng_d3prof.so PROBE_LEVEL=6 PROBE_WHICH=synth4k PROBE_N=64 timeout 300 python tools/probe_deflate2.py 2>&1 | grep "d3 prof" | grep -v "(4096 pos" | head -2 | cut -c1-330
