set -x
timeout 600 python -m pytest tests/test_gpu_decode.py -x -q -k "deflate or encode or mirror" 2>&1 | tail -12
timeout 600 python tools/probe_encode.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_probe_encode3.log
