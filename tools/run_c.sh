set -x
timeout 300 python -m pytest tests/test_gpu_pinflate.py -x -q 2>&1 | tail -8
SPNG_LIB=$PWD/swift_png_amd/libspng_prof.so timeout 300 python tools/probe_deflate_prof.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_probe_deflate_prof.log
