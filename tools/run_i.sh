timeout 200 python tools/probe_unfilter_pieces.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_probe_unfilter_pieces.log
SPNG_LIB=$PWD/variants/libspng_nw4p32.so timeout 200 python tools/probe_unfilter_pieces.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02_probe_unfilter_pieces.log
