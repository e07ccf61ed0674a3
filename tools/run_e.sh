set -x
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_pinflate.py -x -q -k "inflate or pipeline or pngsuite or deflate_full or deflate_vs or deflate_level9 or config5" 2>&1 | tail -6
timeout 300 python tools/probe_encode.py 2>&1 | grep -v amdgpu.ids | grep "level 9" | tee gpurun_out/r02_probe_encode4.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02_e_bench.json 2> gpurun_out/r02_e_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r02_e_bench.json')); print(d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()}); w=d['swiftpng_streams']; print(w['ms_per_step'], {k:v['ms_per_step'] for k,v in w['kernels'].items()})"
