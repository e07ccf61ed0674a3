set -x
python -m pytest tests/test_gpu_pinflate.py -x -q 2>&1 | tail -15
python -m pytest tests/test_gpu_decode.py -x -q -k "inflate or config5 or pngsuite" 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 --no-swiftpng --no-cpu-baseline > gpurun_out/r02_c_bench.json 2> gpurun_out/r02_c_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r02_c_bench.json')); print(d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()}, d['config'])"
