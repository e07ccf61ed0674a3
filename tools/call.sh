cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_files.py tests/test_gpu_gzip.py -q -x 2>&1 | tail -2
timeout 900 python bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline --legs small_images,file_to_pixels > gpurun_out/r06o_bench_small.json 2> gpurun_out/r06o.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r06o_bench_small.json'))
print(json.dumps(d.get('small_images'))[-700:])
print(json.dumps(d.get('file_to_pixels'))[:600])
PY
