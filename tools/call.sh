cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_decode.py -q -x -k "unfilter or pngsuite or fuzz_shapes" 2>&1 | tail -2
timeout 900 python bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline --legs scanline_formats,small_images > gpurun_out/r06u_scan.json 2> gpurun_out/r06u.err; python - <<PY
import json
d=json.load(open('gpurun_out/r06u_scan.json'))
print(json.dumps(d['scanline_formats'])[:1800])
print(d['small_images']['ms_per_step'], d['small_images']['kernels_ms'])
PY
tail -3 gpurun_out/r06u.err
