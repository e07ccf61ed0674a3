cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/probe_groups.py --kind swiftpng --unique 4 --shapes 1x1,2x1,4x1,8x1,32x1 > gpurun_out/r06x_probe_parts128.log 2>&1; grep "call(s)" gpurun_out/r06x_probe_parts128.log | cut -c1-200
timeout 600 python bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline --legs config5,single_image > gpurun_out/r06x_cfg5.json 2>/dev/null; python - <<'PY'
import json
d=json.load(open('gpurun_out/r06x_cfg5.json'))
print(json.dumps(d['config5'])[-420:]); print(json.dumps(d['single_image'])[-300:])
PY
timeout 900 python -m pytest tests/test_gpu_pinflate.py tests/test_gpu_multi.py -q -x 2>&1 | tail -2
