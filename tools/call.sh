cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_decode.py -q -x -k "filter or encode" 2>&1 | tail -2
for v in band noband; do
[ $v = noband ] && export SPNG_EXP_NO_BAND=1 || unset SPNG_EXP_NO_BAND
timeout 900 python bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline --legs scanline_formats > gpurun_out/r06t_scan_$v.json 2> gpurun_out/r06t.err; python - <<PY
import json
d=json.load(open('gpurun_out/r06t_scan_$v.json'))
print('$v', {k:(v['filter']['ms'], v['filter']['frac_of_hbm_peak'], v['unfilter']['frac_of_hbm_peak']) for k,v in d['scanline_formats'].items()} if 'error' not in d['scanline_formats'] else d['scanline_formats'])
PY
timeout 600 python bench.py --mode encode --images 256 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r06t_enc256_$v.json 2>> gpurun_out/r06t.err; python - <<PY
import json
d=json.load(open('gpurun_out/r06t_enc256_$v.json'))
print('$v', json.dumps(d['kernels']['filter']))
PY
done
