set -x
timeout 600 python -m pytest tests/test_gpu_decode.py -x -q -k "unfilter or config5 or pngsuite or decode_batch or filter_defilter" 2>&1 | tail -6
timeout 300 python tools/probe_config5.py 2>&1 | grep -v amdgpu.ids | tail -12
