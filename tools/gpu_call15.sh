#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "unfilter or config5 or pngsuite or context_push" 2>&1 | tail -2
timeout 900 python - <<'PY' 2>&1 | grep -v "amdgpu.ids\|Warning\|as_tensor"
import json, sys, argparse
sys.path.insert(0, ".")
import torch, bench, bench_encode
import swift_png_amd as spng
s = spng.load(0)
print("config5", json.dumps(bench.run_config5(torch, spng, s, 3))[:700])
s.trim(); torch.cuda.empty_cache()
print("photo", json.dumps(bench_encode.run_encode_photographic(torch, spng, s, 9, cpu=False))[:600])
PY
