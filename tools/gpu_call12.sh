#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
F='amdgpu.ids\|Warning\|as_tensor'
timeout 600 python tools/diag_deflate.py 2>&1 | grep -v "$F" | grep -v ": ok" | tail -5
timeout 600 python tools/probe_deflate2.py 2>&1 | grep -v "$F" | tee gpurun_out/r04_probe_deflate2c.log
timeout 900 python -m pytest tests -m gpu -x -q -k "deflate or encode or golden" > gpurun_out/r04_pytest_deflate3.log 2>&1; tail -3 gpurun_out/r04_pytest_deflate3.log
