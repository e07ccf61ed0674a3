#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
F='amdgpu.ids\|Warning\|as_tensor'
timeout 600 python tools/diag_deflate.py 2>&1 | grep -v "$F" | tee gpurun_out/r04_diag_deflate.log | tail -45
