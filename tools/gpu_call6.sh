#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
F='amdgpu.ids\|Warning\|as_tensor'
timeout 900 python -m pytest tests -m gpu -x -q -k "deflate or encode or golden or mirror or gzip" > gpurun_out/r04_pytest_deflate.log 2>&1; tail -5 gpurun_out/r04_pytest_deflate.log
timeout 600 python tools/probe_encode.py 2>&1 | grep -v "$F" | tee gpurun_out/r04_probe_encode.log
timeout 900 python bench.py --mode encode --steps 1 --warmup 1 > gpurun_out/r04_bench_encode.json 2> gpurun_out/r04_bench_encode.err; head -c 3000 gpurun_out/r04_bench_encode.json; tail -5 gpurun_out/r04_bench_encode.err
