#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
F='amdgpu.ids\|Warning\|as_tensor'
timeout 600 python tools/diag_deflate.py 2>&1 | grep -v "$F" | grep -v ": ok" | tail -12
PROBE_N=8 PROBE_WHICH=random,photo SPNG_LIB=$PWD/variants/libspng_dprof.so timeout 600 python tools/probe_deflate2.py 2>&1 | grep -v "$F" | grep "round from 2095094\|streams" | tail -6 | cut -c1-250
timeout 600 python tools/probe_deflate2.py 2>&1 | grep -v "$F" | tee gpurun_out/r04_probe_deflate2b.log
timeout 900 python -m pytest tests -m gpu -x -q -k "deflate or encode or golden or mirror or gzip" > gpurun_out/r04_pytest_deflate2.log 2>&1; tail -3 gpurun_out/r04_pytest_deflate2.log
