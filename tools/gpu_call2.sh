#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 ./variants/probe_copy 128 > gpurun_out/r04_probe_copy2.log 2>&1; cat gpurun_out/r04_probe_copy2.log
timeout 900 python -m pytest tests -m gpu -x -q -k "single_push or retry_pass or vertex_cap or whole_stream" > gpurun_out/r04_pytest_subset2.log 2>&1; tail -5 gpurun_out/r04_pytest_subset2.log
