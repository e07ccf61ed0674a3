cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_resume.py tests/test_gpu_gzip.py tests/test_gpu_pinflate.py -q -x -s 2>&1 | grep -E "one-block|passed|failed|Error" | tail -5
timeout 900 python -m pytest tests/test_gpu_decode.py -q -x -k "inflate or mirror or context or decode_errors or config1" 2>&1 | tail -2
