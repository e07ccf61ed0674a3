cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_resume.py tests/test_gpu_gzip.py tests/test_gpu_pinflate.py -q -x 2>&1 | tail -12
timeout 900 python -m pytest tests/test_gpu_decode.py -q -x -k "inflate or mirror or context or decode_errors" 2>&1 | tail -3
