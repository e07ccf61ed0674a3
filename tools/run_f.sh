set -x
timeout 600 python -m pytest tests/test_gpu_gzip.py tests/test_gpu_files.py tests/test_torch_free.py -x -q 2>&1 | tail -15
