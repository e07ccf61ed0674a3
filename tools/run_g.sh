set -x
timeout 900 python -m pytest tests/test_gpu_resume.py -x -q 2>&1 | tail -25
timeout 600 python -m pytest tests/test_gpu_decode.py -x -q -k "mirror or inflate or pngsuite or config5" 2>&1 | tail -8
