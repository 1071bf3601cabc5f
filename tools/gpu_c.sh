export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_backward.py -q -m gpu -x -k "wide" 2>&1 | tail -5
