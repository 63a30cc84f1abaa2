timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "loss" 2>&1 | tail -3
