timeout 1500 python -m pytest tests/test_fullsize_oracle_gpu.py -x -q -s 2>&1 | grep -i "cos\|passed\|failed" | tail -8
