python bench.py --precision bf16 --steps 4 --warmup 2 --no-also --no-cpu-baseline --no-traffic --no-roofline 2>&1 | tail -12
timeout 900 python -m pytest tests/test_storage_bf16_gpu.py tests/test_kernels_gpu.py -x -q -k "backward_weight or bwd_weight" 2>&1 | tail -12
