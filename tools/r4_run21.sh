mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_storage_bf16_gpu.py tests/test_fullsize_infer_gpu.py tests/test_golden_gpu.py tests/test_mixed_precision_gpu.py -x -q > gpurun_out/r4/run21_tests.log 2>&1; tail -5 gpurun_out/r4/run21_tests.log
