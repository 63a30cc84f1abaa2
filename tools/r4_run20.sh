mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_storage_bf16_gpu.py tests/test_fullsize_infer_gpu.py tests/test_sliding_window_gpu.py -x -q > gpurun_out/r4/run20_tests.log 2>&1; tail -5 gpurun_out/r4/run20_tests.log
python bench.py --workload infer --mirror 0 --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-also 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('infer nomirror mixed', d['value'], d['ms_per_step'])"
python bench.py --workload infer --mirror 1 --precision bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-traffic --no-also 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('infer mirror mixed', d['value'], d['ms_per_step'])"
