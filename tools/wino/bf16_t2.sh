timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_mixed_precision_gpu.py -m gpu -q -x -k "bf16 or strided or mixed" 2>&1 | tail -8
python bench.py --precision bf16 --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-200
python bench.py --precision bf16 --no-cpu-baseline --no-roofline --workload resenc --steps 8 --warmup 2 2>&1 | tail -1 | cut -c1-200
python bench.py --precision bf16 --no-cpu-baseline --no-roofline --workload task100 --steps 8 --warmup 2 2>&1 | tail -1| cut -c1-200
