python bench.py --workload infer --precision bf16 --mirror 0 --steps 2 --warmup 1 2>&1 | tail -1
python bench.py --workload infer --precision bf16 --mirror 1 --steps 1 --warmup 1 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
