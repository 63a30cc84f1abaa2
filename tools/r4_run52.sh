timeout 1200 python -m pytest tests/test_storage_bf16_gpu.py tests/test_kernels_gpu.py tests/test_network_gpu.py tests/test_mixed_precision_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -5
run() { python bench.py "$@" --steps 10 --warmup 3 --no-also --no-cpu-baseline --no-traffic --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2; do
echo -n "task009 fp32: "; run
echo -n "task100 fp32: "; run --workload task100
echo -n "resenc fp32: "; run --workload resenc
echo -n "task009 mixed: "; run --precision bf16
echo -n "task100 mixed: "; run --workload task100 --precision bf16
echo -n "resenc mixed: "; run --workload resenc --precision bf16
done
