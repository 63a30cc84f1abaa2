mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_storage_bf16_gpu.py -x -q -k "gather" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_mixed_precision_gpu.py tests/test_network_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -2
for f in 1 0 1 0; do
for a in "--precision bf16" "--workload resenc --precision bf16" "--workload task100 --precision bf16"; do
  MT_GATHER_BF16=$f python bench.py $a --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-traffic --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gather_bf16=$f', '$a', d['ms_per_step'], d['config'].get('final_loss'))"
done; done
