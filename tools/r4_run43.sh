mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_abi.py tests/test_storage_bf16_gpu.py -x -q > gpurun_out/r4/run43_a.log 2>&1; tail -5 gpurun_out/r4/run43_a.log
timeout 1200 python -m pytest tests/test_mixed_precision_gpu.py tests/test_network_gpu.py tests/test_golden_gpu.py tests/test_fullsize_infer_gpu.py -x -q > gpurun_out/r4/run43_b.log 2>&1; tail -3 gpurun_out/r4/run43_b.log
for f in 1 0 1 0; do
for a in "--precision bf16" "--workload resenc --precision bf16" "--workload task100 --precision bf16"; do
  MT_PW_M16=$f python bench.py $a --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-traffic --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('m16=$f', '$a', d['ms_per_step'], d['config'].get('final_loss'))"
done; done
