mkdir -p gpurun_out/r4
timeout 1700 python -m pytest tests/test_golden_gpu.py tests/test_ddp_world2_gpu.py tests/test_trainer_gpu.py tests/test_fullsize_oracle_gpu.py -x -q > gpurun_out/r4/run36_tests.log 2>&1; tail -3 gpurun_out/r4/run36_tests.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "loss" 2>&1 | tail -2
for f in 1 0; do
for a in "--workload task100" "--workload task100 --precision bf16" "--workload resenc --precision bf16"; do
  MT_LOSS_SPARSE=$f python bench.py $a --steps 8 --warmup 2 --no-cpu-baseline --no-also --no-traffic --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sparse=$f', '$a', d['ms_per_step'], d['config'].get('final_loss'))"
done; done
