mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_network_gpu.py tests/test_mixed_precision_gpu.py -x -q > gpurun_out/r4/run34_tests.log 2>&1; tail -2 gpurun_out/r4/run34_tests.log
for t in 1 0 1 0; do
for a in "--precision bf16" "--workload resenc --precision bf16"; do
  MT_PACK_TILED=$t python bench.py $a --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-traffic --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiled=$t', '$a', d['ms_per_step'])"
done; done
bash tools/profile_r4.sh resenc_bf16 > /dev/null 2>&1; grep pack_weights gpurun_out/prof_r4/resenc_bf16_kernel_stats.csv | cut -c1-120
