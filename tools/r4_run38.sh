timeout 900 python -m pytest tests/test_storage_bf16_gpu.py tests/test_kernels_gpu.py -x -q -k "stem" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_network_gpu.py -x -q 2>&1 | tail -2
for i in 1 2; do python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-also --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32', d['ms_per_step'])"; done
bash tools/profile_r4.sh task009_fp32 > /dev/null 2>&1; grep "stem" gpurun_out/prof_r4/task009_fp32_kernel_stats.csv | cut -c1-110
