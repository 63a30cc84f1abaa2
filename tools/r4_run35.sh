mkdir -p gpurun_out/r4
echo "== MT_BF16_STORAGE=0"; MT_BF16_STORAGE=0 timeout 900 python -m pytest tests/test_mixed_precision_gpu.py tests/test_network_gpu.py -x -q 2>&1 | tail -2
echo "== MT_ACT_STORAGE=bf16"; MT_ACT_STORAGE=bf16 timeout 900 python -m pytest tests/test_mixed_precision_gpu.py -x -q 2>&1 | tail -2
echo "== MT_BWDW_STAGED=0 MT_BWDW_GEMM=0 MT_BWDW_FAST16=0"; MT_BWDW_STAGED=0 MT_BWDW_GEMM=0 MT_BWDW_FAST16=0 timeout 900 python -m pytest tests/test_mixed_precision_gpu.py -x -q 2>&1 | tail -2
for w in task100 resenc; do for p in bf16; do
python bench.py --workload $w --precision $p --patch 96 192 192 --steps 5 --warmup 2 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r4/b_${w}_${p}_patch96.json 2>/dev/null
python -c "import json; d=json.loads(open('gpurun_out/r4/b_${w}_${p}_patch96.json').read().strip().splitlines()[-1]); print('$w $p patch96', d['value'], d['ms_per_step'])"
done; done
for s in 0; do MT_BF16_STORAGE=0 python bench.py --workload resenc --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-traffic --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resenc MT_BF16_STORAGE=0', d['ms_per_step'])"; done
