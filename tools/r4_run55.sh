timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_storage_bf16_gpu.py -x -q -k "bwd_weight or backward_weight" 2>&1 | tail -4
run() { python bench.py "$@" --steps 10 --warmup 3 --no-also --no-cpu-baseline --no-traffic --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2; do for f in 0 1; do
echo -n "task009 fp32 MT_BWDW_VEC4=$f: "; MT_BWDW_VEC4=$f run
echo -n "task100 fp32 MT_BWDW_VEC4=$f: "; MT_BWDW_VEC4=$f run --workload task100
echo -n "resenc fp32 MT_BWDW_VEC4=$f: "; MT_BWDW_VEC4=$f run --workload resenc
done; done
