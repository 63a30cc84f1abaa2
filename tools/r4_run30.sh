mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_storage_bf16_gpu.py -x -q -k "bwdw_wino or wino" > gpurun_out/r4/run30_tests.log 2>&1; tail -3 gpurun_out/r4/run30_tests.log
for w in task009 resenc task100; do
    python bench.py --workload $w --precision bf16 --steps 8 --warmup 2 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r4/run30_${w}.json 2>/dev/null
    python - "$w" <<'PY'
import json,sys
w=sys.argv[1]
d=json.loads(open('gpurun_out/r4/run30_%s.json'%w).read().strip().splitlines()[-1])
bk=d['roofline']['all_conv_launches']['by_kernel_ms_per_step']
print(w,d['ms_per_step'],'loss',d['config'].get('final_loss'), {k[:44]:x for k,x in bk.items() if 'bwdw_wino' in k})
PY
done
