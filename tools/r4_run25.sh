mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_storage_bf16_gpu.py tests/test_mixed_precision_gpu.py tests/test_network_gpu.py -x -q > gpurun_out/r4/run25_tests.log 2>&1; tail -3 gpurun_out/r4/run25_tests.log
for w in task009 resenc; do
    python bench.py --workload $w --precision bf16 --steps 8 --warmup 2 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r4/run25_${w}.json 2>/dev/null
    python - "$w" <<'PY'
import json,sys
w=sys.argv[1]
d=json.loads(open('gpurun_out/r4/run25_%s.json'%w).read().strip().splitlines()[-1])
bk=d['roofline']['all_conv_launches']['by_kernel_ms_per_step']
print(w,d['ms_per_step'],'loss',d['config'].get('final_loss'), {k[:44]:x for k,x in bk.items() if 'conv_bf16' in k})
PY
done
python bench.py --workload infer --mirror 0 --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-also 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('infer', d['value'], d['ms_per_step'])"
