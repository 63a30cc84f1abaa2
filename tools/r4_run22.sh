mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_storage_bf16_gpu.py tests/test_mixed_precision_gpu.py tests/test_network_gpu.py tests/test_fullsize_properties_gpu.py -x -q > gpurun_out/r4/run22_tests.log 2>&1; tail -5 gpurun_out/r4/run22_tests.log
for f in 1 0 1 0; do
  for w in task009 resenc; do
    MT_BF16_PERSIST=$f python bench.py --workload $w --precision bf16 --steps 8 --warmup 2 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r4/run22_${w}_$f.json 2>/dev/null
    python - "$f" "$w" <<'PY'
import json,sys
v,w=sys.argv[1:3]
try:
    d=json.loads(open('gpurun_out/r4/run22_%s_%s.json'%(w,v)).read().strip().splitlines()[-1])
    bk=d['roofline']['all_conv_launches']['by_kernel_ms_per_step']
    print('persist=%s'%v,w,d['ms_per_step'],'loss',d['config'].get('final_loss'), {k[:44]:x for k,x in bk.items() if 'conv_bf16' in k})
except Exception as e: print(v,w,'failed',e)
PY
  done
done
for f in 1 0; do MT_BF16_PERSIST=$f python bench.py --workload infer --mirror 0 --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-also 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('infer persist=$f', d['value'], d['ms_per_step'])"; done
