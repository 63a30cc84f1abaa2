mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_storage_bf16_gpu.py -x -q -k "tiled_backward_weight" > gpurun_out/r4/run29_tests.log 2>&1; tail -5 gpurun_out/r4/run29_tests.log
for f in 1 0 1 0; do
  for w in task009 resenc task100; do
    MT_BWDW_MARCH16=$f python bench.py --workload $w --precision bf16 --steps 8 --warmup 2 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r4/run29_${w}_$f.json 2>/dev/null
    python - "$f" "$w" <<'PY'
import json,sys
v,w=sys.argv[1:3]
try:
    d=json.loads(open('gpurun_out/r4/run29_%s_%s.json'%(w,v)).read().strip().splitlines()[-1])
    bk=d['roofline']['all_conv_launches']['by_kernel_ms_per_step']
    print('march16=%s'%v,w,d['ms_per_step'],'loss',d['config'].get('final_loss'), {k[:48]:x for k,x in bk.items() if 'march16' in k or 'fast16' in k})
except Exception as e: print(v,w,'failed',e)
PY
  done
done
