mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_storage_bf16_gpu.py -x -q > gpurun_out/r4/run15_tests.log 2>&1; tail -5 gpurun_out/r4/run15_tests.log
for f in 1 0 1 0; do
  for w in "resenc" "task009"; do
    MT_BWDW_FAST16=$f python bench.py --workload $w --precision bf16 --steps 8 --warmup 2 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r4/run15_${w}_$f.json 2>/dev/null
    python - "$f" "$w" <<'PY'
import json,sys
v,w=sys.argv[1:3]
d=json.loads(open('gpurun_out/r4/run15_%s_%s.json'%(w,v)).read().strip().splitlines()[-1])
bk=d['roofline']['all_conv_launches']['by_kernel_ms_per_step']
print('fast16=%s'%v,w,d['ms_per_step'], 'final_loss', d['config'].get('final_loss'), {k[:52]:x for k,x in bk.items() if 'bwdw_fast' in k})
PY
  done
done
timeout 900 python -m pytest tests/test_fullsize_oracle_gpu.py -x -q -k "resenc" > gpurun_out/r4/run15_full.log 2>&1; tail -5 gpurun_out/r4/run15_full.log; grep -n "cos" gpurun_out/r4/run15_full.log | head
