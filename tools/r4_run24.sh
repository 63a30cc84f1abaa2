mkdir -p gpurun_out/r4
for v in libmtseg_hip.so libmtseg_hip_pabl1.so libmtseg_hip_pabl4.so libmtseg_hip_pabl5.so libmtseg_hip_pabl8.so; do
  MT_LIB_VARIANT=$v python bench.py --workload task009 --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r4/run24_$v.json 2>/dev/null
  python - "$v" <<'PY'
import json,sys
v=sys.argv[1]
try:
    d=json.loads(open('gpurun_out/r4/run24_%s.json'%v).read().strip().splitlines()[-1])
    bk=d['roofline']['all_conv_launches']['by_kernel_ms_per_step']
    print(v,d['ms_per_step'], {k[:44]:x for k,x in bk.items() if 'conv_bf16' in k})
except Exception as e: print(v,'failed',e)
PY
done
