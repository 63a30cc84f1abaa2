mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_storage_bf16_gpu.py tests/test_kernels_gpu.py -x -q > gpurun_out/r4/run17_tests.log 2>&1; tail -3 gpurun_out/r4/run17_tests.log
for v in libmtseg_hip.so libmtseg_hip_bfabl1.so libmtseg_hip_bfabl4.so libmtseg_hip_bfabl5.so libmtseg_hip_bfabl8.so; do
  MT_LIB_VARIANT=$v python bench.py --workload task009 --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r4/run17_$v.json 2>/dev/null
  python - "$v" <<'PY'
import json,sys
v=sys.argv[1]
try:
    d=json.loads(open('gpurun_out/r4/run17_%s.json'%v).read().strip().splitlines()[-1])
    bk=d['roofline']['all_conv_launches']['by_kernel_ms_per_step']
    print(v,d['ms_per_step'], {k[:50]:x for k,x in bk.items() if 'conv_bf16' in k or 'stem' in k})
except Exception as e: print(v,'failed',e)
PY
done
