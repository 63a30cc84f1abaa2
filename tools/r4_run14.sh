mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_storage_bf16_gpu.py tests/test_mixed_precision_gpu.py tests/test_kernels_gpu.py tests/test_network_gpu.py -x -q > gpurun_out/r4/run14_tests.log 2>&1; tail -3 gpurun_out/r4/run14_tests.log
for v in libmtseg_hip_swz0.so libmtseg_hip.so libmtseg_hip_swz0.so libmtseg_hip.so; do
  for w in "resenc" "task009"; do
    MT_LIB_VARIANT=$v python bench.py --workload $w --precision bf16 --steps 8 --warmup 2 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r4/run14_${w}_$v.json 2>/dev/null
    python - "$v" "$w" <<'PY'
import json,sys
v,w=sys.argv[1:3]
d=json.loads(open('gpurun_out/r4/run14_%s_%s.json'%(w,v)).read().strip().splitlines()[-1])
bk=d['roofline']['all_conv_launches']['by_kernel_ms_per_step']
print(v,w,d['ms_per_step'], {k[:46]:x for k,x in bk.items() if 'conv_bf16' in k})
PY
  done
done
MT_LIB_VARIANT=libmtseg_hip_swz0.so python bench.py --workload infer --mirror 0 --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-also 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('infer swz0', d['value'], d['ms_per_step'])"
python bench.py --workload infer --mirror 0 --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-also 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('infer swz1', d['value'], d['ms_per_step'])"
