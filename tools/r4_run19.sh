mkdir -p gpurun_out/r4
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r4/run19_tests.log 2>&1; tail -3 gpurun_out/r4/run19_tests.log
for w in task009 resenc; do
  python bench.py --workload $w --precision bf16 --steps 8 --warmup 2 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r4/run19_$w.json 2>/dev/null
  python - "$w" <<'PY'
import json,sys
w=sys.argv[1]
d=json.loads(open('gpurun_out/r4/run19_%s.json'%w).read().strip().splitlines()[-1])
bk=d['roofline']['all_conv_launches']['by_kernel_ms_per_step']
print(w,d['ms_per_step'], {k[:50]:x for k,x in list(bk.items())[:4]})
PY
done
