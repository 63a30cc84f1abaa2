mkdir -p gpurun_out/r4
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r4/run18_tests.log 2>&1; tail -3 gpurun_out/r4/run18_tests.log
for w in task009 resenc task100; do
  python bench.py --workload $w --precision bf16 --steps 8 --warmup 2 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r4/run18_$w.json 2>/dev/null
  python - "$w" <<'PY'
import json,sys
w=sys.argv[1]
d=json.loads(open('gpurun_out/r4/run18_%s.json'%w).read().strip().splitlines()[-1])
bk=d['roofline']['all_conv_launches']['by_kernel_ms_per_step']
print(w,d['ms_per_step'], {k[:50]:x for k,x in bk.items()})
PY
done
bash tools/profile_r4.sh task009_bf16 > /dev/null 2>&1
grep -i "stem\|pack_weights" gpurun_out/prof_r4/task009_bf16_kernel_stats.csv | cut -c1-160
