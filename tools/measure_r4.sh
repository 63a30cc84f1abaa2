# every bench line quoted in DESIGN.md / README.md for round 4, one MI355X: gpurun_out/meas_r4/*.json (copied to profiles/r04_*.json),
# rocprofv3 kernel statistics of the same commands (tools/profile_r4.sh) and the per-kernel HBM counters (tools/pmc_per_kernel_r4.sh)
O=gpurun_out/meas_r4; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/default.err
python bench.py --workload task100 --steps 8 --warmup 2 --no-also > $O/bench_task100_fp32.json 2> $O/task100.err
python bench.py --workload resenc --steps 8 --warmup 2 --no-also > $O/bench_resenc_fp32.json 2> $O/resenc.err
python bench.py --workload resenc --precision bf16 --steps 8 --warmup 2 --no-cpu-baseline --no-also > $O/bench_resenc_bf16.json 2>> $O/resenc.err
python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-also > $O/bench_task009_bf16.json 2>> $O/default.err
python bench.py --workload task100 --precision bf16 --steps 8 --warmup 2 --no-cpu-baseline --no-also > $O/bench_task100_bf16.json 2>> $O/task100.err
python bench.py --workload infer --mirror 0 --steps 2 --warmup 1 --no-also > $O/bench_infer_nomirror_fp32.json 2> $O/infer.err
python bench.py --workload infer --mirror 1 --steps 1 --warmup 1 --no-traffic --no-cpu-baseline --no-also > $O/bench_infer_mirror_fp32.json 2>> $O/infer.err
python bench.py --workload infer --mirror 0 --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-also > $O/bench_infer_nomirror_bf16.json 2>> $O/infer.err
python bench.py --workload infer --mirror 1 --precision bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-traffic --no-also > $O/bench_infer_mirror_bf16.json 2>> $O/infer.err
bash tools/profile_r4.sh task009_fp32 task009_fp32_overlap task009_bf16 task100_fp32 resenc_fp32 resenc_bf16 infer_nomirror_fp32 > $O/profile.log 2>&1
cp gpurun_out/prof_r4/*_kernel_stats.csv $O/ 2>/dev/null
bash tools/pmc_per_kernel_r4.sh task009_fp32 task009_bf16 resenc_bf16 > $O/pmc_per_kernel.log 2>&1
cp gpurun_out/pmc_r4/pmc_per_kernel.json $O/ 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/meas_r4/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get('roofline', {}); c = d.get('cpu_baseline', {})
        print('%-34s %8.3f %-12s %9.2f ms | %s frac %s traffic %s | cpu %s' % (f.split('/')[-1], d['value'], d['unit'], d['ms_per_step'], r.get('kernel'), r.get('frac'), r.get('traffic'), c.get('value')))
        for k, v in d.get('also', {}).items():
            print('     also %-22s %8.3f %-12s %9.2f ms | %s frac %s' % (k, v['value'], v['unit'], v['ms_per_step'], v.get('roofline', {}).get('kernel'), v.get('roofline', {}).get('frac')))
    except Exception as e:
        print(f, 'FAILED', e)
PY
tail -80 $O/pmc_per_kernel.log
