# every bench line quoted in DESIGN.md / README.md for round 2, one MI355X: gpurun_out/meas_r2/*.json (copied to profiles/r02_*.json)
O=gpurun_out/meas_r2; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_task009_fp32.json 2> $O/task009.err
python bench.py --workload task100 --steps 8 --warmup 2 > $O/bench_task100_fp32.json 2> $O/task100.err
python bench.py --workload resenc --steps 8 --warmup 2 > $O/bench_resenc_fp32.json 2> $O/resenc.err
python bench.py --workload resenc --precision bf16 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_resenc_bf16.json 2>> $O/resenc.err
python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_task009_bf16.json 2>> $O/task009.err
python bench.py --workload task100 --precision bf16 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_task100_bf16.json 2>> $O/task100.err
python bench.py --workload task100 --patch 96 192 192 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/bench_task100_fp32_patch96.json 2>> $O/task100.err
python bench.py --workload resenc --patch 96 192 192 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/bench_resenc_fp32_patch96.json 2>> $O/resenc.err
python bench.py --workload infer --mirror 0 --steps 2 --warmup 1 > $O/bench_infer_nomirror_fp32.json 2> $O/infer.err
python bench.py --workload infer --mirror 1 --steps 1 --warmup 1 --no-traffic > $O/bench_infer_mirror_fp32.json 2>> $O/infer.err
python bench.py --workload infer --mirror 0 --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > $O/bench_infer_nomirror_bf16.json 2>> $O/infer.err
python bench.py --workload infer --mirror 1 --precision bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-traffic > $O/bench_infer_mirror_bf16.json 2>> $O/infer.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/meas_r2/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get('roofline', {}); c = d.get('cpu_baseline', {})
        print('%-38s %8.3f %-12s %9.2f ms | %s frac %s traffic %s | cpu %s' % (f.split('/')[-1], d['value'], d['unit'], d['ms_per_step'], r.get('kernel'), r.get('frac'), r.get('traffic'), c.get('value')))
    except Exception as e:
        print(f, 'FAILED', e)
PY
