# every number quoted in README.md / DESIGN.md §5 in one run (one MI355X): gpurun_out/all/*.json
O=gpurun_out/all; mkdir -p $O
python bench.py --steps 10 --warmup 3 > $O/task009_fp32.json 2>/dev/null
python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/task009_bf16.json 2>/dev/null
for w in task100 resenc; do for p in fp32 bf16; do
python bench.py --workload $w --precision $p --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > $O/${w}_$p.json 2>/dev/null
done; done
for p in fp32 bf16; do
python bench.py --workload infer --precision $p --mirror 0 --steps 2 --warmup 1 > $O/infer_nomirror_$p.json 2>/dev/null
python bench.py --workload infer --precision $p --mirror 1 --steps 1 --warmup 1 > $O/infer_mirror_$p.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/all/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print('%-28s %8.3f %-12s %9.2f ms' % (f.split('/')[-1], d['value'], d['unit'], d['ms_per_step']))
    except Exception as e:
        print(f, 'FAILED', e)
PY
