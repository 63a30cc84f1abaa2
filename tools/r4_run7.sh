#!/bin/bash
mkdir -p gpurun_out/r4
bash tools/profile_r4.sh resenc_bf16 > /dev/null 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_r4/resenc_bf16_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot/1e6)
for r in rows[:60]:
    print('%-100s %5s %9.3f ms %6.2f%%' % (r['Name'][:100], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['Percentage'])))
PY
for w in task009 task100; do timeout 300 python bench.py --workload $w --precision bf16 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w bf16', d['value'], d['ms_per_step'])"; done
