#!/bin/bash
bash tools/profile_r4.sh resenc_bf16 > /dev/null 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_r4/resenc_bf16_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot/1e6)
for r in rows[:40]:
    print('%-100s %5s %9.3f ms %6.2f%% avg %8.1f us' % (r['Name'][:100], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['Percentage']), float(r['AverageNs'])/1e3))
PY
timeout 900 python -m pytest "tests/test_fullsize_oracle_gpu.py::test_resenc_fullsize_fp32_and_bf16_vs_oracle" -q -m gpu -s 2>&1 | grep -E "passed|failed|bf16 vs|worst" | tail -4
