#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_storage_bf16_gpu.py -q -m gpu > gpurun_out/r4/t_storage.log 2>&1; echo "storage rc=$?"
timeout 900 python -m pytest tests/test_mixed_precision_gpu.py tests/test_network_gpu.py -q -m gpu -s > gpurun_out/r4/t_net.log 2>&1; echo "net rc=$?"
bash tools/profile_r4.sh resenc_bf16 > /dev/null 2>&1
grep -E "passed|failed" gpurun_out/r4/t_storage.log | tail -3; grep -E "passed|failed|cosine" gpurun_out/r4/t_net.log | tail -5
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_r4/resenc_bf16_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot/1e6)
for r in rows[:45]:
    print('%-95s %5s %9.3f ms %6.2f%%' % (r['Name'][:95], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['Percentage'])))
PY
