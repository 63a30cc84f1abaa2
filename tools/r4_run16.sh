mkdir -p gpurun_out/r4
bash tools/profile_r4.sh resenc_bf16 task009_bf16 > gpurun_out/r4/run16_profile.log 2>&1
python - <<'PY'
import csv
for t in ('resenc_bf16','task009_bf16'):
    rows=list(csv.DictReader(open('gpurun_out/prof_r4/%s_kernel_stats.csv'%t)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    print(t,'total ms',tot/1e6)
    for r in rows[:32]:
        print('  %-84s n %5s  tot %8.2f ms  avg %8.1f us  %5.1f%%'%(r['Name'][:84],r['Calls'],float(r['TotalDurationNs'])/1e6,float(r['AverageNs'])/1e3,100*float(r['TotalDurationNs'])/tot))
PY
