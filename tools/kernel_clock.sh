#!/bin/bash
# Effective shader clock per kernel of a command (MI355X_MICROARCH.md "DVFS give-back": GRBM_GUI_ACTIVE / kernel wall time), run on the GPU box:
#   tools/kernel_clock.sh -- python bench.py --steps 3 --warmup 2 --no-also --no-roofline --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
shift
d=$(mktemp -d /tmp/clk.XXXX)
(cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $d -o a -- "$@" > /dev/null 2>&1)
python3 - "$d" <<'PY'
import csv, glob, sys, collections
dur = {}
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r.get('Dispatch_Id') or r.get('Correlation_Id')] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'])
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != 'GRBM_GUI_ACTIVE':
            continue
        k = r['Kernel_Name'].replace('void ', '').split('(')[0]
        did = r.get('Dispatch_Id') or r.get('Correlation_Id')
        ns = dur.get(did, (None,))[0]
        if ns is None:
            ns = int(r['End_Timestamp']) - int(r['Start_Timestamp']) if 'End_Timestamp' in r else None
        if ns:
            a = agg[k]; a[0] += float(r['Counter_Value']); a[1] += ns; a[2] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot_c = sum(v[0] for _, v in rows); tot_t = sum(v[1] for _, v in rows)
print('%-72s %7s %10s %9s' % ('kernel', 'calls', 'total us', 'GHz'))
for k, (cyc, ns, n) in rows[:28]:
    print('%-72s %7d %10.1f %9.3f' % (k[:72], n, ns / 1e3, cyc / ns))
print('%-72s %7s %10.1f %9.3f' % ('all kernels (time-weighted)', '', tot_t / 1e3, tot_c / tot_t))
PY
rm -rf $d
