# how long does the end of a training step run on the weight-gradient stream alone?  rocprofv3 kernel trace of bench.py (default
# streams); for every sgd_nesterov_kernel: the kernels that ended in the last 3 ms before it, per queue.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tail
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -o b -- python $R/bench.py "$@" --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-also --no-traffic > $O/line.json 2> $O/err.log
python - "$O" <<'PY'
import csv, sys, os
O = sys.argv[1]
f = [os.path.join(dp, x) for dp, _, fs in os.walk(O + '/t') for x in fs if x.endswith('kernel_trace.csv')][0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in csv.DictReader(open(f))]
rows.sort()
sgd = [i for i, r in enumerate(rows) if r[2].startswith('sgd_nesterov')]
i = sgd[-2]
t0 = rows[i][0]
print('kernels that ended within 2.5 ms before the optimizer step (us before its start; queue; duration us; name):')
for s, e, name, q in rows[max(0, i - 120):i + 1]:
    if t0 - e < 2500000:
        print('  start -%7.1f end -%7.1f  q%s  %7.1f us  %s' % ((t0 - s) / 1e3, (t0 - e) / 1e3, q, (e - s) / 1e3, name[:70]))
PY
