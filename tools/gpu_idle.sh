# how much of a training step is the GPU idle (no kernel of either stream running)?  rocprofv3 kernel trace of bench.py, union of the
# kernel intervals over the timed steps.  usage: bash tools/gpu_idle.sh [bench args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/idle
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -o b -- python $R/bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-also --no-traffic > $O/line.json 2> $O/err.log
python - "$O" <<'PY'
import csv, sys, os, json
O = sys.argv[1]
f = [os.path.join(dp, x) for dp, _, fs in os.walk(O + '/t') for x in fs if x.endswith('kernel_trace.csv')][0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
# steady state: the longest run of kernels without a gap above 2 ms (warm-up, allocation and teardown phases are separated by such gaps)
segs = []; cur = [rows[0]]; ce = rows[0][1]
for r in rows[1:]:
    if r[0] - ce > 2000000:
        segs.append(cur); cur = []
    cur.append(r); ce = max(ce, r[1])
segs.append(cur)
sel = max(segs, key=len)
busy = 0; cur_s, cur_e = sel[0][0], sel[0][1]
gaps = []; where = []; last = sel[0][2]
for s, e, name in sel[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e); where.append((s - cur_e, last, name)); cur_s, cur_e, last = s, e, name
    else:
        if e > cur_e:
            cur_e, last = e, name
busy += cur_e - cur_s
span = max(r[1] for r in sel) - sel[0][0]
gaps.sort()
d = json.loads(open(O + '/line.json').read().strip().splitlines()[-1])
print('ms_per_step %.3f | window %.2f ms, %d kernels, busy %.2f ms (%.1f %%), idle %.2f ms in %d gaps (median %.1f us, p90 %.1f us, max %.1f us)' % (
    d['ms_per_step'], span / 1e6, len(sel), busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(gaps),
    gaps[len(gaps) // 2] / 1e3 if gaps else 0, gaps[int(len(gaps) * 0.9)] / 1e3 if gaps else 0, gaps[-1] / 1e3 if gaps else 0))
# the longest gaps: which kernel ended last before the gap, which one started after it
where.sort(reverse=True)
for g, a, b in where[:int(os.environ.get('IDLE_TOP', '0'))]:
    print('  gap %7.1f us  after %-60s before %s' % (g / 1e3, a[:60], b[:60]))
PY
