mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/idle_inf
rm -rf $O; mkdir -p $O
for prec in bf16 fp32; do
rocprofv3 --kernel-trace --output-format csv -d $O/$prec -o b -- python $R/bench.py --workload infer --mirror 0 --precision $prec --volume 256 512 512 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-also --no-traffic > $O/line_$prec.json 2> $O/err_$prec.log
python - "$O" "$prec" <<'PY'
import csv, sys, os, json
O, prec = sys.argv[1:3]
f = [os.path.join(dp, x) for dp, _, fs in os.walk(O + '/' + prec) for x in fs if x.endswith('kernel_trace.csv')][0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(f)))
segs = []; cur = [rows[0]]; ce = rows[0][1]
for r in rows[1:]:
    if r[0] - ce > 2000000: segs.append(cur); cur = []
    cur.append(r); ce = max(ce, r[1])
segs.append(cur)
sel = max(segs, key=len)
busy = 0; cs, ce = sel[0]; gaps = []
for s, e in sel[1:]:
    if s > ce: busy += ce - cs; gaps.append(s - ce); cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
span = max(r[1] for r in sel) - sel[0][0]
gaps.sort()
d = json.loads(open(O + '/line_%s.json' % prec).read().strip().splitlines()[-1])
print('infer %s: ms %.1f | window %.1f ms, %d kernels, busy %.1f%%, idle %.2f ms in %d gaps (median %.1f us, p90 %.1f, max %.1f)' % (prec, d['ms_per_step'], span/1e6, len(sel), 100.0*busy/span, (span-busy)/1e6, len(gaps), gaps[len(gaps)//2]/1e3, gaps[int(len(gaps)*.9)]/1e3, gaps[-1]/1e3))
PY
done
