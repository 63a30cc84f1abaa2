# HBM traffic of every kernel of the bench step, per kernel name : separate FETCH_SIZE / WRITE_SIZE passes over
# bench.py (no other trace domain beside --kernel-trace), joined with the serialised kernel durations of tools/profile_r5.sh
# -> gpurun_out/pmc_r5/pmc_per_kernel.json : {workload: {kernel: {launches, fetch_kb, write_kb, avg_us, hbm_gbs, frac_of_8TBs}}}
# usage: bash tools/pmc_per_kernel.sh [tags...]      tags as in tools/profile_r5.sh (their *_kernel_stats.csv must exist)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_r5
mkdir -p $O
export MT_BWDW_STREAMS=0
TAGS=${@:-task009_fp32 task009_bf16 resenc_bf16}
for t in $TAGS; do
  case $t in
    task009_fp32) A="" ;;
    task009_bf16) A="--precision bf16" ;;
    task100_fp32) A="--workload task100" ;;
    task100_bf16) A="--workload task100 --precision bf16" ;;
    resenc_fp32) A="--workload resenc" ;;
    resenc_bf16) A="--workload resenc --precision bf16" ;;
  esac
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${t}_$c -o b -- python $R/bench.py $A --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-also --no-traffic > $O/${t}_$c.log 2>&1
  done
done
python - $TAGS <<'PY'
import csv, json, os, sys, collections
R = os.environ['GRAFT_REPO_ROOT']; O = R + '/gpurun_out/pmc_r5'
out = {}
for t in sys.argv[1:]:
    agg = collections.defaultdict(lambda: {'n': 0, 'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0})
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        f = O + '/%s_%s/b_counter_collection.csv' % (t, c)
        if not os.path.exists(f):
            cand = [os.path.join(dp, x) for dp, _, fs in os.walk(O + '/%s_%s' % (t, c)) for x in fs if x.endswith('counter_collection.csv')]
            f = cand[0]
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            agg[k][c] += float(r['Counter_Value'])
            if c == 'FETCH_SIZE': agg[k]['n'] += 1
    dur = {}
    sf = R + '/gpurun_out/prof_r5/%s_kernel_stats.csv' % t
    if os.path.exists(sf):
        for r in csv.DictReader(open(sf)):
            dur[r['Name']] = float(r['AverageNs']) / 1e3
    res = {}
    for k, v in agg.items():
        if not v['n']: continue
        # both counters are in KB; gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE tallies 64 B per 128-byte
        # request, so the fetched bytes are 2 x FETCH_SIZE (what bench.py's measure_traffic applies); WRITE_SIZE as counted
        fk = 2.0 * v['FETCH_SIZE'] / v['n']; wk = v['WRITE_SIZE'] / v['n']
        e = {'launches': v['n'], 'fetch_kb_per_launch': round(fk, 1), 'write_kb_per_launch': round(wk, 1), 'fetch_correction': 'x2 (gfx950)'}
        if k in dur:
            e['avg_us'] = round(dur[k], 2)
            e['hbm_gbs'] = round((fk + wk) * 1024 / (dur[k] * 1e-6) / 1e9, 1)
            e['frac_of_8TBs'] = round(e['hbm_gbs'] / 8000.0, 3)
        res[k] = e
    out[t] = dict(sorted(res.items(), key=lambda kv: -kv[1]['launches'] * kv[1].get('avg_us', 0))[:24])
json.dump(out, open(O + '/pmc_per_kernel.json', 'w'), indent=1)
for t, v in out.items():
    print(t)
    for k, e in list(v.items())[:24]:
        print('  %-78s n %4d  %9.0f + %9.0f KB  %8s us  %7s GB/s  %s' % (k[:78], e['launches'], e['fetch_kb_per_launch'], e['write_kb_per_launch'], e.get('avg_us'), e.get('hbm_gbs'), e.get('frac_of_8TBs')))
PY
