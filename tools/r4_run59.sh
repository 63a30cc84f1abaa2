# the tiles-per-workgroup condition of the cout-tile form: default (4) against "everywhere" (104) and "at most two" (2)
run() { python bench.py "$@" --steps 10 --warmup 3 --no-also --no-cpu-baseline --no-traffic --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2; do for f in 4 104 2; do
echo -n "task009 fp32 MT_BWDW_CW=$f: "; MT_BWDW_CW=$f run
echo -n "resenc fp32 MT_BWDW_CW=$f: "; MT_BWDW_CW=$f run --workload resenc
echo -n "task009 mixed MT_BWDW_CW=$f: "; MT_BWDW_CW=$f run --precision bf16
echo -n "resenc mixed MT_BWDW_CW=$f: "; MT_BWDW_CW=$f run --workload resenc --precision bf16
done; done
