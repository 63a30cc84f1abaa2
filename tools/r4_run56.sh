# two weight-gradient streams instead of one
run() { python bench.py "$@" --steps 10 --warmup 3 --no-also --no-cpu-baseline --no-traffic --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2; do for f in 1 2; do
echo -n "task009 fp32 MT_BWDW_STREAMS=$f: "; MT_BWDW_STREAMS=$f run
echo -n "task009 mixed MT_BWDW_STREAMS=$f: "; MT_BWDW_STREAMS=$f run --precision bf16
echo -n "resenc mixed MT_BWDW_STREAMS=$f: "; MT_BWDW_STREAMS=$f run --workload resenc --precision bf16
echo -n "task100 fp32 MT_BWDW_STREAMS=$f: "; MT_BWDW_STREAMS=$f run --workload task100
done; done
