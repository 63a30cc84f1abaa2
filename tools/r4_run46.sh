# A/B of scheduling switches in the mixed mode after the round's last kernels (ABI v3)
run() { python bench.py "$@" --steps 10 --warmup 3 --no-also --no-cpu-baseline --no-traffic --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
T9="--precision bf16"; RE="--workload resenc --precision bf16"
echo -n "task009 mixed default: "; run $T9
echo -n "resenc mixed default: "; run $RE
echo -n "task009 mixed MT_BWDW_STREAMS=0: "; MT_BWDW_STREAMS=0 run $T9
echo -n "resenc mixed MT_BWDW_STREAMS=0: "; MT_BWDW_STREAMS=0 run $RE
echo -n "task009 mixed MT_FUSE_NORM_BWD=1: "; MT_FUSE_NORM_BWD=1 run $T9
echo -n "resenc mixed MT_FUSE_NORM_BWD=1: "; MT_FUSE_NORM_BWD=1 run $RE
echo -n "resenc mixed MT_FUSE_NORM_BWD=3: "; MT_FUSE_NORM_BWD=3 run $RE
echo -n "task009 mixed MT_BWDW_MARCH=0: "; MT_BWDW_MARCH=0 run $T9
echo -n "task009 mixed default: "; run $T9
echo -n "resenc mixed default: "; run $RE
