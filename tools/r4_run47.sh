# residual-add form of the fused norm-backward statistics (MT_FUSE_NORM_BWD=3) beside the weight-gradient stream: residual encoder, fp32 and mixed
run() { python bench.py "$@" --steps 10 --warmup 3 --no-also --no-cpu-baseline --no-traffic --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2 3; do for f in 0 3; do
echo -n "resenc mixed MT_FUSE_NORM_BWD=$f: "; MT_FUSE_NORM_BWD=$f run --workload resenc --precision bf16
echo -n "resenc fp32 MT_FUSE_NORM_BWD=$f: "; MT_FUSE_NORM_BWD=$f run --workload resenc
done; done
