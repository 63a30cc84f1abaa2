# rocprofv3 kernel statistics of bench.py for every workload (fp32 + bf16): gpurun_out/prof_${ROUND:-r6}/<tag>_kernel_stats.csv + the bench lines
# usage: bash ROUND=r6 tools/profile_stats.sh [tags...]   (default: all)
# The statistics are taken with the weight-gradient stream serialised (MT_BWDW_STREAMS=0: every kernel runs alone, its duration is
# the kernel's — what bench.py's `roofline` pass measures); the tag *_overlap keeps the default overlap (durations then include
# the time two streams' kernels share the CUs).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_${ROUND:-r6}
mkdir -p $O
run() {  # tag, bench args...
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o b -- python $R/bench.py "$@" --no-cpu-baseline --no-traffic --no-also > $O/${tag}_profiled.json 2> $O/${tag}.err
  f=$(find $O/$tag -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $O/${tag}_kernel_stats.csv
  rm -rf $O/$tag
}
export MT_BWDW_STREAMS=0
TAGS=${@:-task009_fp32 task009_bf16 task100_fp32 task100_bf16 resenc_fp32 resenc_bf16 infer_nomirror_fp32}
for t in $TAGS; do
  case $t in
    task009_fp32) run $t --steps 5 --warmup 2 ;;
    task009_fp32_overlap) MT_BWDW_STREAMS=1 run $t --steps 5 --warmup 2 ;;
    task009_bf16) run $t --steps 5 --warmup 2 --precision bf16 ;;
    task100_fp32) run $t --steps 4 --warmup 2 --workload task100 ;;
    task100_bf16) run $t --steps 4 --warmup 2 --workload task100 --precision bf16 ;;
    resenc_fp32) run $t --steps 4 --warmup 2 --workload resenc ;;
    resenc_bf16) run $t --steps 4 --warmup 2 --workload resenc --precision bf16 ;;
    infer_nomirror_fp32) run $t --steps 1 --warmup 1 --workload infer --mirror 0 --volume 256 512 512 ;;
    infer_mirror_fp32) run $t --steps 1 --warmup 1 --workload infer --mirror 1 --volume 128 384 384 ;;
  esac
done
ls -la $O
