timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/packchk
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_profiled.json 2> $O/stats.err
grep "pack_weights\|conv_wino8" $O/stats/bench_kernel_stats.csv | cut -c1-120
cat $O/bench_profiled.json | cut -c1-200
