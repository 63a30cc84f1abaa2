cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/inferprof
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o inf -- python $R/bench.py --workload infer --precision bf16 --mirror 0 --steps 1 --warmup 1 > $O/line.json 2> $O/err.txt
head -16 $O/stats/inf_kernel_stats.csv | cut -c1-125
