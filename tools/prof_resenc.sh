cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/resencprof
mkdir -p $O
for p in fp32 bf16; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/$p -o r -- python $R/bench.py --workload resenc --precision $p --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/$p.json 2> $O/$p.err
echo "== $p"; head -14 $O/$p/r_kernel_stats.csv | cut -c1-120
done
