# Round profile: bench line, rocprofv3 kernel stats of the same command, PMC traffic counters of the dominant kernel.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round
mkdir -p $O
python $R/bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_profiled.json 2> $O/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o conv -- python $R/tools/bench_conv.py --mode fwd --reps 2 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o conv -- python $R/tools/bench_conv.py --mode fwd --reps 2 > $O/pmc_write.log 2>&1
lscpu | head -20 > $O/lscpu.txt
ls -R $O | head -40
