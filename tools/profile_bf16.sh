# bf16-mode profile: bench line, rocprofv3 kernel stats, PMC traffic counters of conv_bf16_kernel (32->32 @ 48x192x192, B=2)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bf16
mkdir -p $O
python $R/bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_profiled.json 2> $O/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o conv -- python $R/tools/bench_conv.py --mode fwd --cin 32 --cout 32 --mma 1 --reps 2 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o conv -- python $R/tools/bench_conv.py --mode fwd --cin 32 --cout 32 --mma 1 --reps 2 > $O/pmc_write.log 2>&1
ls -R $O | head -40
