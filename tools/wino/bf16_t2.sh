python bench.py --precision bf16 --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/bench_bf16.json
python bench.py --precision bf16 --no-cpu-baseline --no-roofline --workload resenc --steps 8 --warmup 2 2>&1 | tail -1 | cut -c1-200
python bench.py --precision bf16 --no-cpu-baseline --no-roofline --workload task100 --steps 8 --warmup 2 2>&1 | tail -1| cut -c1-200
python bench.py --workload infer --precision bf16 --mirror 0 --steps 2 --warmup 1 2>&1 | tail -1 > gpurun_out/infer_bf16_nomirror.json
python bench.py --workload infer --precision bf16 --mirror 1 --steps 1 --warmup 1 2>&1 | tail -1 > gpurun_out/infer_bf16_mirror.json
cat gpurun_out/bench_bf16.json | cut -c1-200; cat gpurun_out/infer_bf16_nomirror.json | cut -c1-160; cat gpurun_out/infer_bf16_mirror.json | cut -c1-160
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bf16c
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_profiled.json 2> $O/stats.err
