# workgroup counts of the backward-weight kernels rounded down to one round: parity + steps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_storage_bf16_gpu.py tests/test_golden_gpu.py -x -q -m gpu -k "bwd_weight or bwdw or golden or iteration" > gpurun_out/r5_nsg_tests.log 2>&1
tail -n 3 gpurun_out/r5_nsg_tests.log
run() { python tools/bench_conv.py "$@" --reps 20 2>&1 | grep -v amdgpu.ids | tail -n 1; }
run --mode bwdw --cin 240 --cout 240 --shape 6 24 24
run --mode bwdw --cin 480 --cout 240 --shape 6 24 24
run --mode bwdw --cin 240 --cout 120 --shape 12 48 48
for a in "" "--workload task100" "--workload resenc" "--workload resenc --precision bf16" "--precision bf16"; do
  timeout 600 python bench.py --no-also --no-roofline --no-cpu-baseline --steps 30 --warmup 5 $a > gpurun_out/r5_nsg.json 2> gpurun_out/r5_nsg.err
  python -c "import json;d=json.load(open('gpurun_out/r5_nsg.json'));print('$a', d['ms_per_step'])"
done
