# round-5 check on the GPU box: 16-bit storage parity, SQ counters of the tr16 backward-weight kernel, default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_storage_bf16_gpu.py tests/test_autocast_golden.py -x -q -m gpu > gpurun_out/r5_v_tests.log 2>&1
tail -n 3 gpurun_out/r5_v_tests.log
bash tools/pmc_bwdw16.sh "tools/bench_bwdw16.py --modes 1 --only 0 --reps 2" > gpurun_out/r5_pmc_tr16_30.txt 2>&1
bash tools/pmc_bwdw16.sh "tools/bench_bwdw16.py --modes 1 --only 3 --reps 2" > gpurun_out/r5_pmc_tr16_60.txt 2>&1
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/r5_bench2.json 2> gpurun_out/r5_bench2.err
tail -c 600 gpurun_out/r5_bench2.json
