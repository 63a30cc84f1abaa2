# finer voxel blocks for small tensors + 1024-thread inorm_bwd_small_kernel: norm / engine parity, then the steps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_golden_gpu.py tests/test_storage_bf16_gpu.py tests/test_mixed_precision_gpu.py tests/test_autocast_golden.py -x -q -m gpu -k "norm or lrelu or golden or iteration or resenc or mixed or channel_sum or autocast" > gpurun_out/r5_sn_tests.log 2>&1
tail -n 4 gpurun_out/r5_sn_tests.log
for a in "--workload resenc --precision bf16" "--workload resenc" "" "--precision bf16"; do
  timeout 600 python bench.py --no-also --steps 30 --warmup 5 $a > gpurun_out/r5_sn.json 2> gpurun_out/r5_sn.err
  python -c "import json;d=json.load(open('gpurun_out/r5_sn.json'));print('$a', d['ms_per_step'], d.get('step_frac_of_fp32_mfma_roofline'))"
done
