# K-split strided backward-data on under-filled grids: parity, then the fp32 steps with and without (MT_CONV_TAPSPLIT also switches the
# forward tap-split kernels, so the A/B is the per-kernel table of the roofline pass)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_properties_gpu.py tests/test_golden_gpu.py -x -q -m gpu -k "bwd_data_strided or adjoint or golden or iteration" > gpurun_out/r5_ks_tests.log 2>&1
tail -n 5 gpurun_out/r5_ks_tests.log
for w in task009 resenc; do
  timeout 600 python bench.py --no-also --steps 30 --warmup 5 --workload $w > gpurun_out/r5_ks_$w.json 2> gpurun_out/r5_ks.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r5_ks_$w.json'))
print('$w fp32', d['ms_per_step'], d.get('step_frac_of_fp32_mfma_roofline'))
print({k:v for k,v in d['roofline']['all_conv_launches']['by_kernel_ms_per_step'].items() if 'strided' in k or 'tapsplit' in k})
PY
done
