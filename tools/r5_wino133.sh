# KD = 1 instance of the Winograd backward-weight kernel: parity, then the residual encoder's fp32 step with and without it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "bwd_weight" > gpurun_out/r5_w133_tests.log 2>&1
tail -n 5 gpurun_out/r5_w133_tests.log
for m in 1 0; do
  MT_BWDW_WINO=$m timeout 600 python bench.py --no-also --steps 20 --warmup 3 --workload resenc > gpurun_out/r5_w133_resenc_$m.json 2> gpurun_out/r5_w133_$m.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r5_w133_resenc_$m.json'))
print('resenc fp32 MT_BWDW_WINO=$m', d['ms_per_step'], d.get('step_frac_of_fp32_mfma_roofline'))
print({k:v for k,v in d['roofline']['all_conv_launches']['by_kernel_ms_per_step'].items()})
PY
done
