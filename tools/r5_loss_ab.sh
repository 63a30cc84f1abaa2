# fused loss combination: parity tests, then the default training step with and without it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_loss_combine_gpu.py tests/test_golden_gpu.py tests/test_ddp_world2_gpu.py tests/test_trainer_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "loss or golden or world2 or trainer or iteration or fused" > gpurun_out/r5_loss_tests.log 2>&1
tail -n 15 gpurun_out/r5_loss_tests.log
for m in 1 0; do
  MT_FUSED_LOSS=$m timeout 600 python bench.py --no-also --steps 30 --warmup 5 > gpurun_out/r5_loss_ab_fp32_$m.json 2> gpurun_out/r5_loss_ab_$m.err
  python -c "import json;d=json.load(open('gpurun_out/r5_loss_ab_fp32_$m.json'));print('fp32 task009 fused=$m', d['ms_per_step'])"
  MT_FUSED_LOSS=$m timeout 600 python bench.py --no-also --steps 30 --warmup 5 --workload resenc --precision bf16 > gpurun_out/r5_loss_ab_resenc16_$m.json 2>> gpurun_out/r5_loss_ab_$m.err
  python -c "import json;d=json.load(open('gpurun_out/r5_loss_ab_resenc16_$m.json'));print('mixed resenc fused=$m', d['ms_per_step'])"
done
