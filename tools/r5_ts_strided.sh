# strided tap-split: parity, then the default step (fp32 Task009, mixed resenc) with and without it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_storage_bf16_gpu.py tests/test_fullsize_properties_gpu.py -x -q -m gpu -k "tapsplit or strided or fp16_storage or adjoint or conv_fwd" > gpurun_out/r5_ts_tests.log 2>&1
tail -n 5 gpurun_out/r5_ts_tests.log
for m in 1 0; do
  MT_CONV_TAPSPLIT=$m timeout 600 python bench.py --no-also --steps 30 --warmup 5 > gpurun_out/r5_ts_fp32_$m.json 2> gpurun_out/r5_ts_$m.err
  python -c "import json;d=json.load(open('gpurun_out/r5_ts_fp32_$m.json'));print('fp32 task009 tapsplit=$m', d['ms_per_step'], d.get('step_frac_of_fp32_mfma_roofline'))"
done
for m in 1 0; do
  [ $m = 0 ] && export MT_TS_STRIDED_OFF=1
  MT_CONV_TAPSPLIT=$m timeout 600 python bench.py --no-also --steps 30 --warmup 5 --workload resenc --precision bf16 > gpurun_out/r5_ts_resenc16_$m.json 2>> gpurun_out/r5_ts_$m.err
  python -c "import json;d=json.load(open('gpurun_out/r5_ts_resenc16_$m.json'));print('mixed resenc tapsplit=$m', d['ms_per_step'])"
done
