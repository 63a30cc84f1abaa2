#!/bin/bash
mkdir -p gpurun_out/r4
./tools/ubench/mfma_bf16_mix > gpurun_out/r4/ubench_mfma_bf16_mix.txt 2>&1; cat gpurun_out/r4/ubench_mfma_bf16_mix.txt
timeout 1500 python -m pytest tests/test_fullsize_infer_gpu.py "tests/test_fullsize_oracle_gpu.py::test_task100_native_patch_96x192x192_vs_oracle" tests/test_ddp_world2_gpu.py -q -m gpu -s -x > gpurun_out/r4/t_new.log 2>&1; echo "new tests rc=$?"
grep -E "^FAILED|passed|failed|vs oracle|max \|d\||gradient vs fp64" gpurun_out/r4/t_new.log | tail -12
for w in task100 resenc; do for pr in fp32 bf16; do
  timeout 600 python bench.py --workload $w --precision $pr --patch 96 192 192 --steps 6 --warmup 2 --no-cpu-baseline --no-traffic --no-also 2>gpurun_out/r4/b_${w}_${pr}_patch96.err | tail -1 > gpurun_out/r4/b_${w}_${pr}_patch96.json
  python -c "import json; d=json.load(open('gpurun_out/r4/b_${w}_${pr}_patch96.json')); print('$w $pr patch96', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"
done; done
