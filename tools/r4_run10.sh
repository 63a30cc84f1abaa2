#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_storage_bf16_gpu.py -q -m gpu -k "lowres_backward_weight" > gpurun_out/r4/t_storage.log 2>&1; echo "storage rc=$?"
grep -E "^FAILED|passed|failed|^E  " gpurun_out/r4/t_storage.log | tail -8
for g in 0 1; do MT_BWDW_GEMM=$g timeout 600 python bench.py --workload resenc --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-traffic --no-also 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('gemm=$g resenc bf16', d['value'], d['ms_per_step'], d['config']['final_loss'], d['roofline']['all_conv_launches']['by_kernel_ms_per_step'])"; done
for w in task009 task100; do timeout 300 python bench.py --workload $w --precision bf16 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w bf16', d['value'], d['ms_per_step'])"; done
