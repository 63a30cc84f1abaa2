#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_storage_bf16_gpu.py -q -m gpu -k "bwdw_wino" > gpurun_out/r4/t_storage.log 2>&1; echo "storage rc=$?"
grep -E "^FAILED|passed|failed" gpurun_out/r4/t_storage.log | tail -8
for st in 0 1; do MT_BWDW_STAGED=$st timeout 600 python bench.py --workload resenc --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-traffic --no-also 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('staged=$st resenc bf16', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['all_conv_launches']['by_kernel_ms_per_step'])"; done
timeout 600 python -m pytest tests/test_ddp_world2_gpu.py -q -m gpu -k "self_validation" > gpurun_out/r4/t_bench2.log 2>&1; echo "bench2 rc=$?"; grep -E "^FAILED|passed|failed|^E  " gpurun_out/r4/t_bench2.log | head -5
