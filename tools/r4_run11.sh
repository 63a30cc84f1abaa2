#!/bin/bash
mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests/ -q -m gpu -x > gpurun_out/r4/t_all.log 2>&1; echo "all rc=$?"
grep -E "^FAILED|passed|failed|^E  " gpurun_out/r4/t_all.log | tail -8
timeout 600 python bench.py --workload resenc --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-traffic --no-also 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('resenc bf16', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['all_conv_launches']['by_kernel_ms_per_step'])"
