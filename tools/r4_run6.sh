#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_storage_bf16_gpu.py -q -m gpu > gpurun_out/r4/t_storage.log 2>&1; echo "storage rc=$?"
grep -E "^FAILED|passed|failed" gpurun_out/r4/t_storage.log | tail -30
MT_IO_DEBUG=1 timeout 300 python tools/debug_bf16_ops.py resenc bf16 > gpurun_out/r4/dbg_resenc_bf16.log 2>&1; echo "resenc dbg rc=$?"; tail -2 gpurun_out/r4/dbg_resenc_bf16.log; grep "mt io" gpurun_out/r4/dbg_resenc_bf16.log | sort | uniq | head -40
timeout 1500 python -m pytest tests/ -q -m gpu -x --deselect tests/test_storage_bf16_gpu.py > gpurun_out/r4/t_all.log 2>&1; echo "all rc=$?"
grep -E "^FAILED|passed|failed" gpurun_out/r4/t_all.log | tail -8
MT_IO_DEBUG=1 timeout 600 python bench.py --workload resenc --precision bf16 --steps 8 --warmup 3 --no-roofline --no-cpu-baseline > gpurun_out/r4/b_resenc_bf16.json 2> gpurun_out/r4/b_resenc_bf16.err; echo "bench rc=$?"
grep "mt io" gpurun_out/r4/b_resenc_bf16.json | sort | uniq | head -40; tail -c 300 gpurun_out/r4/b_resenc_bf16.json
