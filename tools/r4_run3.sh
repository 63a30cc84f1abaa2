#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_storage_bf16_gpu.py -q -m gpu > gpurun_out/r4/t_storage.log 2>&1; echo "storage rc=$?"
MT_IO_DEBUG=1 timeout 300 python tools/debug_bf16_ops.py plain bf16 > gpurun_out/r4/dbg_plain_bf16.log 2>&1; echo "plain bf16 rc=$?"
MT_IO_DEBUG=1 timeout 300 python tools/debug_bf16_ops.py resenc bf16 > gpurun_out/r4/dbg_resenc_bf16.log 2>&1; echo "resenc bf16 rc=$?"
timeout 900 python -m pytest tests/test_mixed_precision_gpu.py tests/test_network_gpu.py -x -q -m gpu > gpurun_out/r4/t_net.log 2>&1; echo "net rc=$?"
MT_IO_DEBUG=1 timeout 600 python bench.py --workload resenc --precision bf16 --steps 8 --warmup 3 --no-roofline > gpurun_out/r4/b_resenc_bf16.json 2> gpurun_out/r4/b_resenc_bf16.err; echo "bench rc=$?"
grep -E "AssertionError|passed|failed" gpurun_out/r4/t_storage.log | tail -8; tail -3 gpurun_out/r4/dbg_plain_bf16.log; tail -3 gpurun_out/r4/dbg_resenc_bf16.log; tail -3 gpurun_out/r4/t_net.log; tail -c 400 gpurun_out/r4/b_resenc_bf16.json
