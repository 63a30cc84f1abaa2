#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_storage_bf16_gpu.py -q -m gpu > gpurun_out/r4/t_storage.log 2>&1; echo "storage rc=$?"
grep -E "^FAILED|passed|failed" gpurun_out/r4/t_storage.log | tail -12
MT_IO_DEBUG=1 timeout 300 python tools/debug_bf16_ops.py plain bf16 > gpurun_out/r4/dbg_plain_bf16.log 2>&1; echo "plain rc=$?"; tail -2 gpurun_out/r4/dbg_plain_bf16.log
timeout 900 python -m pytest tests/test_mixed_precision_gpu.py tests/test_network_gpu.py -q -m gpu -s > gpurun_out/r4/t_net.log 2>&1; echo "net rc=$?"
grep -E "^FAILED|passed|failed|cosine" gpurun_out/r4/t_net.log | tail -5
timeout 800 python tools/bf16_accuracy.py 1,0,fp16 1,0,bf16 2>&1 | grep -v Warn | tail -4
MT_IO_DEBUG=1 timeout 600 python bench.py --workload resenc --precision bf16 --steps 8 --warmup 3 --no-roofline --no-cpu-baseline > gpurun_out/r4/b_resenc_bf16.json 2> gpurun_out/r4/b_resenc_bf16.err; echo "bench rc=$?"
tail -c 300 gpurun_out/r4/b_resenc_bf16.json
