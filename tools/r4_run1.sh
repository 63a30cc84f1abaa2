#!/bin/bash
# round 4, first GPU call: bf16 storage kernels + the mixed-precision network tests + a residual-encoder bench line
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_storage_bf16_gpu.py -x -q -m gpu > gpurun_out/r4/t_storage.log 2>&1; echo "storage rc=$?" 
timeout 900 python -m pytest tests/test_mixed_precision_gpu.py tests/test_network_gpu.py -x -q -m gpu > gpurun_out/r4/t_net.log 2>&1; echo "net rc=$?"
MT_IO_DEBUG=1 timeout 600 python bench.py --workload resenc --precision bf16 --steps 8 --warmup 3 --no-roofline > gpurun_out/r4/b_resenc_bf16.json 2> gpurun_out/r4/b_resenc_bf16.err; echo "bench rc=$?"
tail -3 gpurun_out/r4/t_storage.log; tail -3 gpurun_out/r4/t_net.log; tail -c 600 gpurun_out/r4/b_resenc_bf16.json
