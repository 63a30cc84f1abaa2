mkdir -p gpurun_out/r4
timeout 1500 python tools/bf16_accuracy.py 0,0,bf16 1,0,bf16 1,2048,bf16 1,0,fp16 > gpurun_out/r4/bf16_accuracy.txt 2> gpurun_out/r4/bf16_accuracy.err
cat gpurun_out/r4/bf16_accuracy.txt
( echo "# share of a steady-state training step without any kernel running (tools/gpu_idle.sh: rocprofv3 kernel trace, union of the kernel intervals of both streams)"
  for a in "" "--precision bf16" "--workload resenc --precision bf16"; do echo "bench.py $a"; bash tools/gpu_idle.sh $a; done ) > gpurun_out/r4/gpu_idle.txt 2>&1
cat gpurun_out/r4/gpu_idle.txt
