# where are the idle gaps of a steady-state step? (top gaps with the kernels either side)
O=gpurun_out/s4; mkdir -p $O
export IDLE_TOP=30
( bash tools/gpu_idle.sh ) > $O/idle_fp32.txt 2>&1
( bash tools/gpu_idle.sh --precision bf16 ) > $O/idle_task009_mixed.txt 2>&1
( bash tools/gpu_idle.sh --workload resenc --precision bf16 ) > $O/idle_resenc_mixed.txt 2>&1
head -12 $O/idle_fp32.txt $O/idle_task009_mixed.txt $O/idle_resenc_mixed.txt
