cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="${PMC_ARGS:---mode fwd}"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc1 -o a -- python $R/tools/bench_conv.py $ARGS --reps 2 > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc2 -o a -- python $R/tools/bench_conv.py $ARGS --reps 2 > $R/gpurun_out/pmc2.log 2>&1
