#!/bin/bash
# conv_x16_kernel / conv_bf16_kernel: where the cycles go.  Needs the variant libraries built by
#   for a in 7 12 4 3 35; do tools/build_x16_variant.sh abl$a -DX16_ABL=$a; done; tools/build_x16_variant.sh ts -DX16_TS=1
# Layers: 0 = 30->30 @ 2x48x192x192, 4 = 60->60 @ 2x48x96x96 (tools/bench_fwd16.py).  Output: stdout (copy to profiles/).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "# both kernels, every layer of tools/bench_fwd16.py (mode 1 = conv_x16_kernel wherever eligible, 0 = conv_bf16_kernel)"
python $R/tools/bench_fwd16.py 2>&1 | grep -v "amdgpu.ids"
for L in 0 4; do
  echo; echo "# layer $L: per-phase s_memtime totals of conv_x16_kernel (library built with -DX16_TS=1)"
  MT_LIB_VARIANT=libmtseg_hip_ts.so python $R/tools/bench_fwd16.py --modes 1 --only $L --ts 1 2>&1 | grep -v "amdgpu.ids\|^   max"
  echo "# layer $L: compile-time ablations (X16_ABL bits: 1 no patch / weight loads, 2 no conversion + LDS writes, 4 no epilogue, 8 no MFMAs, 32 no global stores)"
  for v in abl7:MFMA_phase_only abl12:loads+conversion_only abl4:no_epilogue abl3:MFMA+epilogue_only abl35:MFMA+epilogue_without_its_global_stores; do
    lib=${v%%:*}; what=${v##*:}
    echo "## $what (X16_ABL=${lib#abl})"
    MT_LIB_VARIANT=libmtseg_hip_$lib.so python $R/tools/bench_fwd16.py --modes 1 --only $L 2>&1 | grep conv_x16
  done
done
echo; echo "# SQ counters (rocprofv3 --pmc, two passes), summed over the launches of: bench_fwd16.py --only 4 --reps 2 (60->60 @ 2x48x96x96, fp16 + bf16, both kernels)"
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU"; do
  d=$(mktemp -d /tmp/x16pmc.XXXX)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o a -- python $R/tools/bench_fwd16.py --only 4 --reps 2 > /dev/null 2>&1
  python3 - "$d" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ', '').split('(')[0]
        if k.startswith('conv_x16') or k.startswith('conv_bf16'):
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r.get('Dispatch_Id', r.get('Correlation_Id', '')))
for k in sorted(agg):
    print('%s  (%d launches)' % (k, len(n[k])))
    for c, v in sorted(agg[k].items()):
        print('    %-28s %16.0f per launch' % (c, v / max(len(n[k]), 1)))
PY
  rm -rf $d
done
