#!/bin/bash
# SQ counters of the kernels whose name starts with PREFIX, per launch (rocprofv3 --pmc, two passes; run on the GPU box):
#   tools/pmc_kernels.sh PREFIX -- python tools/bench_conv.py ...
cd /tmp && export TMPDIR=/tmp
prefix=$1; shift; shift
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU"; do
  d=$(mktemp -d /tmp/pmc.XXXX)
  (cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o a -- "$@" > /dev/null 2>&1)
  python3 - "$d" "$prefix" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ', '').split('(')[0]
        if k.startswith(sys.argv[2]):
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r.get('Dispatch_Id', r.get('Correlation_Id', '')))
for k in sorted(agg):
    print('%s  (%d launches)' % (k, len(n[k])))
    for c, v in sorted(agg[k].items()):
        print('    %-28s %16.0f per launch' % (c, v / max(len(n[k]), 1)))
PY
  rm -rf $d
done
