# SQ counters of one mixed-precision launch: bash tools/pmc_bwdw16.sh "<python script + args>"   (e.g. "tools/bench_bwdw16.py --modes 1 --only 0 --reps 2")
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD=${1:-"tools/bench_bwdw16.py --modes 1 --only 0 --reps 2"}
mkdir -p $R/gpurun_out
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmcA -o a -- python $R/$CMD > $R/gpurun_out/pmcA.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmcB -o a -- python $R/$CMD > $R/gpurun_out/pmcB.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ['GRAFT_REPO_ROOT']
for tag in ('pmcA', 'pmcB'):
    for f in glob.glob(R + '/gpurun_out/%s/**/*counter_collection.csv' % tag, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:40]
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        for k, d in agg.items():
            if 'tr16' in k or 'conv_bf16' in k:
                print(tag, k, {c: int(v) for c, v in d.items()})
PY
