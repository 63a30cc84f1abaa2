# PMC traffic of the WHOLE bench step, per kernel name: separate FETCH_SIZE / WRITE_SIZE passes over bench.py (no other tracing)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcbench
mkdir -p $O
for prec in fp32 bf16; do
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${prec}_fetch -o b -- python $R/bench.py --precision $prec --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-also --no-traffic > $O/${prec}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${prec}_write -o b -- python $R/bench.py --precision $prec --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-also --no-traffic > $O/${prec}_write.log 2>&1
done
python - <<'PY'
import csv, json, os, collections
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmcbench'
out={}
for prec in ('fp32','bf16'):
    agg=collections.defaultdict(lambda: {'n':0,'FETCH_SIZE':0.0,'WRITE_SIZE':0.0})
    for nm,ctr in (('fetch','FETCH_SIZE'),('write','WRITE_SIZE')):
        f=O+'/%s_%s/b_counter_collection.csv'%(prec,nm)
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name']
            agg[k][ctr]+=float(r['Counter_Value'])
            if nm=='fetch': agg[k]['n']+=1
    res={k:{'launches':v['n'],'fetch_kb_per_launch':v['FETCH_SIZE']/max(v['n'],1),'write_kb_per_launch':v['WRITE_SIZE']/max(v['n'],1)} for k,v in agg.items() if v['n']}
    out[prec]=dict(sorted(res.items(), key=lambda kv:-kv[1]['launches']*(kv[1]['fetch_kb_per_launch']+kv[1]['write_kb_per_launch']))[:14])
json.dump(out, open(O+'/pmc_per_kernel.json','w'), indent=1)
print(json.dumps({p:list(v.items())[:3] for p,v in out.items()}, indent=1)[:1500])
PY
