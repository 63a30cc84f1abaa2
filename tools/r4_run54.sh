# plans' native patch 96x192x192 after the round's last kernels
mkdir -p gpurun_out/r4
for w in task100 resenc; do for p in fp32 bf16; do
python bench.py --workload $w --precision $p --patch 96 192 192 --steps 5 --warmup 2 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r4/bench_${w}_${p}_patch96.json 2>/dev/null
python -c "import json; d=json.loads(open('gpurun_out/r4/bench_${w}_${p}_patch96.json').read().strip().splitlines()[-1]); print('$w $p patch96', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"
done; done
