#!/bin/bash
# the DDP code path (RCCL process group of one rank, bucketed all-reduce on the side stream, collective Dice statistics) vs the plain
# single-process step, same workloads: what the data-parallel plumbing costs before any second GPU exists
run() { python bench.py "$@" --steps 8 --no-also --no-cpu-baseline --no-traffic --no-roofline 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('comm'))"; }
for f in 0 1; do
  if [ $f = 1 ]; then export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MT_FORCE_REDUCER=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517; fi
  echo "== forced reducer: $f"
  echo -n "task009: "; run
  echo -n "task100: "; run --workload task100
  echo -n "resenc bf16: "; run --workload resenc --precision bf16
done
