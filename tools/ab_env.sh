#!/bin/bash
# A/B of an environment switch over the bench workloads: tools/ab_env.sh VAR v1 v2 ...
var=$1; shift
run() { python bench.py "$@" --steps 8 --no-also --no-cpu-baseline --no-traffic --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for v in "$@"; do
  export $var=$v
  echo "== $var=$v"
  echo -n "task009 fp32: "; run
  echo -n "task100 fp32: "; run --workload task100
  echo -n "resenc fp32: "; run --workload resenc
  echo -n "resenc bf16: "; run --workload resenc --precision bf16
done
