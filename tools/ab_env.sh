#!/bin/bash
# A/B of an environment switch over the bench workloads: tools/ab_env.sh VAR v1 v2 ...   (WL="task009 resenc_bf16" restricts the workloads)
# Kernel families (ABI 4: no switches inside the library): tools/ab_env.sh MT_SELECT default x16=off wino=off "x16=force,tapsplit=off"
# (fields: wino | m16 | x16 | tapsplit | bwdw_wino | bwdw_tr16 = default | off | force; bwdw_cw = 4 | 2 | 1 | 104; read by multitalent_amd/ops.py)
var=$1; shift
WL=${WL:-"task009 task100 resenc resenc_bf16"}
run() { python bench.py "$@" --steps 8 --no-also --no-cpu-baseline --no-traffic --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for v in "$@"; do
  if [ "$var" = MT_SELECT ] && [ "$v" = default ]; then unset MT_SELECT; else export $var=$v; fi
  echo "== $var=$v"
  for w in $WL; do
    case $w in
      task009) echo -n "task009 fp32: "; run ;;
      task100) echo -n "task100 fp32: "; run --workload task100 ;;
      resenc) echo -n "resenc fp32: "; run --workload resenc ;;
      resenc_bf16) echo -n "resenc bf16: "; run --workload resenc --precision bf16 ;;
      task009_bf16) echo -n "task009 bf16: "; run --precision bf16 ;;
    esac
  done
done
