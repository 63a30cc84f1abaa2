#!/bin/bash
# A/B of the tile order (MT_TILE_ORDER 1 = default library, 0 = libmtseg_hip_ord0.so built by tools/build_variant.sh ord0 -DMT_TILE_ORDER=0)
run() { python bench.py "$@" --steps 8 --no-also --no-cpu-baseline --no-traffic --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for v in libmtseg_hip.so libmtseg_hip_ord0.so; do
  export MT_LIB_VARIANT=$v
  echo "== $v"
  echo -n "task009 fp32: "; run
  echo -n "task100 fp32: "; run --workload task100
  echo -n "resenc fp32: "; run --workload resenc
  echo -n "resenc bf16: "; run --workload resenc --precision bf16
  echo -n "task009 bf16: "; run --precision bf16
done
