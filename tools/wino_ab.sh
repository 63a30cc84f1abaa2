#!/bin/bash
# A/B of the persistent Winograd kernels on the layers of the benchmark network: MT_WINO_DMA=0 (registers) vs 1 (LDS-DMA)
for cfg in "30 30 48 192 192" "60 30 48 192 192" "60 60 24 96 96" "120 60 24 96 96" "120 120 12 48 48" "240 120 12 48 48"; do
  set -- $cfg
  for dma in 0 1; do
    echo -n "cin $1 cout $2 @ $3x$4x$5 dma=$dma: "
    MT_WINO_DMA=$dma python tools/bench_conv.py --cin $1 --cout $2 --shape $3 $4 $5 --reps 10 2>&1 | tail -1
  done
done
