# phase ablations of the 8-wave Winograd kernels (persistent / one tile per workgroup) on 30 -> 30 @ 2 x 48x192x192
# WN_ABL bits: 1 no staging, 2 no input transform, 4 no MFMA, 8 no output transform
for v in "" _wn1 _wn2 _wn3 _wn4 _wn8 _wn12 _wn15; do
  for pers in 1 0; do
    echo -n "variant=${v:-full} persist=$pers  "
    MT_WINO_PERSIST=$pers MT_LIB_VARIANT=libmtseg_hip$v.so python tools/bench_conv.py --mode fwd --cin ${CIN:-30} --cout ${COUT:-30} --reps 5 --lazy 1 2>&1 | tail -1
  done
done
