for v in "" _wn15 _wn16 _wn20; do
  echo -n "variant=${v:-full} persist=1  "
  MT_WINO_PERSIST=1 MT_LIB_VARIANT=libmtseg_hip$v.so python tools/bench_conv.py --mode fwd --cin ${CIN:-30} --cout ${COUT:-30} --reps 5 --lazy 1 2>&1 | tail -1
done
