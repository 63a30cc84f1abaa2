# where does conv_tapsplit_kernel's time go?  ablation variants (tools/build_conv_variant.sh tsablN -DTS_ABL=N) on the low-resolution layers
cd $GRAFT_REPO_ROOT
for v in "" tsabl1 tsabl2 tsabl4 tsabl7; do
  [ -n "$v" ] && export MT_LIB_VARIANT=libmtseg_hip_$v.so
  echo "== variant ${v:-default}"
  python tools/bench_conv.py --mode fwd --cin 320 --cout 320 --shape 3 12 12 --reps 30 2>&1 | grep -v amdgpu.ids | tail -n 1
  python tools/bench_conv.py --mode fwd --cin 320 --cout 320 --shape 6 12 12 --reps 30 2>&1 | grep -v amdgpu.ids | tail -n 1
  python tools/bench_conv.py --mode fwd --cin 240 --cout 320 --shape 6 24 24 --stride 2 2 2 --reps 30 2>&1 | grep -v amdgpu.ids | tail -n 1
done
