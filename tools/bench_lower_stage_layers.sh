# per-layer A/B of kernel families on the lower stages (fp32, batch 2, isolated launches)
cd $GRAFT_REPO_ROOT
run() { python tools/bench_conv.py "$@" --reps 20 2>&1 | grep -v amdgpu.ids | tail -n 1; }
for sh in "12 48 48 120 120" "12 48 48 240 120" "6 24 24 240 240" "6 24 24 480 240" "6 24 24 320 320" "3 12 12 320 320"; do
  set -- $sh
  for mode in fwd bwdw; do
    for w in default off; do          # kernel selection of the problems this process builds (multitalent_amd/ops.py: MT_SELECT)
      echo -n "MT_SELECT=wino=$w,bwdw_wino=$w : "
      MT_SELECT="wino=$w,bwdw_wino=$w" run --mode $mode --cin $4 --cout $5 --shape $1 $2 $3
    done
  done
done
