cd $GRAFT_REPO_ROOT
for m in default off; do          # MT_SELECT: kernel selection of the problems this process builds (multitalent_amd/ops.py)
  echo "MT_SELECT=tapsplit=$m"
  MT_SELECT=tapsplit=$m python tools/bench_conv.py --mode bwdd --cin 240 --cout 320 --shape 6 24 24 --stride 2 2 2 --reps 20 2>&1 | tail -n 2
  MT_SELECT=tapsplit=$m python tools/bench_conv.py --mode bwdd --cin 320 --cout 320 --shape 3 12 12 --stride 1 2 2 --reps 20 2>&1 | tail -n 2
  MT_SELECT=tapsplit=$m python tools/bench_conv.py --mode fwd --cin 240 --cout 320 --shape 6 24 24 --stride 2 2 2 --reps 20 2>&1 | tail -n 2
  MT_SELECT=tapsplit=$m python tools/bench_conv.py --mode fwd --cin 320 --cout 320 --shape 3 12 12 --stride 1 2 2 --reps 20 2>&1 | tail -n 2
done
