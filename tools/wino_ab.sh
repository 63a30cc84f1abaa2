# A/B of conv_wino8p_kernel variants inside ONE gpurun call (boxes differ by ~1 %): tools/wino_ab.sh "" _base ...  (MT_LIB_VARIANT suffixes)
for v in "${@:-_base ""}"; do echo "== variant=$v";
for cfg in "30 30 48 192 192" "60 30 48 192 192" "60 60 24 96 96" "120 60 24 96 96" "120 120 12 48 48" "240 120 12 48 48"; do set -- $cfg
MT_LIB_VARIANT=libmtseg_hip$v.so python tools/bench_conv.py --mode fwd --cin $1 --cout $2 --shape $3 $4 $5 --reps 10 --lazy 1 2>&1 | tail -1; done
MT_LIB_VARIANT=libmtseg_hip$v.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
