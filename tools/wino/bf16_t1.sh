for v in "" _bf1 _bf4 _bf8 _bf5 _bf13; do echo "variant=$v"; 
MT_BF16_CFG=4 MT_BF16_VEC2=1 MT_LIB_VARIANT=libmtseg_hip$v.so python tools/bench_conv.py --mode fwd --cin 32 --cout 32 --reps 5 --mma 1 2>&1 | tail -1
MT_BF16_CFG=4 MT_BF16_VEC2=1 MT_LIB_VARIANT=libmtseg_hip$v.so python tools/bench_conv.py --mode fwd --cin 64 --cout 32 --reps 5 --mma 1 2>&1 | tail -1
MT_BF16_CFG=3 MT_BF16_VEC2=1 MT_LIB_VARIANT=libmtseg_hip$v.so python tools/bench_conv.py --mode fwd --cin 64 --cout 32 --reps 5 --mma 1 2>&1 | tail -1
done
