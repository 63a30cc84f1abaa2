python tools/bench_conv.py --mode bwdw --cin 32 --cout 32 --reps 5 --mma 1 2>&1 | tail -2
python tools/bench_conv.py --mode bwdw --cin 64 --cout 64 --shape 24 96 96 --reps 5 --mma 1 2>&1 | tail -2
