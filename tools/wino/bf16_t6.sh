for args in "--cin 32 --cout 64 --stride 2 2 2" "--cin 30 --cout 60 --stride 2 2 2" "--cin 64 --cout 128 --shape 24 96 96 --stride 2 2 2" "--cin 256 --cout 320 --shape 6 24 24 --stride 2 2 2"; do
python tools/bench_conv.py --mode fwd $args --reps 5 --mma 1 | tail -3
python tools/bench_conv.py --mode fwd $args --reps 5 --mma 0 | tail -1
done
