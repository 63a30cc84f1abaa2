for args in "--cin 32 --cout 32" "--cin 64 --cout 32" "--cin 64 --cout 64 --shape 24 96 96" "--cin 30 --cout 30 --shape 9 18 70"; do
python tools/bench_conv.py --mode bwdw $args --reps 5 --mma 1 | tail -2
python tools/bench_conv.py --mode bwdw $args --reps 5 --mma 0 | tail -1
done
