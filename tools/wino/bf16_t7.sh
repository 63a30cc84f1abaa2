python tools/list_kernels.py bf16 | grep "(6, 24, 24)\|(3, 12, 12)"
python tools/bench_conv.py --mode fwd --cin 240 --cout 240 --shape 6 24 24 --reps 10 --mma 1 | tail -3
python tools/bench_conv.py --mode fwd --cin 240 --cout 240 --shape 6 24 24 --reps 10 --mma 0 | tail -1
python tools/bench_conv.py --mode fwd --cin 480 --cout 240 --shape 6 24 24 --reps 10 --mma 1 | tail -3
python tools/bench_conv.py --mode fwd --cin 480 --cout 240 --shape 6 24 24 --reps 10 --mma 0 | tail -1
python bench.py --precision bf16 --no-cpu-baseline --no-roofline --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-200
