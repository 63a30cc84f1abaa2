export MT_WINO_WAVES=8
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_properties_gpu.py tests/test_golden_gpu.py -m gpu -q -k "winograd or conv_fwd or adjoint or golden or two_training or resenc" 2>&1 | tail -3
for w in 8 4; do
MT_WINO_WAVES=$w python tools/bench_conv.py --mode fwd --cin 32 --cout 32 --reps 5 | tail -1
MT_WINO_WAVES=$w python tools/bench_conv.py --mode fwd --cin 64 --cout 32 --reps 5 | tail -1
MT_WINO_WAVES=$w python tools/bench_conv.py --mode fwd --cin 64 --cout 64 --shape 24 96 96 --reps 5 | tail -1
done
