mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/run26_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/run26_tests.log
tail -4 gpurun_out/run26_tests.log
timeout 2700 bash tools/measure_r4.sh > gpurun_out/run26_measure.log 2>&1
tail -100 gpurun_out/run26_measure.log
