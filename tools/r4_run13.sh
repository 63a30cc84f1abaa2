mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/run13_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/run13_tests.log
tail -5 gpurun_out/run13_tests.log
timeout 2400 bash tools/measure_r4.sh > gpurun_out/run13_measure.log 2>&1
tail -120 gpurun_out/run13_measure.log
