mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/run33_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/run33_tests.log
tail -4 gpurun_out/run33_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
