timeout 1500 python -m pytest tests -m gpu -x -q < /dev/null > gpurun_out/quick_tests.log 2>&1; tail -2 gpurun_out/quick_tests.log
