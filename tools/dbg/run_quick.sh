timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/quick_tests.log 2>&1; tail -15 gpurun_out/quick_tests.log
