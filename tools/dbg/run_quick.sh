timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_training.py -x -q > gpurun_out/quick_tests.log 2>&1; tail -3 gpurun_out/quick_tests.log
timeout 600 bash tools/prof_serial.sh > gpurun_out/r3f_serial.log 2>&1; grep "embed\|total kernel" gpurun_out/stats_serial.txt
