python -m pytest tests/test_hip_parity.py tests/test_hip_training.py -m gpu -x -q > gpurun_out/r3a_pytest.log 2>&1; tail -3 gpurun_out/r3a_pytest.log
grep -n "Error\|error\|assert\|mismatch" gpurun_out/r3a_pytest.log | head -10
bash tools/prof_serial.sh; head -34 gpurun_out/stats_serial.txt | tail -31
bash tools/ab.sh
