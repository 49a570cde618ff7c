python -m pytest tests/test_hip_parity.py tests/test_hip_training.py -m gpu -x -q > gpurun_out/r3a_pytest.log 2>&1; tail -3 gpurun_out/r3a_pytest.log
grep -n "Error\|error\|assert\|mismatch" gpurun_out/r3a_pytest.log | head -10
bash tools/bench_shapes.sh 2>&1 | head -4
bash tools/prof_serial.sh --batch 16 --T 1024; head -8 gpurun_out/stats_serial.txt
