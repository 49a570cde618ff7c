bash tools/prof_serial.sh; grep "vproj" gpurun_out/stats_serial.txt
python -m pytest tests/test_hip_parity.py tests/test_hip_training.py tests/test_hip_rnn.py -m gpu -x -q > gpurun_out/r3a_pytest.log 2>&1; tail -3 gpurun_out/r3a_pytest.log
bash tools/ab.sh
