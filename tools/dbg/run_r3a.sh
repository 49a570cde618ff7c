python -m pytest tests/test_hip_parity.py tests/test_hip_training.py tests/test_bf16_mode.py tests/test_hip_rnn.py tests/test_module_api.py -m gpu -x -q > gpurun_out/r3a_pytest.log 2>&1; tail -5 gpurun_out/r3a_pytest.log
grep -n "Error\|error\|assert" gpurun_out/r3a_pytest.log | head -20
