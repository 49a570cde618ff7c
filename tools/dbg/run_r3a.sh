python -m pytest tests/test_hip_parity.py tests/test_hip_training.py tests/test_hip_rnn.py -m gpu -x -q > gpurun_out/r3a_pytest.log 2>&1; tail -5 gpurun_out/r3a_pytest.log
grep -n "Error\|error\|assert" gpurun_out/r3a_pytest.log | head -20
bash tools/ab.sh 2>&1
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --profile-all 2>&1 >/dev/null | head -12
