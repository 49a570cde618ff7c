timeout 700 python tools/fuzz_parity.py 40 7 < /dev/null > gpurun_out/fuzz.log 2>&1; grep -c "^ok" gpurun_out/fuzz.log; grep "FAIL\|shapes failed" gpurun_out/fuzz.log | head -10
timeout 900 python tools/fuzz_parity.py 60 99 < /dev/null > gpurun_out/fuzz2.log 2>&1; grep -c "^ok" gpurun_out/fuzz2.log; grep "FAIL\|shapes failed" gpurun_out/fuzz2.log | head -10
timeout 600 python -m pytest tests/test_hip_training.py -x -q < /dev/null 2>&1 | tail -2
