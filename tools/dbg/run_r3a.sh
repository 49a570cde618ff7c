python -m pytest tests/test_hip_parity.py tests/test_hip_training.py tests/test_hip_rnn.py tests/test_bf16_mode.py -m gpu -x -q -s 2>&1 | grep "relu-noise\|passed\|failed" | sort | uniq -c | sort -k7 -g | tail -12
echo "--- fp32 kernels"
VSL_F32_GEMM=1 VSL_WGRAD_F32=1 python -m pytest tests/test_hip_parity.py tests/test_hip_training.py -m gpu -x -q -s 2>&1 | grep "relu-noise\|passed\|failed" | sort | uniq -c | sort -k7 -g | tail -6
