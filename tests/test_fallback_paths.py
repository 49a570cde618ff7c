"""The library keeps an A/B switch for the split-bf16 kernels of round 3 (VSL_F32_GEMM / VSL_WGRAD_F32 = 1: the fp32-input MFMA kernels of
round 2) and a few shape-selected variants that can be forced on (VSL_WGRAD4=0: the LDS-free split weight gradient, which still serves
the bf16-feature jobs, for every job).  The switches are read once per process, so the parity suite is re-run
in a child process per setting.  (The superseded kernels of rounds 1-2 -- per-layer conv kernels, LDS-staged weight gradient, 16-sample
LSTM workgroups -- are gone; baselines for A/B runs come from git revisions: tools/build_base.py.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32_PATHS = dict(VSL_F32_GEMM='1', VSL_WGRAD_F32='1', VSL_STOP_EVENTS='0')


@pytest.mark.parametrize('env', [F32_PATHS, dict(VSL_MULTI_STREAM='0', VSL_LSTM1='0', VSL_ATTN_WAVES='8', VSL_WGRAD4='0')],
                         ids=['fp32-input-mfma-kernels', 'single-stream-and-forced-variants'])
def test_parity_suite_on_the_previous_kernels(env):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider',
                        'tests/test_hip_parity.py', 'tests/test_hip_rnn.py', 'tests/test_bf16_mode.py'], cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
