"""The library keeps the previous implementation behind every fused / newer kernel (environment switches, README).  They are read once per
process, so the parity suite is re-run in a child process with ALL of them switched to the older path: per-layer conv kernels, LDS-staged
weight gradient, two-kernel attention forward, 16-sample LSTM workgroups, hipEventRecord ordering, two-kernel loss."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OLD_PATHS = dict(VSL_CONVBLOCK='0', VSL_CONVBLOCK_BWD='0', VSL_WGRAD2='0', VSL_ATTN_BLOCK='0', VSL_LSTM1='0', VSL_LSTM4='0',
                 VSL_STOP_EVENTS='0', VSL_LOSS_FUSED='0')


@pytest.mark.parametrize('env', [OLD_PATHS, dict(VSL_MULTI_STREAM='0'), dict(VSL_LSTM1='0'), dict(VSL_ATTN_WAVES='8')],
                         ids=['all-previous-kernels', 'single-stream', 'lstm-4-sample-groups', 'attention-8-waves'])
def test_parity_suite_on_the_previous_kernels(env):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider',
                        'tests/test_hip_parity.py', 'tests/test_hip_rnn.py'], cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
