"""The library keeps two switches that force a variant the shapes of the suite would not select: VSL_MULTI_STREAM=0 (every kernel on the
caller's stream: the isolated-kernel profiling mode of tools/prof_serial.sh) and VSL_LSTM1=0 (the 4-sample MFMA-group LSTM that serves B > 256,
at the suite's small batches).  They are read once per process, so the parity suite is re-run once in a child process with both set.
(Round 4 removed the A/B switches of earlier rounds -- the fp32-input GEMM / weight-gradient kernels, the 8-wave attention block override, the
LDS-free weight gradient for fp32 jobs: baselines for A/B runs come from git revisions, tools/build_base.py.  The variants that remain are all
selected by SHAPE and covered by the shapes of the suite: k_attn_fwd + k_attn_out_fwd and k_attn_bwd_long for L > 256, the 8-wave attention
block for 128 < L <= 256, k_wgrad3 for bfloat16 features, k_linear_fwd for embedding widths that are not multiples of 16, k_loss_a / b / c
when the caller leaves the mask sum to the device.)

VSL_RNN_FUSED=0 is the third: the rnn head as chunked launches over three streams -- what batches of 81 .. 256 samples take -- instead of the
one-launch dataflow pipeline (k_rnn_fwd / k_rnn_bwd) every shape of the suite selects.

Round 5 added two of the same kind: VSL_QKV_FUSED=0 (k_qkv_bwd as its own launch instead of inside the conv block's backward kernel: what ragged row
tiles and L > 256 take) and VSL_HEADS_FUSED=0 (k_head_fwd instead of the tail of the second predictor pass' attention-block kernel: what T > 128
takes) -- the training suite's whole-tile shapes select the fused paths, so it is re-run once with both off.

Round 6: VSL_QUERY_FUSED=0 -- the query branch as row-tile launches (linear_fwd + convblock_fwd<0> + attn_block_fwd / attn_out_bwd + attn_bwd +
convblock_bwd<0>) instead of the sample-local k_query_fwd / k_query_bwd that every Lq <= 32 shape now selects; what Lq > 32 takes.  Same re-run.
Later in round 6: VSL_CQ_FOLD=0 (k_cq_col as its own launch: what T > 128 or Lq > 32 takes) and VSL_LOSS_INLINE=0 / VSL_FUSED_TAIL=1 (the lazy loss in front
of the heads' backward; the final reduction and AdamW as one launch), the last two also through tests/test_fused_loss.py and tests/test_optimizer.py, which drive
those paths.  (The variants round 6 replaced outright left no switch: k_attn_block_fwd<2> and k_attn_bwd_fused<256> for 128 < L <= 256, 256-row chunks for the step's
last weight-gradient batch, the unmerged tail batch -- their A/B numbers are in profiles/r06_notes.md.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parity_suite_single_stream_and_forced_lstm_groups():
    e = dict(os.environ, VSL_MULTI_STREAM='0', VSL_LSTM1='0')
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider',
                        'tests/test_hip_parity.py', 'tests/test_hip_rnn.py', 'tests/test_bf16_mode.py'], cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_rnn_suite_with_the_chunked_launches():
    e = dict(os.environ, VSL_RNN_FUSED='0')
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider', 'tests/test_hip_rnn.py'],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_training_suite_with_the_unfused_launches():
    e = dict(os.environ, VSL_QKV_FUSED='0', VSL_HEADS_FUSED='0', VSL_QUERY_FUSED='0', VSL_CQ_FOLD='0', VSL_LOSS_INLINE='0')
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider', 'tests/test_hip_training.py', 'tests/test_hip_parity.py',
                        'tests/test_fused_loss.py'],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
