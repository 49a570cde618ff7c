"""The rnn head as one launch per direction (kernels_lstm.hip: k_rnn_fwd / k_rnn_bwd, round 4): three workgroups per sample hand every
time step over through tagged write-through granules.  The parity of the path is tests/test_hip_rnn.py's (every shape there takes it); what
only this file checks is the hand-off protocol itself, the way MI355X_MICROARCH.md asks for it -- under uneven load, with the workspace
poisoned between runs, every output word compared:

* 64 ragged samples x 128 steps (192 workgroups: three quarters of the chip polling or publishing), twelve forward + backward passes with a
  GEMM stream hammering the other CUs, the caller's workspace overwritten with noise before each: logits and the whole gradient bucket must
  come out bit-identical every time.  A granule read before its producer wrote it would carry an old value (different noise, different tag)
  -- or stall until the spin limit and surface as NaN.
* the same batch through the chunked launches (VSL_RNN_FUSED=0, a child process): the two paths differ only in how the end LSTM's input
  projection is computed (fp32 FMAs here, the bf16x6 GEMM there), so logits and gradients agree to fp32 rounding.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import vslnet_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(B=64, T=128):
    from vslnet_amd.engine import Engine, flat_from_state_dict
    cfg = O.make_cfg(video_feature_dim=256, max_pos_len=T, word_size=52, predictor='rnn', drop_rate=0.2)
    P = O.random_params(cfg, seed=3)
    eng = Engine(cfg)
    flat = flat_from_state_dict(eng, P)
    d = {k: v.cuda().contiguous() for k, v in O.synthetic_batch(cfg, B=B, T=T, Lq=12, Lc=8, seed=5, ragged=True).items()}
    return cfg, eng, flat, P, d


def _step(eng, flat, P, d, seed):
    pad, glove = P['embedding_net.word_emb.pad_vec'].cuda(), P['embedding_net.word_emb.glove_vec'].cuda()
    h, sl, el = eng.forward(flat, pad, glove, d['word_ids'], d['char_ids'], d['vfeats'], d['v_mask'], d['q_mask'], training=True, seed=seed)
    _, d_h, d_sl, d_el = eng.loss(d['s_labels'], d['e_labels'], d['h_labels'], 1.0, 5.0)
    g = torch.full((eng.param_floats,), float('nan'), device='cuda')
    eng.backward(d_h, d_sl, d_el, g)
    real = torch.zeros(eng.param_floats, dtype=torch.bool, device='cuda')      # (the alignment pad floats between tensors are never written)
    for _, off, numel, _ in eng.layout:
        real[off:off + numel] = True
    return sl, el, g[real]


def test_granule_handoffs_are_exact_under_uneven_load_and_a_poisoned_workspace():
    cfg, eng, flat, P, d = _setup()
    side = torch.cuda.Stream()
    a = torch.randn(2048, 2048, device='cuda')
    ref = None
    for rep in range(12):
        if eng._ws is not None:                    # the caller-owned workspace (granule buffers included): noise, different every pass
            eng._ws.normal_(std=float(10 ** (rep % 5)))
        with torch.cuda.stream(side):              # uneven load: a GEMM stream on whatever CUs it gets, of varying length
            for _ in range(1 + rep % 4):
                a = torch.tanh(a @ a * 1e-3)
        sl, el, g = _step(eng, flat, P, d, seed=11)
        torch.cuda.synchronize()
        out = (sl.clone(), el.clone(), g.clone())
        assert bool(torch.isfinite(out[2]).all()), 'rep %d: a spin limit was hit (NaN gradients)' % rep
        if ref is None:
            ref = out
        else:
            for name, x, y in zip(('start logits', 'end logits', 'gradients'), out, ref):
                assert torch.equal(x, y), 'rep %d: %s differ from the first pass (%d words)' % (rep, name, int((x != y).sum()))


_CHILD = r'''
import json, sys, torch
sys.path.insert(0, %r)
from tests.test_rnn_fused import _setup, _step
cfg, eng, flat, P, d = _setup(B=24, T=96)
sl, el, g = _step(eng, flat, P, d, seed=11)
torch.cuda.synchronize()
torch.save({'sl': sl.cpu(), 'el': el.cpu(), 'g': g.cpu()}, sys.argv[1])
'''


def test_fused_launch_agrees_with_the_chunked_launches(tmp_path):
    outs = []
    for fused in ('1', '0'):
        f = str(tmp_path / ('fused%s.pt' % fused))
        r = subprocess.run([sys.executable, '-c', _CHILD % ROOT, f], env=dict(os.environ, VSL_RNN_FUSED=fused), cwd=ROOT, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(torch.load(f))
    a, b = outs
    fin = torch.isfinite(a['sl']) & (a['sl'].abs() < 1e29)
    assert float((a['sl'] - b['sl'])[fin].abs().max()) <= 2e-5 and float((a['el'] - b['el'])[fin].abs().max()) <= 2e-5
    scale = float(b['g'].abs().max())
    err = float((a['g'] - b['g']).abs().max())
    print('[fused vs chunked] logits %.2e, gradients %.2e of max |g| %.2e' % (float((a['sl'] - b['sl'])[fin].abs().max()), err, scale))
    assert err <= 2e-5 * scale


def test_granule_tags_stay_unique_across_the_21_bit_wrap():
    """the launch whose counter value is 0 mod 2^21 used to be remapped onto the tag of the launch AFTER it; two forward-only launches
    (eval / inference loops: no backward in between) then shared a tag and the second one's consumers could take the first one's granules as valid.
    Forward-only launches on alternating inputs across the wrap must reproduce what the same inputs give far away from it, bit for bit."""
    from vslnet_amd.engine import Engine, flat_from_state_dict
    cfg = O.make_cfg(video_feature_dim=64, max_pos_len=32, word_size=52, predictor='rnn')
    P = O.random_params(cfg, seed=7)
    eng = Engine(cfg)
    flat = flat_from_state_dict(eng, P)
    pad, glove = P['embedding_net.word_emb.pad_vec'].cuda(), P['embedding_net.word_emb.glove_vec'].cuda()
    batches = [{k: v.cuda().contiguous() for k, v in O.synthetic_batch(cfg, B=8, T=32, Lq=6, Lc=6, seed=s, ragged=True).items()} for s in (21, 22)]

    def fwd(d):
        h, sl, el = eng.forward(flat, pad, glove, d['word_ids'], d['char_ids'], d['vfeats'], d['v_mask'], d['q_mask'])
        torch.cuda.synchronize()
        return torch.stack([h, sl, el]).clone()

    prev = eng.lib.vsl_debug_rnn_launches(1000)
    try:
        ref = [fwd(batches[0]), fwd(batches[1])]
        assert not torch.equal(ref[0], ref[1])
        eng.lib.vsl_debug_rnn_launches((1 << 21) - 3)
        for i in range(8):                         # counter values 2^21 - 2 .. 2^21 + 5: the wrap sits in the middle
            out = fwd(batches[i & 1])
            assert torch.equal(out, ref[i & 1]), 'launch %d behind the wrap differs' % i
        assert eng.lib.vsl_debug_rnn_launches(0) == (1 << 21) + 5
    finally:
        eng.lib.vsl_debug_rnn_launches(max(prev, 1 << 22))
