"""Pins the CPU oracle (oracle/vslnet_oracle.py) against golden vectors produced by the reference itself.

CPU-only (`-m "not gpu"`).  Tolerances: SURVEY.md 8c -- fp32 noise floor of the reference is ~1.4e-5 on logits,
so the oracle is gated at 2e-5 on activations/logits and 1e-4*||g||inf + 1e-6 on gradients."""
import numpy as np
import pytest
import torch

from oracle import vslnet_oracle as O
from tests.helpers import load_golden, grad_tol

CASES = ['tiny_tf', 'tiny_rnn', 'real_tf', 'long_tf', 'chardim100_tf', 'wordtable_tf']
ATOL = 2e-5


def _close(a, b, atol=ATOL, what=''):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    finite = np.abs(b) < 1e29
    assert np.array_equal(a[~finite], b[~finite]), what + ': masked entries must be exactly -1e30'
    err = np.abs(a[finite] - b[finite]).max() if finite.any() else 0.0
    atol = atol * max(1.0, float(np.abs(b[finite]).max()) if finite.any() else 1.0)   # relative to the tensor's scale
    assert err <= atol, '%s: max abs err %.3e > %.1e' % (what, err, atol)


@pytest.mark.parametrize('name', CASES)
def test_forward_taps_and_losses(name):
    cfg, P, b, z = load_golden(name)
    want = {}
    with torch.no_grad():
        h, sl, el = O.forward(P, cfg, b['word_ids'], b['char_ids'], b['vfeats'], b['v_mask'], b['q_mask'],
                              training=False, want=want)
        _close(want['video_affine'], z['tap.video_affine.0'], what='video_affine')
        _close(want['embedding_net'], z['tap.embedding_net.0'], what='embedding_net')
        _close(O.word_embedding(P, b['word_ids'], 0.0, False), z['tap.word_emb.0'], what='word_emb')
        _close(O.char_embedding(P, b['char_ids'], 0.0, False), z['tap.char_emb.0'], what='char_emb')
        _close(want['venc'], z['tap.feature_encoder.0'], what='feature_encoder(video)')
        _close(want['qenc'], z['tap.feature_encoder.1'], what='feature_encoder(query)')
        _close(want['venc_parts']['conv_outs'][-1], z['tap.fe_conv_block.0'], what='conv_block(video)')
        _close(want['qenc_parts']['conv_outs'][-1], z['tap.fe_conv_block.1'], what='conv_block(query)')
        _close(want['cq_attention'], z['tap.cq_attention.0'], what='cq_attention')
        _close(want['cq_concat'], z['tap.cq_concat.0'], what='cq_concat')
        _close(want['h_score'], z['tap.highlight_layer.0'], atol=5e-6, what='highlight')
        if cfg.predictor == 'rnn':
            _close(want['pred_parts']['pred_s'], z['tap.pred_start_rnn.0'], what='start rnn')
            _close(want['pred_parts']['pred_e'], z['tap.pred_end_rnn.0'], what='end rnn')
        else:
            _close(want['pred_parts']['pred_s'], z['tap.pred_encoder.0'], what='pred enc 1')
            _close(want['pred_parts']['pred_e'], z['tap.pred_encoder.1'], what='pred enc 2')
        _close(h, z['out.h_score'], atol=5e-6, what='h_score')
        _close(sl, z['out.start_logits'], what='start_logits')
        _close(el, z['out.end_logits'], what='end_logits')
        # masked positions: exactly -1e30 / exactly 0
        pad = b['v_mask'].numpy() == 0
        assert np.all(sl.numpy()[pad] == np.float32(-1e30)) and np.all(h.numpy()[pad] == 0.0)
        _close(O.highlight_loss(h, b['h_labels'], b['v_mask']), z['out.highlight_loss'], atol=1e-5, what='hl loss')
        _close(O.span_loss(sl, el, b['s_labels'], b['e_labels']), z['out.loc_loss'], atol=2e-5, what='loc loss')
        si, ei = O.extract_index(sl, el)
        assert np.array_equal(si.numpy(), z['out.start_index']) and np.array_equal(ei.numpy(), z['out.end_index'])


@pytest.mark.parametrize('name', CASES)
def test_backward_all_parameter_grads(name):
    cfg, P, b, z = load_golden(name)
    P = {k: (v.clone().requires_grad_(k not in O.FROZEN)) for k, v in P.items()}
    total, _ = O.total_loss(P, cfg, b, training=False)
    total.backward()
    n = 0
    for k in z.files:
        if not k.startswith('grad.'):
            continue
        g_ref = z[k]
        g = P[k[5:]].grad
        g = torch.zeros_like(P[k[5:]]) if g is None else g
        err = float(np.abs(g.numpy() - g_ref).max())
        assert err <= grad_tol(g_ref), '%s: grad err %.3e (tol %.3e)' % (k, err, grad_tol(g_ref))
        n += 1
    assert n >= 40


@pytest.mark.parametrize('name', ['train_tf', 'train_rnn'])
def test_training_mode_matches_the_reference_under_seeded_masks(name):
    """TRAINING mode (drop_rate 0.2, the benchmark's mode): the reference ran with every nn.Dropout call replaced by the seeded mask
    O.train_mask(call number, shape, p) (oracle/make_golden.py: run_train_case); the oracle replays the same masks through
    O.force_dropout.  Pins the placement AND the call order of the 41 / 23 dropout sites, on which every GPU training-mode parity
    test (tests/test_hip_training.py) rests."""
    cfg, P, b, z = load_golden(name)
    assert cfg.drop_rate == 0.2
    Pg = {k: (v.clone().requires_grad_(k not in O.FROZEN)) for k, v in P.items()}
    O.force_dropout(O.train_mask)
    try:
        total, (h, sl, el, hl, loc) = O.total_loss(Pg, cfg, b, training=True)
        n_calls = O.DROP_CALLS
    finally:
        O.force_dropout(None)
    assert n_calls == int(z['n_dropout_calls']) == (23 if cfg.predictor == 'rnn' else 41)
    _close(h, z['out.h_score'], atol=5e-6, what='h_score')
    _close(sl, z['out.start_logits'], what='start_logits')
    _close(el, z['out.end_logits'], what='end_logits')
    _close(hl, z['out.highlight_loss'], atol=1e-5, what='hl loss')
    _close(loc, z['out.loc_loss'], atol=2e-5, what='loc loss')
    total.backward()
    worst = 0.0
    for k in z.files:
        if not k.startswith('grad.'):
            continue
        g_ref = z[k]
        g = Pg[k[5:]].grad
        g = torch.zeros_like(Pg[k[5:]]) if g is None else g
        err = float(np.abs(g.numpy() - g_ref).max())
        assert err <= grad_tol(g_ref), '%s: grad err %.3e (tol %.3e)' % (k, err, grad_tol(g_ref))
        worst = max(worst, err / grad_tol(g_ref))
    print('%s: %d dropout calls, worst gradient at %.2f of its gate' % (name, n_calls, worst))
    # the masks matter: with the call numbers shifted by one the logits must move far outside the gate
    O.force_dropout(lambda call, shape, p: O.train_mask(call + 1, shape, p))
    try:
        with torch.no_grad():
            _, (_, sl2, _, _, _) = O.total_loss(P, cfg, b, training=True)
    finally:
        O.force_dropout(None)
    fin = np.abs(z['out.start_logits']) < 1e29
    assert float(np.abs(sl2.numpy() - z['out.start_logits'])[fin].max()) > 1e-2


def test_structural_zero_grads():
    """key bias and the final 1-channel logit biases get structurally zero gradient."""
    cfg, P, b, z = load_golden('tiny_tf')
    for k in ['grad.feature_encoder.attention_block.key.conv1d.bias', 'grad.predictor.start_block.2.conv1d.bias',
              'grad.predictor.end_block.2.conv1d.bias']:
        assert np.abs(z[k]).max() < 1e-6


def test_pad_sensitivity_is_reproduced():
    """SURVEY section 5 note: logits of valid positions DEPEND on the padded length; the oracle must reproduce
    that (it may not mask the depthwise convs or skip padded rows)."""
    cfg, P, b, _ = load_golden('tiny_tf')
    with torch.no_grad():
        _, sl, _ = O.forward(P, cfg, b['word_ids'], b['char_ids'], b['vfeats'], b['v_mask'], b['q_mask'])
        T = b['vfeats'].shape[1]
        vf2 = torch.cat([b['vfeats'], torch.zeros(b['vfeats'].shape[0], 6, b['vfeats'].shape[2])], dim=1)
        vm2 = torch.cat([b['v_mask'], torch.zeros(b['v_mask'].shape[0], 6)], dim=1)
        _, sl2, _ = O.forward(P, cfg, b['word_ids'], b['char_ids'], vf2, vm2, b['q_mask'])
    valid = b['v_mask'].bool()
    assert float((sl2[:, :T][valid] - sl[valid]).abs().max()) > 1e-4


def test_host_helpers_against_reference():
    import os
    from tests.helpers import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'host_helpers.npz'))
    lens = torch.from_numpy(z['lens'])
    assert np.array_equal(O.convert_length_to_mask(lens).numpy(), z['mask'])
    h = O.highlight_labels(z['s_labels'], z['e_labels'], z['vlens'], int(z['vlens'].max()))
    assert np.array_equal(h.numpy(), z['h_labels'])


def test_init_matches_reference_under_the_same_seed():
    """a19 (VSLNet_t7.py:42-50): tests/golden/init.npz holds checksums of the reference's freshly initialised state_dict
    (constructed under torch.manual_seed; oracle/make_golden.py run_init).  The build's module -- same sub-module order, same
    initialisers, LSTM.reset_parameters() for the rnn head -- must reproduce every tensor bit for bit under the same seed."""
    import os
    from tests.helpers import GOLDEN
    from vslnet_amd.model.VSLNet import VSLNet
    z = np.load(os.path.join(GOLDEN, 'init.npz'), allow_pickle=False)
    for pred in ('transformer', 'rnn', 'transformer_wordtable'):
        cfg = O.make_cfg(video_feature_dim=64, max_pos_len=32, word_size=52, predictor=pred.split('_')[0])
        glove = None if pred.endswith('wordtable') else np.random.RandomState(0).randn(cfg.word_size - 2, cfg.word_dim).astype(np.float32)
        torch.manual_seed(int(z['seed']))
        sd = VSLNet(cfg, glove).state_dict()
        assert list(sd.keys()) == [str(k) for k in z['keys.' + pred]]
        for k, v in sd.items():
            v64 = v.detach().double()
            got = np.array([float(v64.sum()), float(v64.abs().sum()), float(v64.flatten()[-1])])
            assert np.array_equal(got, z['%s.%s' % (pred, k)]), (pred, k, got, z['%s.%s' % (pred, k)])
