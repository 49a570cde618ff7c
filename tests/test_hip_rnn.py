"""rnn predictor (a15, DynamicRNN layers_t7.py:302-313 + the rnn branch :341-343) on the GPU: the persistent LSTM kernels
against the reference's own outputs (golden case tiny_rnn) and against the oracle on a batch that is not a multiple of the
16-sample workgroup."""
import numpy as np
import pytest
import torch

from oracle import vslnet_oracle as O
from tests.helpers import assert_forced_relu_inside_noise
from tests.helpers import load_golden

pytestmark = pytest.mark.gpu


def _dev(t):
    return t.cuda().contiguous()


def _run(cfg, P, b):
    from vslnet_amd.engine import Engine, flat_from_state_dict
    eng = Engine(cfg)
    flat = flat_from_state_dict(eng, P)
    h, sl, el = eng.forward(flat, _dev(P['embedding_net.word_emb.pad_vec']), _dev(P['embedding_net.word_emb.glove_vec']),
                            _dev(b['word_ids']), _dev(b['char_ids']), _dev(b['vfeats']), _dev(b['v_mask']), _dev(b['q_mask']),
                            training=False, seed=0)
    losses, d_h, d_sl, d_el = eng.loss(_dev(b['s_labels']), _dev(b['e_labels']), _dev(b['h_labels']), 1.0, 5.0)
    grads = torch.full((eng.param_floats,), float('nan'), device=eng.device)
    eng.backward(d_h, d_sl, d_el, grads)
    torch.cuda.synchronize()
    return eng, h, sl, el, losses, grads


def _oracle(cfg, P, b, eng=None):
    """oracle forward + backward; with `eng`, on the ReLU branches the GPU path took (tests/helpers.py: relu_flips)"""
    want = {}
    if eng is not None:
        from tests.helpers import relu_flips
        B, T = b['v_mask'].shape
        O.record_relu_signs()
        with torch.no_grad():
            O.forward(P, cfg, b['word_ids'], b['char_ids'], b['vfeats'], b['v_mask'], b['q_mask'])
        want['flips'], masks = relu_flips(eng, B, T, b['q_mask'].shape[1], predictor='rnn')
        O.record_relu_signs(False)
        O.force_relu_signs(masks)
    Pg = {k: v.clone().requires_grad_(k not in O.FROZEN) for k, v in P.items()}
    oh, osl, oel = O.forward(Pg, cfg, b['word_ids'], b['char_ids'], b['vfeats'], b['v_mask'], b['q_mask'], want=want)
    O.force_relu_signs(None)
    if eng is not None:
        assert_forced_relu_inside_noise(O)
    total = O.span_loss(osl, oel, b['s_labels'], b['e_labels']) + 5.0 * O.highlight_loss(oh, b['h_labels'], b['v_mask'])
    total.backward()
    return Pg, want, oh.detach(), osl.detach(), oel.detach(), float(total.detach())


def _check_grads(eng, grads, ref, bad, flips=0):
    for k, t in eng.views(grads).items():
        r = ref(k)
        err = float((t.cpu() - r).abs().max())
        tol = 1e-4 * float(r.abs().max()) + 1e-6
        if not err <= tol:
            bad.append((k, err, tol))


def test_rnn_head_matches_reference_golden():
    cfg, P, b, z = load_golden('tiny_rnn')
    assert cfg.predictor == 'rnn'
    eng, h, sl, el, losses, grads = _run(cfg, P, b)
    names = [n for n, _, _, _ in eng.layout]
    assert 'predictor.start_encoder.lstm.weight_hh_l0' in names and not any(n.startswith('predictor.encoder.') for n in names)
    fin = np.abs(z['out.start_logits']) < 1e29
    assert float(np.abs(sl.cpu().numpy() - z['out.start_logits'])[fin].max()) <= 1e-4
    assert float(np.abs(el.cpu().numpy() - z['out.end_logits'])[fin].max()) <= 1e-4
    assert np.all(sl.cpu().numpy()[~fin] == np.float32(-1e30))
    assert float(np.abs(h.cpu().numpy() - z['out.h_score']).max()) <= 2e-5
    assert abs(float(losses[0]) - float(z['out.loc_loss'])) <= 2e-5 * max(1.0, abs(float(z['out.loc_loss'])))
    si, ei = eng.extract_index(sl, el)
    assert np.array_equal(si.cpu().numpy(), z['out.start_index']) and np.array_equal(ei.cpu().numpy(), z['out.end_index'])
    Pg, want, _, _, _, _ = _oracle(cfg, P, b, eng)
    bad = []
    golden = lambda k: torch.from_numpy(z['grad.' + k])                                   # noqa: E731  the reference's own gradients,
    forced = lambda k: Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(Pg[k])  # noqa: E731  or the oracle on the GPU branch
    _check_grads(eng, grads, forced if want['flips'] else golden, bad, want['flips'])
    assert not bad, (want['flips'], bad[:6])
    # the LSTM outputs themselves (masked h sequences) against the oracle's taps
    B, T = b['v_mask'].shape
    for nm in ('pred_s', 'pred_e'):
        assert float((eng.ws_view(nm, (B, T, 128)).cpu() - want['pred_parts'][nm].detach()).abs().max()) <= 1e-5, nm


@pytest.mark.parametrize('shape', [dict(B=21, T=37, Lq=6, Lc=5), dict(B=3, T=128, Lq=20, Lc=10),
                                   dict(B=259, T=9, Lq=3, Lc=4),                  # B > 256: the 4-sample MFMA-group LSTM kernels (ragged last group)
                                   dict(B=16, T=128, Lq=20, Lc=10, Dv=1024)])     # the last one = BASELINE configs[0] as written
def test_rnn_head_against_oracle(shape):
    cfg = O.make_cfg(video_feature_dim=shape.get('Dv', 64), max_pos_len=128, word_size=60, predictor='rnn')
    P = O.random_params(cfg, seed=21)
    b = O.synthetic_batch(cfg, shape['B'], shape['T'], shape['Lq'], shape['Lc'], seed=22, ragged=True)
    eng, h, sl, el, losses, grads = _run(cfg, P, b)
    Pg, want, oh, osl, oel, total = _oracle(cfg, P, b, eng)
    fin = osl.abs() < 1e29
    scale = max(1.0, float(osl[fin].abs().max()))
    assert float((sl.cpu() - osl)[fin].abs().max()) <= 1e-4 * scale
    assert float((el.cpu() - oel)[fin].abs().max()) <= 1e-4 * scale
    assert abs(float(losses[2]) - total) <= 1e-4 * max(1.0, abs(total))
    bad = []
    _check_grads(eng, grads, lambda k: Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(Pg[k]), bad, want['flips'])
    assert not bad, (want['flips'], bad[:6])
