"""vsl_io.fused_loss (round 6): the loss folded into vsl_backward (`Engine.loss(lazy=True)` + `backward()` on its seeds).

For T >= 32 (a row tile then touches at most two samples) with the caller's mask sum the loss kernel leaves the dependent chain -- it runs behind the
main stream's last kernel -- and the
span heads' backward and the highlight layer's backward compute their seeds from the logits themselves (the same expressions: tile_bodies.hpp
loss_ce_seed / loss_hl_seed, kernels_bwd.hip loss_sample_lse).  Whatever the path: the same losses, the same seeds, the same gradients as
loss() followed by backward()."""
import pytest
import torch

from oracle import vslnet_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,T,predictor,mask_sum_given', [(4, 64, 'transformer', True), (3, 128, 'transformer', True), (5, 40, 'transformer', True),
                                                          (4, 64, 'transformer', False), (4, 64, 'rnn', True), (2, 320, 'transformer', True), (6, 24, 'transformer', True), (3, 117, 'transformer', True)])
def test_lazy_loss_rides_in_the_backward(B, T, predictor, mask_sum_given):
    from vslnet_amd.engine import Engine, flat_from_state_dict
    cfg = O.make_cfg(video_feature_dim=64, max_pos_len=max(T, 16), word_size=52, predictor=predictor, drop_rate=0.2)
    P = O.random_params(cfg, seed=5)
    eng = Engine(cfg)
    d = {k: v.cuda().contiguous() for k, v in O.synthetic_batch(cfg, B=B, T=T, Lq=7, Lc=6, seed=6, ragged=True).items()}
    pad, glove = P['embedding_net.word_emb.pad_vec'].cuda(), P['embedding_net.word_emb.glove_vec'].cuda()
    flat = flat_from_state_dict(eng, P)
    msum = float(d['v_mask'].sum()) if mask_sum_given else 0.0
    out = []
    for lazy in (False, True):
        eng.forward(flat, pad, glove, d['word_ids'], d['char_ids'], d['vfeats'], d['v_mask'], d['q_mask'], training=True, seed=9)
        losses, d_h, d_sl, d_el = eng.loss(d['s_labels'], d['e_labels'], d['h_labels'], 1.0, 5.0, inv_batch=1.0 / B, mask_sum=msum, lazy=lazy)
        if lazy:
            assert eng._pending_loss is not None                       # nothing launched yet
        g = eng.backward(d_h, d_sl, d_el, eng.new_flat())
        assert eng._pending_loss is None
        torch.cuda.synchronize()
        out.append([t.clone() for t in (losses, d_h, d_sl, d_el, g)])
    for a, b, nm in zip(out[0], out[1], ('losses', 'd_h', 'd_start', 'd_end', 'grads')):
        assert torch.isfinite(b).all(), nm
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= 2e-6 * scale + 1e-12, (nm, float((a - b).abs().max()), scale)
    assert float(out[1][0][2]) > 0


def test_a_lazy_loss_is_issued_when_the_caller_does_something_else():
    """lazy=True followed by anything but backward() on its seeds: the loss is launched as its own call (same values as the eager one)."""
    from vslnet_amd.engine import Engine, flat_from_state_dict
    cfg = O.make_cfg(video_feature_dim=64, max_pos_len=64, word_size=52, predictor='transformer', drop_rate=0.0)
    P = O.random_params(cfg, seed=5)
    eng = Engine(cfg)
    d = {k: v.cuda().contiguous() for k, v in O.synthetic_batch(cfg, B=3, T=64, Lq=7, Lc=6, seed=6, ragged=True).items()}
    pad, glove = P['embedding_net.word_emb.pad_vec'].cuda(), P['embedding_net.word_emb.glove_vec'].cuda()
    flat = flat_from_state_dict(eng, P)
    fwd = lambda: eng.forward(flat, pad, glove, d['word_ids'], d['char_ids'], d['vfeats'], d['v_mask'], d['q_mask'], training=False, seed=0)
    fwd()
    ref = [t.clone() for t in eng.loss(d['s_labels'], d['e_labels'], d['h_labels'], 1.0, 5.0)]
    lz = eng.loss(d['s_labels'], d['e_labels'], d['h_labels'], 1.0, 5.0, lazy=True)
    fwd()                                                               # another forward: the pending loss goes out first
    torch.cuda.synchronize()
    for a, b in zip(ref, lz):
        assert torch.equal(a, b)
    lz = eng.loss(d['s_labels'], d['e_labels'], d['h_labels'], 1.0, 5.0, lazy=True)
    other = [t.clone() * 2 for t in ref[1:]]
    g2 = eng.backward(other[0], other[1], other[2], eng.new_flat()).clone()       # a backward on OTHER seeds: the pending loss first, then those seeds
    g1 = eng.backward(ref[1], ref[2], ref[3], eng.new_flat())
    torch.cuda.synchronize()
    assert torch.equal(lz[0], ref[0])
    assert torch.allclose(g2, 2 * g1, rtol=1e-5, atol=1e-9)
