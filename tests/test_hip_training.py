"""GPU tests of what the golden (eval-mode) fixtures cannot pin: training-mode dropout and the larger BASELINE shapes.

* dropout masks are counter-based hashes, so bit-parity with torch's CPU RNG is impossible: they are
  tested statistically, for determinism, and -- most importantly -- for forward/backward consistency through a
  finite-difference directional derivative of the training-mode loss at a FIXED seed.
* BASELINE configs 3-5 (T=256 Dv=4096 ; T=256 ; T=1024) are checked against the oracle at a batch the oracle finishes in
  seconds; the kernels' behaviour does not depend on B beyond the grid size.
"""
import numpy as np
import pytest
import torch

from oracle import vslnet_oracle as O
from tests.helpers import RELU_NOISE_KAPPA, assert_forced_relu_inside_noise

pytestmark = pytest.mark.gpu


def _engine(cfg, P):
    from vslnet_amd.engine import Engine, flat_from_state_dict
    eng = Engine(cfg)
    return eng, flat_from_state_dict(eng, P)


def _dev(b):
    return {k: v.cuda().contiguous() for k, v in b.items()}


def _fwd(eng, flat, P, d, training, seed):
    pad, glove = P.get('embedding_net.word_emb.pad_vec'), P.get('embedding_net.word_emb.glove_vec')     # absent: trainable word table
    return eng.forward(flat, None if pad is None else pad.cuda(), None if glove is None else glove.cuda(),
                       d['word_ids'], d['char_ids'], d['vfeats'], d['v_mask'], d['q_mask'], training=training, seed=seed)


def _loss(eng, flat, P, d, seed):
    _fwd(eng, flat, P, d, True, seed)
    losses, d_h, d_sl, d_el = eng.loss(d['s_labels'], d['e_labels'], d['h_labels'], 1.0, 5.0)
    return float(losses[2].item()), (d_h, d_sl, d_el)


@pytest.mark.parametrize('dv', [64, 500, 1024])
def test_visual_projection_weight_gradient_uses_the_forward_dropout_mask(dv):
    """Exact identity instead of finite differences (whose noise is ~3 % at these widths): with m the dropout mask of the
    features, vf = (X * m) W^T + b and dW = G^T (X * m), so for ANY matrix Z:  <dW, Z> = <G, (X * m) Z^T>, and the right-hand
    product is what the forward writes when the weight is Z and the bias 0 (same seed -> same mask).  Covers every k-chunk
    of the forward GEMM and every k-tile of the weight-gradient kernel, incl. a width that is not a multiple of 8 or 128."""
    cfg = O.make_cfg(video_feature_dim=dv, max_pos_len=64, word_size=52, drop_rate=0.2)
    P = O.random_params(cfg, seed=3)
    B, T = 3, 37
    d = _dev(O.synthetic_batch(cfg, B=B, T=T, Lq=4, Lc=5, seed=4, ragged=True))
    eng, flat = _engine(cfg, P)
    seed = 424242
    _, seeds = _loss(eng, flat, P, d, seed)
    g = eng.backward(*seeds, eng.new_flat())
    dW = eng.views(g)['video_affine.linear.conv1d.weight'].double().squeeze(-1).clone()      # (128, dv)
    G = eng.ws_view('d_video_affine', (B * T, 128)).double().clone()
    Z = torch.randn(128, dv, generator=torch.Generator().manual_seed(1)).cuda()
    flat_z = flat.clone()
    vz = eng.views(flat_z)
    vz['video_affine.linear.conv1d.weight'].copy_(Z.unsqueeze(-1))
    vz['video_affine.linear.conv1d.bias'].zero_()
    _fwd(eng, flat_z, P, d, True, seed)
    vf_z = eng.ws_view('video_affine', (B * T, 128)).double()
    lhs, rhs = float((dW * Z.double()).sum()), float((G * vf_z).sum())
    scale = float((G.abs() * vf_z.abs()).sum())
    assert abs(lhs - rhs) <= 1e-5 * scale, (lhs, rhs, scale)
    # and the mask really is there: without it the identity fails by far more than the tolerance
    _fwd(eng, flat_z, P, d, False, seed)
    rhs_nomask = float((G * eng.ws_view('video_affine', (B * T, 128)).double()).sum())
    assert abs(lhs - rhs_nomask) > 1e-3 * scale


@pytest.mark.parametrize('dv,B,T,Lq,Lc,predictor', [(64, 4, 48, 9, 7, 'transformer'), (64, 5, 33, 6, 5, 'rnn')])
def test_dropout_forward_backward_consistency_by_finite_differences(dv, B, T, Lq, Lc, predictor):
    cfg = O.make_cfg(video_feature_dim=dv, max_pos_len=64, word_size=52, drop_rate=0.2, predictor=predictor)
    P = O.random_params(cfg, seed=3)
    d = _dev(O.synthetic_batch(cfg, B=B, T=T, Lq=Lq, Lc=Lc, seed=4, ragged=True))
    eng, flat = _engine(cfg, P)
    seed = 1234567
    f0, seeds = _loss(eng, flat, P, d, seed)
    g = eng.backward(*seeds, eng.new_flat()).clone()
    f0b, _ = _loss(eng, flat, P, d, seed)
    assert f0 == f0b, 'training-mode forward is not deterministic for a fixed seed'
    f_other, _ = _loss(eng, flat, P, d, seed + 1)
    assert f_other != f0, 'the dropout seed has no effect'
    gd = g.double()
    gn = float(gd.norm())
    assert np.isfinite(gn) and gn > 0
    rs = np.random.RandomState(0)
    for trial in range(3):
        # direction: the gradient itself (trial 0), then the gradient restricted to a random half of the parameters (the
        # predicted slope is then the norm of that half: never small, so eps stays in the linear range)
        v = gd if trial == 0 else gd * torch.from_numpy(rs.choice([0.0, 1.0], size=gd.numel())).cuda()
        v = v / float(v.norm())
        pred = float((gd * v).sum())
        eps = 0.02 / max(abs(pred), 1e-3)                      # predicted change of the loss ~ 2e-2 (fp32 noise ~ 3e-5)
        fp, _ = _loss(eng, (flat.double() + eps * v).float(), P, d, seed)
        fm, _ = _loss(eng, (flat.double() - eps * v).float(), P, d, seed)
        fd = (fp - fm) / (2 * eps)
        assert abs(fd - pred) <= 0.03 * abs(pred) + 2e-3, 'trial %d: finite difference %.6f vs g.v %.6f' % (trial, fd, pred)


@pytest.mark.parametrize('dv,B,T,Lq,Lc,predictor,char_dim', [
    (64, 3, 40, 7, 6, 'transformer', 50), (1024, 2, 128, 20, 10, 'transformer', 50),
    (500, 4, 33, 5, 4, 'rnn', 50),
    (64, 2, 40, 82, 30, 'transformer', 50),     # ActivityNet's longest query
    (64, 2, 36, 128, 5, 'transformer', 50),     # the reference's bound (max_pos_len words): lean CQAttention layouts
    (1024, 16, 128, 20, 10, 'rnn', 50),         # BASELINE configs[0] as written
    (64, 3, 40, 7, 20, 'transformer', 100),     # main_t7.py:24's ActivityNet char_dim: two input-channel blocks x two position tiles forward, 7 channel tiles backward
    (64, 2, 30, 9, 40, 'transformer', 128),     # the engine's bounds: longest token, widest character embedding
    (64, 5, 24, 6, 8, 'transformer', 72),       # a width that is no multiple of 16
    (64, 2, 301, 7, 6, 'transformer', 50),      # T > 256: k_attn_fwd / k_attn_bwd_long, two dropout decisions per hash over an odd row length
    (64, 6, 24, 9, 6, 'transformer', -50)])     # (negative: WordEmbedding(word_vectors=None), the trainable table; 54 words over 100 ids: repeats)
def test_training_mode_matches_oracle_on_the_same_dropout_masks(dv, B, T, Lq, Lc, predictor, char_dim):
    """drop_rate 0.2 (the benchmark's mode), all 41 dropout sites: the oracle is handed the HIP path's masks -- recomputed on
    the host from the documented counter-based hash (tests/helpers.py) -- and must then agree with the training-mode forward
    (logits 1e-4) and with every gradient (1e-4 * |g|inf + 1e-6), like the eval-mode parity tests."""
    from tests.helpers import relu_flips, hip_dropout
    cfg = O.make_cfg(video_feature_dim=dv, max_pos_len=max(T, Lq), word_size=102, drop_rate=0.2, predictor=predictor, char_dim=abs(char_dim),
                     word_table=char_dim < 0)
    P = O.random_params(cfg, seed=21)
    b = O.synthetic_batch(cfg, B, T, Lq, Lc, seed=22, ragged=True)
    d = _dev(b)
    eng, flat = _engine(cfg, P)
    seed = (7 << 33) + 12345                     # exercises the high word of the 64-bit seed too
    h, sl, el = _fwd(eng, flat, P, d, True, seed)
    losses, d_h, d_sl, d_el = eng.loss(d['s_labels'], d['e_labels'], d['h_labels'], 1.0, 5.0)
    g = eng.backward(d_h, d_sl, d_el, eng.new_flat())
    torch.cuda.synchronize()
    O.record_relu_signs()
    O.force_dropout(hip_dropout(seed))
    with torch.no_grad():
        O.total_loss(P, cfg, b, training=True)
    _, hip_masks = relu_flips(eng, B, T, Lq, predictor=predictor)
    O.record_relu_signs(False)
    O.force_relu_signs(hip_masks)
    O.force_dropout(hip_dropout(seed))
    Pg = {k: v.clone().requires_grad_(k not in O.FROZEN) for k, v in P.items()}
    total, (oh, osl, oel, _, _) = O.total_loss(Pg, cfg, b, training=True)
    n_sites = O.DROP_CALLS
    O.force_relu_signs(None)
    O.force_dropout(None)
    assert n_sites == (41 if predictor == 'transformer' else 23)
    assert_forced_relu_inside_noise(O)
    total.backward()
    fin = osl.detach().abs() < 1e29
    scale = max(1.0, float(osl.detach()[fin].abs().max()))
    assert float((sl.cpu() - osl.detach())[fin].abs().max()) <= 1e-4 * scale
    assert float((el.cpu() - oel.detach())[fin].abs().max()) <= 1e-4 * scale
    assert float((h.cpu() - oh.detach()).abs().max()) <= 2e-5
    assert abs(float(losses[2]) - float(total.detach())) <= 1e-4 * max(1.0, abs(float(total.detach())))
    bad = []
    for k, t in eng.views(g).items():
        ref = Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(Pg[k])
        err, tol = float((t.cpu() - ref).abs().max()), 1e-4 * float(ref.abs().max()) + 1e-6
        if not err <= tol:
            bad.append((k, err, tol))
    assert not bad, bad[:6]


def test_headline_shape_at_full_size():
    """BASELINE configs[1] exactly as bench.py runs it (B=64, T=128, Dv=1024, Lq=20, Lc=10, drop_rate 0.2, train mode):
    (1) against the oracle on the same dropout masks -- logits 1e-4, every gradient 1e-4 * |g|inf + 1e-6;
    (2) size-independent property the data-parallel path rests on: with the GLOBAL normalisers (1/B, sum of the mask) and
        `sample_offset`, the training-mode gradients of three uneven shards add up to the gradient of the full batch."""
    from tests.helpers import relu_flips, hip_dropout
    B, T, Lq, Lc = 64, 128, 20, 10
    cfg = O.make_cfg(video_feature_dim=1024, max_pos_len=128, word_size=1002, drop_rate=0.2)
    P = O.random_params(cfg, seed=12345)
    b = O.synthetic_batch(cfg, B, T, Lq, Lc, seed=0, ragged=True)
    d = _dev(b)
    eng, flat = _engine(cfg, P)
    seed = 20260928
    h, sl, el = _fwd(eng, flat, P, d, True, seed)
    losses, d_h, d_sl, d_el = eng.loss(d['s_labels'], d['e_labels'], d['h_labels'], 1.0, 5.0)
    g = eng.backward(d_h, d_sl, d_el, eng.new_flat()).clone()
    torch.cuda.synchronize()
    O.record_relu_signs()
    O.force_dropout(hip_dropout(seed))
    with torch.no_grad():
        O.total_loss(P, cfg, b, training=True)
    flips, hip_masks = relu_flips(eng, B, T, Lq)
    O.record_relu_signs(False)
    O.force_relu_signs(hip_masks)
    O.force_dropout(hip_dropout(seed))
    Pg = {k: v.clone().requires_grad_(k not in O.FROZEN) for k, v in P.items()}
    total, (oh, osl, oel, _, _) = O.total_loss(Pg, cfg, b, training=True)
    O.force_relu_signs(None)
    O.force_dropout(None)
    assert_forced_relu_inside_noise(O, flips)
    total.backward()
    fin = osl.detach().abs() < 1e29
    scale = max(1.0, float(osl.detach()[fin].abs().max()))
    assert float((sl.cpu() - osl.detach())[fin].abs().max()) <= 1e-4 * scale
    assert float((el.cpu() - oel.detach())[fin].abs().max()) <= 1e-4 * scale
    assert abs(float(losses[2]) - float(total.detach())) <= 1e-4 * max(1.0, abs(float(total.detach())))
    bad = []
    for k, t in eng.views(g).items():
        ref = Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(Pg[k])
        err, tol = float((t.cpu() - ref).abs().max()), 1e-4 * float(ref.abs().max()) + 1e-6
        if not err <= tol:
            bad.append((k, err, tol))
    assert not bad, (flips, bad[:6])
    # ---- (2) shards add up, in TRAINING mode: `sample_offset` continues the dropout counters, so the masks of a shard are
    #      the rows of the full batch's masks and N ranks reproduce the single-process gradient
    from tests.helpers import hip_relu_masks
    moved = [0]                                  # ReLU decisions that differ between the full-batch run and a shard run

    def grads_of(lo, hi):
        dd = {k: v[lo:hi].contiguous() for k, v in d.items()}
        eng.forward(flat, P['embedding_net.word_emb.pad_vec'].cuda(), P['embedding_net.word_emb.glove_vec'].cuda(), dd['word_ids'],
                    dd['char_ids'], dd['vfeats'], dd['v_mask'], dd['q_mask'], training=True, seed=seed, sample_offset=lo)
        _, dh, dsl, del_ = eng.loss(dd['s_labels'], dd['e_labels'], dd['h_labels'], 1.0, 5.0, inv_batch=1.0 / B,
                                    mask_sum=float(b['v_mask'].sum()))
        out = eng.backward(dh, dsl, del_, eng.new_flat()).double().clone()
        moved[0] += sum(int((m != f[lo:hi]).sum()) for m, f in zip(hip_relu_masks(eng, hi - lo, T, Lq), hip_masks))
        return out
    parts = grads_of(0, 24) + grads_of(24, 40) + grads_of(40, B)
    full = g.double()
    # Every GEMM contracts k in a fixed order since round 3 (no per-workgroup rotation) and the shards start on tile boundaries (24 / 40 samples
    # x 128 clips, sample tiles on the query side), so a row's forward values -- and with them every ReLU decision -- are bit-identical
    # between the full batch and a shard: no decision may move, and the sums differ by summation order only.
    print('[shards] ReLU decisions that moved between the full batch and the shards: %d' % moved[0])
    assert moved[0] == 0, moved
    rel = 2e-5
    for k, t in eng.views(full).items():
        err = float((t - eng.views(parts)[k]).abs().max())
        assert err <= rel * float(t.abs().max()) + 1e-6, (k, err, moved)


@pytest.mark.parametrize('B,T,Dv,name', [(32, 256, 4096, 'configs[2] TACoS C3D'), (32, 256, 1024, 'configs[3] ActivityNet per-GPU shard'),
                                         (16, 1024, 1024, 'configs[4] long-video per-GPU shard')])
def test_other_baseline_shapes_at_full_size(B, T, Dv, name):
    """BASELINE configs[2..4] at their PER-GPU batch (the oracle comparison of these shapes runs at B = 1..3:
    test_baseline_shapes_against_oracle; a CPU oracle step at B = 32, Dv = 4096 takes minutes).  Size-independent properties, train mode:
    (1) two identical steps are bit-identical (logits, losses, every gradient: the race detector for the barrier-lean kernels);
    (2) masked logits are exactly -1e30, everything else finite;
    (3) the gradients of three uneven shards (global normalisers, sample_offset) add up to the full-batch gradient -- the data-parallel
        property, through the T > 128 attention kernels (T = 1024: the single-pass backward with partial dQ slabs)."""
    from tests.helpers import hip_relu_masks
    Lq, Lc = 20, 10
    cfg = O.make_cfg(video_feature_dim=Dv, max_pos_len=T, word_size=1002, drop_rate=0.2)
    P = O.random_params(cfg, seed=12345)
    b = O.synthetic_batch(cfg, B, T, Lq, Lc, seed=3, ragged=True)
    d = _dev(b)
    eng, flat = _engine(cfg, P)
    seed = 777
    pad, glove = P['embedding_net.word_emb.pad_vec'].cuda(), P['embedding_net.word_emb.glove_vec'].cuda()
    msum = float(b['v_mask'].sum())

    def run(lo, hi):
        dd = {k: v[lo:hi].contiguous() for k, v in d.items()}
        h, sl, el = eng.forward(flat, pad, glove, dd['word_ids'], dd['char_ids'], dd['vfeats'], dd['v_mask'], dd['q_mask'],
                                training=True, seed=seed, sample_offset=lo)
        losses, dh, dsl, del_ = eng.loss(dd['s_labels'], dd['e_labels'], dd['h_labels'], 1.0, 5.0, inv_batch=1.0 / B, mask_sum=msum)
        g = eng.backward(dh, dsl, del_, eng.new_flat()).clone()
        torch.cuda.synchronize()
        return h.clone(), sl.clone(), el.clone(), losses.clone(), g, hip_relu_masks(eng, hi - lo, T, Lq)
    h1, sl1, el1, l1, g1, m1 = run(0, B)
    h2, sl2, el2, l2, g2, _ = run(0, B)
    assert torch.equal(sl1, sl2) and torch.equal(el1, el2) and torch.equal(h1, h2) and torch.equal(l1, l2), name
    assert torch.equal(g1, g2), (name, float((g1 - g2).abs().max()))
    pad_pos = d['v_mask'] == 0
    assert bool((sl1[pad_pos] == -1e30).all()) and bool((el1[pad_pos] == -1e30).all()) and bool((h1[pad_pos] == 0).all())
    assert bool(torch.isfinite(sl1[~pad_pos]).all()) and bool(torch.isfinite(g1).all()) and bool(torch.isfinite(l1).all())
    cuts = [0, B // 3 + 1, B // 2 + 3, B]
    parts, moved = torch.zeros_like(g1, dtype=torch.float64), 0
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        _, _, _, _, g, m = run(lo, hi)
        parts += g.double()
        moved += sum(int((a != f[lo:hi]).sum()) for a, f in zip(m, m1))
    assert moved <= 8, (name, moved)              # ReLU pre-activations inside the rounding noise may land on the other side in a shard run
    rel = 2e-5 if moved == 0 else 1e-3
    full = g1.double()
    for k, t in eng.views(full).items():
        err = float((t - eng.views(parts)[k]).abs().max())
        assert err <= rel * float(t.abs().max()) + 1e-6, (name, k, err, moved)


@pytest.mark.parametrize('B,T,Lq,Lc,predictor', [(64, 128, 20, 10, 'transformer'), (5, 83, 9, 7, 'transformer'), (16, 128, 20, 10, 'rnn')])
def test_a_training_step_is_bitwise_reproducible(B, T, Lq, Lc, predictor):
    """Race detector for the fused kernels (barriers removed between phases that touch thread-private rows, three streams, stop-event
    ordering): the same step six times -> logits, losses and all 104 gradients bit-identical, at the headline shape (one workgroup
    per CU on every fused kernel), a ragged small batch and configs[0]'s rnn head."""
    cfg = O.make_cfg(video_feature_dim=1024, max_pos_len=128, word_size=302, drop_rate=0.2, predictor=predictor)
    P = O.random_params(cfg, seed=5)
    d = _dev(O.synthetic_batch(cfg, B=B, T=T, Lq=Lq, Lc=Lc, seed=11, ragged=True))
    eng, flat = _engine(cfg, P)
    ref = None
    for _ in range(6):
        h, sl, el = _fwd(eng, flat, P, d, True, 777)
        losses, d_h, d_sl, d_el = eng.loss(d['s_labels'], d['e_labels'], d['h_labels'], 1.0, 5.0)
        g = eng.backward(d_h, d_sl, d_el, eng.new_flat())
        torch.cuda.synchronize()
        cur = [t.clone() for t in (h, sl, el, losses, g)]
        if ref is None:
            ref = cur
        else:
            for a, b in zip(ref, cur):
                assert torch.equal(a, b)


@pytest.mark.parametrize('B,T', [(64, 128), (5, 83), (3, 700)])
def test_one_launch_loss_equals_the_two_kernel_path(B, T):
    """vsl_loss with the global mask sum supplied (the data-parallel path, bench.py) runs ONE kernel: per-sample gradient seeds plus a
    last-arriver reduction of the loss values.  Against the default path (mask sum computed on the device, two kernels): same losses and
    seeds to rounding, six calls in a row bit-identical (the arrival counter resets itself)."""
    cfg = O.make_cfg(video_feature_dim=64, max_pos_len=1024, word_size=52, drop_rate=0.0)
    P = O.random_params(cfg, seed=2)
    d = _dev(O.synthetic_batch(cfg, B=B, T=T, Lq=6, Lc=5, seed=9, ragged=True))
    eng, flat = _engine(cfg, P)
    _fwd(eng, flat, P, d, False, 0)
    ref = [t.clone() for t in eng.loss(d['s_labels'], d['e_labels'], d['h_labels'], 1.0, 5.0)]
    msum = float(d['v_mask'].sum().item())
    first = None
    for _ in range(6):
        cur = [t.clone() for t in eng.loss(d['s_labels'], d['e_labels'], d['h_labels'], 1.0, 5.0, inv_batch=1.0 / B, mask_sum=msum)]
        torch.cuda.synchronize()
        if first is None:
            first = cur
        for a, b in zip(first, cur):
            assert torch.equal(a, b)
    assert abs(float(first[0][3]) - msum) <= 1e-3
    for a, b in zip(ref, first):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max()))


def test_dropout_mask_statistics_and_scaling():
    """Word-embedding dropout (layers_t7.py:45) is directly observable in the saved concat buffer: kept entries equal
    table / (1 - p), the rest are exactly 0, and the drop fraction is p within sampling error."""
    p = 0.3
    cfg = O.make_cfg(video_feature_dim=64, max_pos_len=32, word_size=500, drop_rate=p)
    P = O.random_params(cfg, seed=5)
    b = O.synthetic_batch(cfg, B=16, T=16, Lq=20, Lc=6, seed=6)
    d = _dev(b)
    eng, flat = _engine(cfg, P)
    _fwd(eng, flat, P, d, True, 99)
    E = eng.ws_view('emb_concat', (16, 20, cfg.word_dim + 100))[:, :, :cfg.word_dim].cpu()
    table = torch.cat([P['embedding_net.word_emb.pad_vec'], P['embedding_net.word_emb.unk_vec'], P['embedding_net.word_emb.glove_vec']])
    ref = table[b['word_ids']]
    dropped = E == 0
    n = E.numel()
    frac = float(dropped.float().mean())
    assert abs(frac - p) < 5 * np.sqrt(p * (1 - p) / n), frac
    assert torch.allclose(E[~dropped], ref[~dropped] / (1 - p), rtol=1e-6, atol=1e-7)
    # a different seed gives a different, equally dense mask; eval mode gives no dropout at all
    _fwd(eng, flat, P, d, True, 100)
    E2 = eng.ws_view('emb_concat', (16, 20, cfg.word_dim + 100))[:, :, :cfg.word_dim].cpu()
    agree = float(((E2 == 0) == dropped).float().mean())
    assert abs(agree - (p * p + (1 - p) * (1 - p))) < 0.01        # independent masks agree with prob p^2 + (1-p)^2
    _fwd(eng, flat, P, d, False, 100)
    E3 = eng.ws_view('emb_concat', (16, 20, cfg.word_dim + 100))[:, :, :cfg.word_dim].cpu()
    assert torch.equal(E3, ref)


@pytest.mark.parametrize('shape', [
    dict(name='cfg3 TACoS C3D', T=256, Dv=4096, B=2, Lq=20, Lc=10),
    dict(name='cfg4 ActivityNet', T=256, Dv=1024, B=3, Lq=33, Lc=12),
    dict(name='cfg5 long video', T=1024, Dv=1024, B=1, Lq=20, Lc=10),
    dict(name='edge: B=1 minimal chars, odd lengths', T=37, Dv=64, B=1, Lq=5, Lc=4),
    dict(name='edge: TACoS-size query', T=40, Dv=64, B=2, Lq=64, Lc=24),
    dict(name='ActivityNet extremes: longest query (82 words), 30-char word', T=256, Dv=1024, B=2, Lq=82, Lc=30),
    dict(name='edge: Lq=96 (last length on the 4-partial-tile CQAttention layout), Lc=40', T=48, Dv=64, B=2, Lq=96, Lc=40),
    dict(name='edge: engine limits Lq=128 = the reference max_pos_len bound, Lc=40', T=72, Dv=64, B=2, Lq=128, Lc=40),
    dict(name='edge: Lq=97 (first length on the lean layout), full word tiles absent', T=33, Dv=64, B=3, Lq=97, Lc=5),
    dict(name='edge: ActivityNet C3D width (500 = 4 * 125), one-word queries, ragged row count', T=50, Dv=500, B=3, Lq=1, Lc=4),
    dict(name='embedding: char_dim 100, a 130-character alphabet (9 table tiles), trainable word table', T=40, Dv=64, B=3, Lq=11, Lc=17,
         char_dim=100, char_size=130, word_table=True),
    dict(name='embedding: char_dim 64 with 25-character tokens (two input-channel blocks in the long-token instantiation; found by tools/fuzz_parity.py)',
         T=31, Dv=4, B=6, Lq=64, Lc=25, char_dim=64),
    dict(name='embedding: word_dim 52 (152 columns, not a multiple of 16: the fp32-input k_linear_fwd serves the embedding linear)', T=40, Dv=64, B=3, Lq=9, Lc=7,
         word_dim=52),
])
def test_baseline_shapes_against_oracle(shape):
    check_shape_against_oracle(shape)


def check_shape_against_oracle(shape, scaled_bias_floor=False):
    """Forward logits, losses, every gradient and extract_index of one shape against the oracle.  scaled_bias_floor (tests/test_fuzz_parity.py):
    the absolute floor of a Conv1D bias gradient's gate is max(1e-6, KAPPA * 2^-24 * sum |dY|) -- what was actually added -- so that tiny shapes
    with cancelling bias sums are judged by their summands (two T = 4 shapes of the round-4 sweep tripped the fixed 1e-6 on a structural zero)."""
    cfg = O.make_cfg(video_feature_dim=shape['Dv'], max_pos_len=max(shape['T'], shape['Lq']), word_size=102, char_dim=shape.get('char_dim', 50),
                     char_size=shape.get('char_size', 40), word_table=shape.get('word_table', False), word_dim=shape.get('word_dim', 300))
    P = O.random_params(cfg, seed=11)
    b = O.synthetic_batch(cfg, shape['B'], shape['T'], shape['Lq'], shape['Lc'], seed=12, ragged=shape['B'] > 1)
    d = _dev(b)
    eng, flat = _engine(cfg, P)
    h, sl, el = _fwd(eng, flat, P, d, False, 0)
    losses, d_h, d_sl, d_el = eng.loss(d['s_labels'], d['e_labels'], d['h_labels'], 1.0, 5.0)
    g = eng.backward(d_h, d_sl, d_el, eng.new_flat())
    torch.cuda.synchronize()
    from tests.helpers import relu_flips
    O.record_relu_signs()
    with torch.no_grad():
        O.total_loss(P, cfg, b)
    flips, hip_masks = relu_flips(eng, shape['B'], shape['T'], shape['Lq'])
    O.record_relu_signs(False)
    # gradients are discontinuous where a ReLU pre-activation crosses zero; when the saved masks show that the two forwards
    # disagree on a branch (likely among the 131 k pre-activations per layer of the long-video shape) the oracle is evaluated
    # on the branch the GPU path took, so the strict gate 1e-4 * ||g||inf + 1e-6 holds in every case
    # (always: torch's own no_grad and autograd forwards can land on different sides of a |z| ~ 1e-8 pre-activation, so the
    # count above is not enough; the branch taken is legitimate iff the pre-activation it overrides is inside the noise)
    O.force_relu_signs(hip_masks)
    Pg = {k: v.clone().requires_grad_(k not in O.FROZEN) for k, v in P.items()}
    O.record_bias_terms(scaled_bias_floor)
    total, (oh, osl, oel, _, _) = O.total_loss(Pg, cfg, b)
    O.force_relu_signs(None)
    assert_forced_relu_inside_noise(O, (shape['name'], flips))
    total.backward()
    bias_S = O.bias_term_sums(Pg) if scaled_bias_floor else {}
    O.record_bias_terms(False)
    fin = osl.detach().abs() < 1e29
    scale = max(1.0, float(osl.detach()[fin].abs().max()))
    assert float((sl.cpu() - osl.detach())[fin].abs().max()) <= 1e-4 * scale, shape['name']
    assert float((el.cpu() - oel.detach())[fin].abs().max()) <= 1e-4 * scale
    assert float((h.cpu() - oh.detach()).abs().max()) <= 2e-5
    assert abs(float(losses[2]) - float(total.detach())) <= 1e-4 * max(1.0, abs(float(total.detach())))
    bad = []
    for k, t in eng.views(g).items():
        ref = Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(Pg[k])
        err = float((t.cpu() - ref).abs().max())
        tol = 1e-4 * float(ref.abs().max()) + max(1e-6, RELU_NOISE_KAPPA * 2.0 ** -24 * bias_S.get(k, 0.0))
        if k == 'cq_attention.w4Q' and shape['Lq'] == 1:
            # one query word: the softmax over j is the constant 1, so dw4Q is structurally zero (w4Q only acts
            # through that softmax).  What both sides compute is the cancellation residue of sum_i S_col (dS - dot); the gate
            # for a structural zero is absolute, at the size of that residue (|dot| ~ 1, softmax normalised to ~1e-6)
            tol = 2e-5
        if not err <= tol:
            bad.append((k, err, tol))
    assert not bad, (shape['name'], flips, bad[:5])
    si, ei = eng.extract_index(sl, el)
    osi, oei = O.extract_index(osl.detach(), oel.detach())
    assert torch.equal(si.cpu(), osi) and torch.equal(ei.cpu(), oei)


def test_limits_fail_loudly():
    from vslnet_amd.engine import VslError
    cfg = O.make_cfg(video_feature_dim=64, max_pos_len=32, word_size=52)
    P = O.random_params(cfg, seed=1)
    eng, flat = _engine(cfg, P)
    d = _dev(O.synthetic_batch(cfg, B=1, T=40, Lq=5, Lc=5, seed=1))          # T > max_pos_len
    with pytest.raises(IndexError):
        _fwd(eng, flat, P, d, False, 0)
    d = _dev(O.synthetic_batch(cfg, B=1, T=16, Lq=5, Lc=3, seed=1))          # Lc < 4: the widest char conv does not fit
    with pytest.raises(VslError, match='Lc'):
        _fwd(eng, flat, P, d, False, 0)
