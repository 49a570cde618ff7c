"""Optimizer step on the flat buckets: clip_grad_norm_(1.0) + AdamW(eps 1e-6, wd 0.01 except bias / layer_norm)
+ linear decay, main_t7.py:111-113 / VSLNet_t7.py:8-17.  The reference's transformers.AdamW is gone from transformers 5.x
("optimizer parity unpinned"), so the pin is torch.optim.AdamW + torch.nn.utils.clip_grad_norm_, which is what
the reference's own import shim resolves to today."""
import numpy as np
import pytest
import torch

from oracle import vslnet_oracle as O
from vslnet_amd import dp


def _layout(P):
    layout, off = [], 0
    for k, v in P.items():
        if k in O.FROZEN:
            continue
        layout.append((k, off, v.numel(), tuple(v.shape)))
        off += (v.numel() + 3) & ~3
    return layout, off


def _reference_steps(P, layout, grads_seq, lr0, N):
    params = {n: torch.nn.Parameter(P[n].clone()) for n, _, _, _ in layout}
    no_decay = ('bias', 'layer_norm', 'LayerNorm')                                   # VSLNet_t7.py:9-13
    groups = [{'params': [p for n, p in params.items() if not any(k in n for k in no_decay)], 'weight_decay': 0.01},
              {'params': [p for n, p in params.items() if any(k in n for k in no_decay)], 'weight_decay': 0.0}]
    opt = torch.optim.AdamW(groups, lr=lr0, eps=1e-6)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda n: max(0.0, (N - n) / N))   # linear decay, no warm-up
    norms = []
    for g in grads_seq:
        for n, o, k, shp in layout:
            params[n].grad = g[o:o + k].view(shp).clone()
        norms.append(float(torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0)))
        opt.step()
        sched.step()
    return params, norms


def _case(seed=0, scale=(0.01, 3.0, 0.2)):
    cfg = O.make_cfg(video_feature_dim=64, max_pos_len=32, word_size=52)
    P = O.random_params(cfg, seed=seed)
    layout, n = _layout(P)
    g = torch.Generator().manual_seed(seed + 1)
    grads_seq = []
    for s in scale:                                                                   # below and above the clip threshold
        f = torch.zeros(n)
        for _, o, k, _ in layout:
            f[o:o + k] = torch.randn(k, generator=g) * s / np.sqrt(n)
        grads_seq.append(f)
    flat = torch.zeros(n)
    for nm, o, k, _ in layout:
        flat[o:o + k] = P[nm].reshape(-1)
    return cfg, P, layout, flat, grads_seq


def test_flat_adamw_torch_path_matches_torch_optim():
    cfg, P, layout, flat, grads_seq = _case()
    ref, norms = _reference_steps(P, layout, grads_seq, 1e-3, 100)
    assert norms[0] < 1.0 < norms[1]                                                  # both clip branches are exercised
    opt = dp.FlatAdamW(flat, layout, lr=1e-3, num_train_steps=100)
    for g in grads_seq:
        opt.step(g)
    for n, o, k, shp in layout:
        assert torch.allclose(flat[o:o + k].view(shp), ref[n].data, rtol=1e-5, atol=1e-7), n


@pytest.mark.gpu
def test_fused_adamw_kernels_match_torch_optim():
    from vslnet_amd.engine import Engine, flat_from_state_dict
    cfg, P, _, _, _ = _case()
    eng = Engine(cfg)
    layout = eng.layout
    n = eng.param_floats
    g = torch.Generator().manual_seed(7)
    grads_seq = []
    for s in (0.01, 3.0, 0.2):
        f = torch.zeros(n)
        for _, o, k, _ in layout:
            f[o:o + k] = torch.randn(k, generator=g) * s / np.sqrt(n)
        grads_seq.append(f)
    ref, norms = _reference_steps(P, layout, grads_seq, 1e-3, 100)
    flat = flat_from_state_dict(eng, P)
    opt = dp.FlatAdamW(flat, layout, lr=1e-3, num_train_steps=100, engine=eng)
    gn = torch.zeros(1, device='cuda')
    for i, gr in enumerate(grads_seq):
        gd = gr.cuda()
        keep = gd.clone()
        opt.step(gd)
        assert torch.equal(gd, keep), 'the gradient bucket must not be modified'
        eng.adamw_step(flat.clone(), gd, opt.m.clone(), opt.v.clone(), 0.0, i + 1, grad_norm_out=gn)   # lr 0: only reports the norm
        assert abs(float(gn) - norms[i]) <= 1e-5 * norms[i]
    out = eng.views(flat)
    for nm, o, k, shp in layout:
        assert torch.allclose(out[nm].cpu(), ref[nm].data, rtol=2e-5, atol=1e-7), nm
    with pytest.raises(Exception):
        eng.adamw_step(flat, grads_seq[0].cuda(), opt.m, opt.v, 1e-3, 0)               # step must be >= 1


@pytest.mark.gpu
@pytest.mark.parametrize('predictor', ['transformer', 'rnn'])
def test_norm_from_backward_matches_the_two_kernel_step(predictor):
    """vsl_adamw.norm_from_backward (round 4): the clip's global norm out of the sums of squares the backward's final reduction records per
    block, instead of a k_sqsum pass over the bucket.  Same norm to fp32 rounding, same update; NaNs in the bucket's alignment pads (which
    no kernel writes) do not reach it; another bucket than the last backward's is refused."""
    from oracle import vslnet_oracle as O
    from vslnet_amd.engine import Engine, VslError, flat_from_state_dict
    cfg = O.make_cfg(video_feature_dim=64, max_pos_len=64, word_size=52, predictor=predictor, drop_rate=0.2)
    P = O.random_params(cfg, seed=2)
    eng = Engine(cfg)
    d = {k: v.cuda().contiguous() for k, v in O.synthetic_batch(cfg, B=5, T=40, Lq=7, Lc=6, seed=4, ragged=True).items()}
    pad, glove = P['embedding_net.word_emb.pad_vec'].cuda(), P['embedding_net.word_emb.glove_vec'].cuda()
    flat = flat_from_state_dict(eng, P)
    eng.forward(flat, pad, glove, d['word_ids'], d['char_ids'], d['vfeats'], d['v_mask'], d['q_mask'], training=True, seed=3)
    _, d_h, d_sl, d_el = eng.loss(d['s_labels'], d['e_labels'], d['h_labels'], 1.0, 5.0)
    grads = torch.full((eng.param_floats,), float('nan'), device='cuda')       # pads stay NaN: the k_sqsum form could not take this bucket
    eng.backward(d_h, d_sl, d_el, grads)
    real = torch.zeros(eng.param_floats, dtype=torch.bool, device='cuda')
    for _, off, numel, _ in eng.layout:
        real[off:off + numel] = True
    want = float(torch.linalg.vector_norm(grads[real].double()))
    clean = torch.where(real, grads, torch.zeros_like(grads))
    out = []
    for fused in (False, True):
        f, m, v, gn = flat.clone(), torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros(1, device='cuda')
        eng.adamw_step(f, grads if fused else clean, m, v, 1e-3, 1, clip_norm=0.05, grad_norm_out=gn, norm_from_backward=fused) if fused else \
            eng.adamw_step(f, clean, m, v, 1e-3, 1, clip_norm=0.05, grad_norm_out=gn)
        torch.cuda.synchronize()
        out.append((f[real], m[real], v[real], float(gn)))
    assert abs(out[0][3] - want) <= 2e-6 * want and abs(out[1][3] - want) <= 2e-6 * want, (out[0][3], out[1][3], want)
    assert want > 0.05                                                          # the clip is active: the norm matters
    for a, b, nm in zip(out[0][:3], out[1][:3], ('params', 'exp_avg', 'exp_avg_sq')):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-9), nm
    with pytest.raises(VslError, match='norm_from_backward'):
        eng.adamw_step(flat.clone(), clean, torch.zeros_like(flat), torch.zeros_like(flat), 1e-3, 1, norm_from_backward=True)


@pytest.mark.gpu
@pytest.mark.parametrize('variant', ['transformer', 'rnn', 'word_table', 'hf_order', 'two launches'])
def test_fused_step_matches_backward_then_adamw(variant, monkeypatch):
    """vsl_io.fused_step (round 6): the optimizer step applied by the backward's LAST launch (final reduction + global norm + AdamW in one
    kernel whose workgroups hand their sums of squares to each other) against backward() followed by adamw_step(norm_from_backward=True).
    Three consecutive training steps from the same state: identical gradients, the same norm to fp32 rounding (another summation order),
    the same parameters and moments.  word_table = 1 (its table gradient does not leave the reduction) takes the library's two-launch form, and
    so does every configuration unless VSL_FUSED_TAIL=1 asks for the one launch (the library's default: profiles/r06_notes.md section 7)."""
    monkeypatch.setenv('VSL_FUSED_TAIL', '0' if variant == 'two launches' else '1')
    from oracle import vslnet_oracle as O
    from vslnet_amd.engine import Engine, flat_from_state_dict
    from vslnet_amd.dp import FlatAdamW
    kw = dict(video_feature_dim=64, max_pos_len=64, word_size=52, predictor='rnn' if variant == 'rnn' else 'transformer', drop_rate=0.2)
    if variant == 'word_table':
        kw['word_table'] = True
    cfg = O.make_cfg(**kw)
    P = O.random_params(cfg, seed=2)
    eng = Engine(cfg)
    batches = [{k: v.cuda().contiguous() for k, v in O.synthetic_batch(cfg, B=5, T=40, Lq=7, Lc=6, seed=4 + i, ragged=True).items()} for i in range(3)]
    pad, glove = P.get('embedding_net.word_emb.pad_vec'), P.get('embedding_net.word_emb.glove_vec')       # (absent: trainable word table)
    pad, glove = (None if pad is None else pad.cuda()), (None if glove is None else glove.cuda())
    real = torch.zeros(eng.param_floats, dtype=torch.bool, device='cuda')
    for _, off, numel, _ in eng.layout:
        real[off:off + numel] = True
    runs = []
    for fused in (False, True):
        flat = flat_from_state_dict(eng, P).clone()
        opt = FlatAdamW(flat, eng.layout, lr=1e-3, num_train_steps=10, clip_norm=0.05, engine=eng, hf_order=variant == 'hf_order')
        norms, gsnap = [], []
        for i, d in enumerate(batches):
            eng.forward(flat, pad, glove, d['word_ids'], d['char_ids'], d['vfeats'], d['v_mask'], d['q_mask'], training=True, seed=3 + i)
            _, d_h, d_sl, d_el = eng.loss(d['s_labels'], d['e_labels'], d['h_labels'], 1.0, 5.0)
            grads, gn = torch.zeros(eng.param_floats, device='cuda'), torch.zeros(1, device='cuda')
            if fused:
                eng.backward(d_h, d_sl, d_el, grads, fused_step=opt.fused_step(grad_norm_out=gn))
            else:
                eng.backward(d_h, d_sl, d_el, grads)
                lr = opt.lr()
                opt.t += 1
                eng.adamw_step(flat, grads, opt.m, opt.v, lr, opt.t, clip_norm=0.05, grad_norm_out=gn, hf_order=opt.hf_order, norm_from_backward=True)
            torch.cuda.synchronize()
            norms.append(float(gn))
            gsnap.append(grads[real].clone())
        runs.append((flat[real].clone(), opt.m[real].clone(), opt.v[real].clone(), norms, gsnap, opt.t))
    a, b = runs
    assert a[5] == b[5] == 3
    assert torch.equal(a[4][0], b[4][0])                                          # same weights, same batch, same seed: the same gradients
    for na, nb in zip(a[3], b[3]):
        assert na > 0.05 and abs(na - nb) <= 4e-6 * na, (a[3], b[3])               # (the clip is active)
    for x, y, nm in zip(a[:3], b[:3], ('params', 'exp_avg', 'exp_avg_sq')):
        assert torch.isfinite(y).all(), nm
        assert torch.allclose(x, y, rtol=2e-5, atol=1e-8), (nm, float((x - y).abs().max()))


@pytest.mark.gpu
def test_fused_step_rejects_what_it_cannot_do():
    from oracle import vslnet_oracle as O
    from vslnet_amd.engine import Engine, VslError, flat_from_state_dict
    cfg = O.make_cfg(video_feature_dim=64, max_pos_len=64, word_size=52, predictor='transformer', drop_rate=0.0)
    P = O.random_params(cfg, seed=2)
    eng = Engine(cfg)
    d = {k: v.cuda().contiguous() for k, v in O.synthetic_batch(cfg, B=2, T=32, Lq=5, Lc=4, seed=4).items()}
    flat = flat_from_state_dict(eng, P)
    eng.forward(flat, P['embedding_net.word_emb.pad_vec'].cuda(), P['embedding_net.word_emb.glove_vec'].cuda(), d['word_ids'], d['char_ids'],
                d['vfeats'], d['v_mask'], d['q_mask'], training=True, seed=1)
    _, d_h, d_sl, d_el = eng.loss(d['s_labels'], d['e_labels'], d['h_labels'], 1.0, 5.0)
    z = torch.zeros_like(flat)
    fs = dict(flat=flat.clone(), exp_avg=z.clone(), exp_avg_sq=z.clone(), lr=1e-3, step=0)
    with pytest.raises(VslError, match='step'):
        eng.backward(d_h, d_sl, d_el, eng.new_flat(), fused_step=fs)
    with pytest.raises(ValueError, match='early_event'):
        eng.backward(d_h, d_sl, d_el, eng.new_flat(), fused_step=dict(fs, step=1), early_event=torch.cuda.Event())


def _hf_adamw_steps(P, layout, grads_seq, lr0, total, eps=1e-6, wd=0.01, clip=1.0):
    """The historical transformers.AdamW (the class VSLNet_t7.py:5 imports; removed from transformers 5.x), restated from its
    published source, per parameter, in float64: exp_avg / exp_avg_sq update, step_size = lr * sqrt(1 - b2^t) / (1 - b1^t),
    p -= step_size * m / (sqrt(v) + eps), THEN p -= lr * wd * p; decay skipped for names with bias / layer_norm / LayerNorm
    (VSLNet_t7.py:9-13); clip_grad_norm_(1.0) first (main_t7.py:111); linear decay of lr (VSLNet_t7.py:15-16)."""
    p = {n: P[n].double().clone() for n, _, _, _ in layout}
    m = {n: torch.zeros_like(v) for n, v in p.items()}
    v2 = {n: torch.zeros_like(v) for n, v in p.items()}
    for t, g in enumerate(grads_seq, 1):
        lr = lr0 * (total - (t - 1)) / total
        g = g.double()
        coef = min(1.0, clip / (float(g.norm()) + 1e-6))
        for n, o, k, shp in layout:
            ge = g[o:o + k].view(shp) * coef
            m[n].mul_(0.9).add_(ge, alpha=0.1)
            v2[n].mul_(0.999).addcmul_(ge, ge, value=0.001)
            step = lr * (1 - 0.999 ** t) ** 0.5 / (1 - 0.9 ** t)
            p[n].addcdiv_(m[n], v2[n].sqrt() + eps, value=-step)
            if not any(s in n for s in ('bias', 'layer_norm', 'LayerNorm')):
                p[n].add_(p[n], alpha=-lr * wd)
    return p


def test_flat_adamw_hf_order_matches_transformers_adamw_restatement():
    cfg, P, layout, flat, grads_seq = _case()
    ref = _hf_adamw_steps(P, layout, grads_seq, 1e-3, 100)
    opt = dp.FlatAdamW(flat, layout, lr=1e-3, num_train_steps=100, hf_order=True)
    for g in grads_seq:
        opt.step(g)
    for n, o, k, shp in layout:
        assert torch.allclose(flat[o:o + k].view(shp).double(), ref[n], rtol=1e-5, atol=1e-7), n


@pytest.mark.gpu
def test_fused_adamw_kernels_hf_order():
    from vslnet_amd.engine import Engine, flat_from_state_dict
    cfg, P, _, _, _ = _case()
    eng = Engine(cfg)
    layout, n = eng.layout, eng.param_floats
    g = torch.Generator().manual_seed(7)
    grads_seq = []
    for s in (0.01, 3.0, 0.2):
        f = torch.zeros(n)
        for _, o, k, _ in layout:
            f[o:o + k] = torch.randn(k, generator=g) * s / np.sqrt(n)
        grads_seq.append(f)
    ref = _hf_adamw_steps(P, layout, grads_seq, 1e-3, 100)
    flat = flat_from_state_dict(eng, P)
    opt = dp.FlatAdamW(flat, layout, lr=1e-3, num_train_steps=100, engine=eng, hf_order=True)
    for gr in grads_seq:
        opt.step(gr.cuda())
    out = eng.views(flat)
    for nm, o, k, shp in layout:
        assert torch.allclose(out[nm].cpu().double(), ref[nm], rtol=2e-5, atol=1e-7), nm
