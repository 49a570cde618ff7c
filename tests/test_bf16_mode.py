"""bf16 THROUGHPUT mode (BASELINE configs[1] says "bf16"; the reference is fp32 end to end): bfloat16 features in HBM and a
bf16-MFMA VisualProjection (bf16 features x bf16-rounded weight, fp32 accumulation), everything downstream fp32.
Tolerances of this mode, stated here and nowhere claimed for the parity path:
  * against the fp32 oracle evaluated on the SAME rounded inputs (features and video_affine weight rounded to bf16): the products
    of two bf16 numbers are exact in fp32, so only the summation order differs -> the parity gates apply (logits 1e-4, gradients
    1e-4 * |g|inf + 1e-6);
  * against the fp32 oracle on the unrounded inputs: logits within 8e-2 (rounding the features alone moves logits by
    2.4e-2 at the real shape)."""
import numpy as np
import pytest
import torch

from oracle import vslnet_oracle as O
from tests.helpers import assert_forced_relu_inside_noise
from tests.helpers import relu_flips

pytestmark = pytest.mark.gpu


def _round_bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize('B,T,Dv,training', [(3, 40, 64, False), (2, 128, 1024, False), (3, 40, 1024, True)])
def test_bf16_mode_matches_the_oracle_on_rounded_inputs(B, T, Dv, training):
    from vslnet_amd.engine import Engine, flat_from_state_dict
    from tests.helpers import hip_dropout
    cfg = O.make_cfg(video_feature_dim=Dv, max_pos_len=T, word_size=52, drop_rate=0.2 if training else 0.0)
    P = O.random_params(cfg, seed=31)
    b = O.synthetic_batch(cfg, B, T, 9, 6, seed=32, ragged=True)
    eng = Engine(cfg)
    flat = flat_from_state_dict(eng, P)
    dev = lambda t: t.cuda().contiguous()                    # noqa: E731
    seed = 991
    h, sl, el = eng.forward(flat, dev(P['embedding_net.word_emb.pad_vec']), dev(P['embedding_net.word_emb.glove_vec']),
                            dev(b['word_ids']), dev(b['char_ids']), dev(b['vfeats'].to(torch.bfloat16)), dev(b['v_mask']), dev(b['q_mask']),
                            training=training, seed=seed)
    losses, d_h, d_sl, d_el = eng.loss(dev(b['s_labels']), dev(b['e_labels']), dev(b['h_labels']), 1.0, 5.0)
    g = eng.backward(d_h, d_sl, d_el, eng.new_flat())
    torch.cuda.synchronize()
    # the oracle on the rounded inputs; the gradient wrt the video_affine weight is taken at the rounded weight (straight through)
    Pr = dict(P)
    Pr['video_affine.linear.conv1d.weight'] = _round_bf16(P['video_affine.linear.conv1d.weight'])
    br = dict(b)
    br['vfeats'] = _round_bf16(b['vfeats'])
    O.record_relu_signs()
    if training:
        O.force_dropout(hip_dropout(seed))
    with torch.no_grad():
        O.total_loss(Pr, cfg, br, training=training)
    _, hip_masks = relu_flips(eng, B, T, 9)
    O.record_relu_signs(False)
    O.force_relu_signs(hip_masks)
    if training:
        O.force_dropout(hip_dropout(seed))
    Pg = {k: v.clone().requires_grad_(k not in O.FROZEN) for k, v in Pr.items()}
    total, (oh, osl, oel, _, _) = O.total_loss(Pg, cfg, br, training=training)
    O.force_relu_signs(None)
    O.force_dropout(None)
    assert_forced_relu_inside_noise(O)
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
    assert not bad, bad[:5]
    if not training:
        # the price of the mode, against the unrounded fp32 oracle: stated, loose
        with torch.no_grad():
            _, (fh, fsl, fel, _, _) = O.total_loss(P, cfg, b)
        assert float((sl.cpu() - fsl)[fin].abs().max()) <= 8e-2 * scale
        assert float((h.cpu() - fh).abs().max()) <= 2e-2
