"""Diagnostic: training-mode gradients of uneven shards (sample_offset) vs the full batch, per parameter.  Usage: python tools/dbg/shard_sum.py [T] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import vslnet_oracle as O
from vslnet_amd.engine import Engine, flat_from_state_dict
from tests.helpers import hip_relu_masks
T = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 7
cfg = O.make_cfg(video_feature_dim=64, max_pos_len=max(48, T), word_size=52, drop_rate=float(os.environ.get('DROP', '0.2')))
P = O.random_params(cfg, seed=5)
b = O.synthetic_batch(cfg, B=B, T=T, Lq=6, Lc=5, seed=9, ragged=True)
eng = Engine(cfg)
flat = flat_from_state_dict(eng, P)
pad, glove = P['embedding_net.word_emb.pad_vec'].cuda(), P['embedding_net.word_emb.glove_vec'].cuda()
def run(lo, hi):
    d = {k: v[lo:hi].cuda().contiguous() for k, v in b.items() if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B}
    h, sl, el = eng.forward(flat, pad, glove, d['word_ids'], d['char_ids'], d['vfeats'], d['v_mask'], d['q_mask'], training=True, seed=1000, sample_offset=lo)
    _, dh, dsl, del_ = eng.loss(d['s_labels'], d['e_labels'], d['h_labels'], 1.0, 5.0, inv_batch=1.0 / B, mask_sum=float(b['v_mask'].sum()))
    g = eng.backward(dh, dsl, del_, eng.new_flat())
    torch.cuda.synchronize()
    masks = hip_relu_masks(eng, hi - lo, T, 6)
    taps = {n: eng.ws_view(n, (hi - lo, T, 128)).clone() for n in ('venc_x0', 'venc_y0', 'venc_y1', 'venc_y2', 'venc_y3', 'venc', 'gated', 'p1_y3', 'pred_s', 'pred_e')}
    return g.clone(), sl.clone(), masks, taps
gf, slf, mf, tf = run(0, B)
cut = int(os.environ.get("CUT", B // 2))
g1, sl1, m1, t1 = run(0, cut)
g2, sl2, m2, t2 = run(cut, B)
print('logit diff shard1 %.3e shard2 %.3e' % (float((sl1 - slf[:cut]).abs().max()), float((sl2 - slf[cut:]).abs().max())))
for n in tf:
    print('%-10s %.3e %.3e' % (n, float((t1[n] - tf[n][:cut]).abs().max()), float((t2[n] - tf[n][cut:]).abs().max())))
flips = sum(int((a != f[:cut]).sum()) for a, f in zip(m1, mf)) + sum(int((a != f[cut:]).sum()) for a, f in zip(m2, mf))
print('relu flips', flips, 'per site', [int((a != f[:cut]).sum()) + int((b2 != f[cut:]).sum()) for a, b2, f in zip(m1, m2, mf)])
gs = g1 + g2
vf, vs = eng.views(gf), eng.views(gs)
rows = []
for k in vf:
    d = float((vf[k] - vs[k]).abs().max()); s = float(vf[k].abs().max())
    rows.append((d / (s + 1e-12), d, s, k))
for r in sorted(rows, reverse=True)[:12]:
    print('%.3e  abs %.3e  max %.3e  %s' % r)
print('--- elementwise Adam sensitivity |dg| / (|g| + 1e-6)')
rows = []
for k in vf:
    a, s_ = vf[k].flatten(), vs[k].flatten()
    sens = (a - s_).abs() / (a.abs() + 1e-6)
    i = int(sens.argmax())
    rows.append((float(sens[i]), float(a[i]), float(s_[i]), i, k))
for r in sorted(rows, reverse=True)[:10]:
    print('%.3e  full %.4e  shards %.4e  idx %d  %s' % r)
