"""Diagnostic: the 2-rank training test emulated in ONE process (shards run one after the other)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import vslnet_oracle as O
from vslnet_amd import dp
from vslnet_amd.engine import Engine, flat_from_state_dict
cfg = O.make_cfg(video_feature_dim=64, max_pos_len=48, word_size=52, drop_rate=0.2)
batch = O.synthetic_batch(cfg, B=7, T=40, Lq=6, Lc=5, seed=9, ragged=True)
def train(world, steps=3):
    P = O.random_params(cfg, seed=5)
    eng = Engine(cfg)
    flat = flat_from_state_dict(eng, P)
    opt = dp.FlatAdamW(flat, eng.layout, lr=1e-3, num_train_steps=100, clip_norm=1.0, engine=eng)
    B = batch['vfeats'].shape[0]
    inv_b, msum = dp.global_normalisers(batch['lens'].tolist())
    pad, glove = P['embedding_net.word_emb.pad_vec'].cuda(), P['embedding_net.word_emb.glove_vec'].cuda()
    glist = []
    for step in range(steps):
        tot = None
        for rank in range(world):
            sl = dp.shard_slice(B, rank, world)
            d = {k: v[sl].cuda().contiguous() for k, v in batch.items() if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B}
            grads = eng.new_flat()
            eng.forward(flat, pad, glove, d['word_ids'], d['char_ids'], d['vfeats'], d['v_mask'], d['q_mask'], training=True, seed=1000 + step, sample_offset=sl.start)
            _, dh, dsl, del_ = eng.loss(d['s_labels'], d['e_labels'], d['h_labels'], 1.0, 5.0, inv_batch=inv_b, mask_sum=msum)
            eng.backward(dh, dsl, del_, grads)
            torch.cuda.synchronize()
            tot = grads.clone() if tot is None else tot + grads
        glist.append(tot.clone())
        opt.step(tot)
    torch.cuda.synchronize()
    return eng, flat.cpu(), [g.cpu() for g in glist]
print('shards', [dp.shard_slice(7, r, 2) for r in range(2)])
eng, one, g1 = train(1)
_, two, g2 = train(2)
d = (two - one).abs()
print('max param diff %.3e' % float(d.max()))
i = int(d.argmax())
for name, off, n, _ in eng.layout:
    if off <= i < off + n:
        print('at', name, i - off, 'one', float(one[i]), 'two', float(two[i]))
for s in range(3):
    print('step', s, 'grad at idx: one %.4e two %.4e ; max |dg| %.3e' % (float(g1[s][i]), float(g2[s][i]), float((g1[s] - g2[s]).abs().max())))
