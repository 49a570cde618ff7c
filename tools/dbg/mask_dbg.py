import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import vslnet_oracle as O
from vslnet_amd.engine import Engine, flat_from_state_dict
from tests.helpers import hip_relu_masks
T, B, cut = 40, 7, 4
cfg = O.make_cfg(video_feature_dim=64, max_pos_len=48, word_size=52, drop_rate=0.0)
P = O.random_params(cfg, seed=5)
b = O.synthetic_batch(cfg, B=B, T=T, Lq=6, Lc=5, seed=9, ragged=True)
pad, glove = P['embedding_net.word_emb.pad_vec'].cuda(), P['embedding_net.word_emb.glove_vec'].cuda()
def fwd(eng, flat, lo, hi):
    d = {k: v[lo:hi].cuda().contiguous() for k, v in b.items() if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B}
    eng.forward(flat, pad, glove, d['word_ids'], d['char_ids'], d['vfeats'], d['v_mask'], d['q_mask'], training=False, seed=1, sample_offset=lo)
    torch.cuda.synchronize()
    return hip_relu_masks(eng, hi - lo, T, 6), eng.ws_view('venc_y0', (hi - lo, T, 128)).clone().cpu()
def cmp(a, f, lo, hi):
    return [int((x != y[lo:hi]).sum()) for x, y in zip(a, f)]
eA = Engine(cfg); fA = flat_from_state_dict(eA, P)
mA, yA = fwd(eA, fA, 0, B)
eB = Engine(cfg); fB = flat_from_state_dict(eB, P)
mB, yB = fwd(eB, fB, 0, cut)
print('fresh engines: shard1 vs full', cmp(mB, mA, 0, cut), 'y0 diff', float((yB - yA[:cut]).abs().max()))
mC, yC = fwd(eB, fB, cut, B)
print('fresh engine B: shard2 vs full', cmp(mC, mA, cut, B), 'y0 diff', float((yC - yA[cut:]).abs().max()))
mD, yD = fwd(eA, fA, 0, B)
print('engine A again full vs full', cmp(mD, mA, 0, B))
# oracle decisions
O.record_relu_signs()
with torch.no_grad():
    O.forward(P, cfg, b['word_ids'], b['char_ids'], b['vfeats'], b['v_mask'], b['q_mask'])
sites = [sg for _, sg in O.RELU_SIGNS]
O.record_relu_signs(False)
print('full vs oracle', [int((m != s).sum()) for m, s in zip(mA, sites)])
print('shard1 vs oracle', [int((m != s[:cut]).sum()) for m, s in zip(mB, sites)])
d = (mD[0] != mA[0]).view(B * T, 128)
rows = d.any(1).nonzero().flatten().tolist()
print('site0 rows with differences between two identical runs:', rows[:80])
cols = d.any(0).nonzero().flatten().tolist()
print('cols', cols[:64])
x0 = eA.ws_view('venc_x0', (B, T, 128)).cpu(); y0 = eA.ws_view('venc_y0', (B, T, 128)).cpu()
ref = (y0 - x0) > 0
for nm, m in (('runA1', mA[0]), ('runA2', mD[0])):
    bad = (m != ref).view(B * T, 128)
    print(nm, 'wrong bits', int(bad.sum()), 'rows', bad.any(1).nonzero().flatten().tolist()[:40], 'cols', sorted(set((bad.any(0).nonzero().flatten() // 16).tolist())))
print('--- sentinel test')
ws = eA._ws
ws.view(torch.int32).fill_(0x5A5A5A5A)
mE, yE = fwd(eA, fA, 0, B)
for site in range(4):
    words = eA.ws_view('relu_venc_%d' % site, (B * T * 4,)).view(torch.int32).cpu().view(B * T, 4)
    lo, hi = words & 0xFFFF, (words >> 16) & 0xFFFF
    halves = torch.stack([lo[:, 0], hi[:, 0], lo[:, 1], hi[:, 1], lo[:, 2], hi[:, 2], lo[:, 3], hi[:, 3]], 1)   # (R, 8) = wave index
    miss = halves == 0x5A5A
    print('site', site, 'unwritten halves', int(miss.sum()), 'by wave', miss.sum(0).tolist(), 'rows', miss.any(1).nonzero().flatten().tolist()[:30])
