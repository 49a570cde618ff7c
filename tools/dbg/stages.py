"""Diagnostic: eval-mode forward stages vs the oracle for a shape.  usage: python tools/dbg/stages.py T Lq Lc [B] [Dv]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import vslnet_oracle as O
from vslnet_amd.engine import Engine, flat_from_state_dict
T, Lq, Lc = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 2
Dv = int(sys.argv[5]) if len(sys.argv) > 5 else 64
cfg = O.make_cfg(video_feature_dim=Dv, max_pos_len=max(T, Lq), word_size=102)
P = O.random_params(cfg, seed=11)
b = O.synthetic_batch(cfg, B, T, Lq, Lc, seed=12, ragged=B > 1)
eng = Engine(cfg)
flat = flat_from_state_dict(eng, P)
dev = lambda t: t.cuda().contiguous()
h, sl, el = eng.forward(flat, dev(P['embedding_net.word_emb.pad_vec']), dev(P['embedding_net.word_emb.glove_vec']), dev(b['word_ids']),
                        dev(b['char_ids']), dev(b['vfeats']), dev(b['v_mask']), dev(b['q_mask']))
torch.cuda.synchronize()
want = {}
with torch.no_grad():
    oh, osl, oel = O.forward(P, cfg, b['word_ids'], b['char_ids'], b['vfeats'], b['v_mask'], b['q_mask'], want=want)
    econ = torch.cat([O.word_embedding(P, b['word_ids'], 0, False), O.char_embedding(P, b['char_ids'], 0, False)], -1)
def chk(name, got, ref):
    print('%-16s %.3e' % (name, float((got.cpu() - ref).abs().max())))
chk('emb_concat', eng.ws_view('emb_concat', (B, Lq, cfg.word_dim + 100)), econ)
chk('embedding_net', eng.ws_view('embedding_net', (B, Lq, 128)), want['embedding_net'])
chk('video_affine', eng.ws_view('video_affine', (B, T, 128)), want['video_affine'])
chk('venc', eng.ws_view('venc', (B, T, 128)), want['venc'])
chk('qenc', eng.ws_view('qenc', (B, Lq, 128)), want['qenc'])
Cq = want['cq_parts']
for nm, key in (('cq_score', 'score'), ('cq_srow', 'srow'), ('cq_scol', 'scol')):
    if key in Cq:
        chk(nm, eng.ws_view(nm, (B, T, Lq)), Cq[key])
print('cq keys', list(Cq.keys()))
chk('cq_attention', eng.ws_view('cq_attention', (B, T, 128)), want['cq_attention'])
chk('cq_concat', eng.ws_view('cq_concat', (B, T, 128)), want['cq_concat'])
fin = osl.abs() < 1e29
print('logits %.3e' % float((sl.cpu() - osl)[fin].abs().max()))
