"""k_attn_fwd / k_attn_bwd_long at configs[4] (B=16, T=1024): per-launch time with and without dropout (eval mode = no hash, no mask select)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vslnet_amd.model.VSLNet import VSLNet
from vslnet_amd.synthetic import make_configs, synthetic_batch
B, T = 16, 1024
for drop in (0.2, 0.0):
    configs = make_configs(video_feature_dim=1024, max_pos_len=T, drop_rate=drop, predictor='transformer')
    torch.manual_seed(0)
    glove = torch.randn(configs.word_size - 2, configs.word_dim).numpy()
    model = VSLNet(configs, glove).cuda().train()
    flat, grads = model.flat_parameters
    eng = model._engine
    b = synthetic_batch(configs, B, T, 20, 10, seed=1)
    pad_vec, glove_vec = model.embedding_net.word_emb.pad_vec.data, model.embedding_net.word_emb.glove_vec.data
    def step(i):
        eng.forward(flat, pad_vec, glove_vec, b['word_ids'], b['char_ids'], b['vfeats'], b['v_mask'], b['q_mask'], training=True, seed=i)
        _, d_h, d_sl, d_el = eng.loss(b['s_labels'], b['e_labels'], b['h_labels'], 1.0, 5.0, inv_batch=1.0 / B, mask_sum=float(b['v_mask'].sum()))
        eng.backward(d_h, d_sl, d_el, grads)
    for i in range(3):
        step(i)
    eng.profile_select('*')
    for i in range(5):
        step(10 + i)
    torch.cuda.synchronize()
    r = eng.profile_read()
    eng.profile_select(None)
    print('drop %.1f : ' % drop + '  '.join('%s %.1f us x %d' % (k, 1e3 * v[0] / v[1], v[1] // 5) for k, v in sorted(r.items()) if k in ('attn_fwd', 'attn_bwd', 'attn_out_fwd', 'qkv_bwd', 'convblock_fwd', 'convblock_bwd')))
