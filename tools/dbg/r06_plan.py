"""Which partial slabs does the final reduction read?  VSL_DEBUG_PLAN=1 makes build_plan print every segment >= 256 KiB; this adds the parameter names."""
import os, sys
os.environ['VSL_DEBUG_PLAN'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vslnet_amd.model.VSLNet import VSLNet
from vslnet_amd.synthetic import make_configs, synthetic_batch
configs = make_configs(video_feature_dim=1024, max_pos_len=128, drop_rate=0.2, predictor='transformer')
torch.manual_seed(0)
glove = torch.randn(configs.word_size - 2, configs.word_dim).numpy()
model = VSLNet(configs, glove).cuda().train()
flat, grads = model.flat_parameters
eng = model._engine
for name, off, num, shape in eng.layout:
    print('[param] %8d %s %s' % (off, name, tuple(shape)))
b = synthetic_batch(configs, 64, 128, 20, 10, seed=1)
pad_vec, glove_vec = model.embedding_net.word_emb.pad_vec.data, model.embedding_net.word_emb.glove_vec.data
eng.forward(flat, pad_vec, glove_vec, b['word_ids'], b['char_ids'], b['vfeats'], b['v_mask'], b['q_mask'], training=True, seed=1)
torch.cuda.synchronize()
