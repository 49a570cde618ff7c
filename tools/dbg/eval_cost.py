import sys, time, cProfile, pstats, io
sys.path.insert(0, '.')
import torch
import main
from vslnet_amd import data, runner
from vslnet_amd.model.VSLNet import VSLNet
parser = main.build_parser()
configs = parser.parse_args(['--task', 'synthetic', '--predictor', 'transformer', '--batch_size', '64', '--max_pos_len', '128', '--synthetic_train', '256', '--synthetic_test', '64'])
dataset, features = data.load_dataset(configs)
configs.char_size, configs.word_size = dataset['n_chars'], dataset['n_words']
dev = torch.device('cuda', 0)
model = VSLNet(configs=configs, word_vectors=dataset['word_vector']).to(dev)
test_loader = data.ResidentSplit(dataset['test_set'], features, configs, dev, train=False)
model.eval()
for _ in range(3):
    runner.eval_test(model, test_loader, dev, 'test', 1, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    runner.eval_test(model, test_loader, dev, 'test', 1, 1)
torch.cuda.synchronize()
print('eval_test: %.3f ms per call' % ((time.perf_counter() - t0) / 10 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    runner.eval_test(model, test_loader, dev, 'test', 1, 1)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18); print(s.getvalue()[:3500])

# ---- the same evaluation between training steps, as main.py runs it
from vslnet_amd import dp
model.train()
flat, grads = model.flat_parameters
eng = model._engine
opt = dp.FlatAdamW(flat, eng.layout, lr=1e-4, num_train_steps=1000, clip_norm=1.0, engine=eng)
pad_vec, glove_vec = model.embedding_net.word_emb.pad_vec.data, model.embedding_net.word_emb.glove_vec.data
gen = torch.Generator().manual_seed(1)
train_loader = data.ResidentSplit(dataset['train_set'], features, configs, dev, train=True, generator=gen)
step = 0
for rep in range(4):
    for batch in train_loader.shards(0, 1):
        step += 1
        inv_batch, mask_sum = dp.global_normalisers(batch['lens_global'])
        q_mask = (batch['word_ids'] != 0).float()
        eng.forward(flat, pad_vec, glove_vec, batch['word_ids'], batch['char_ids'], batch['vfeats'], batch['v_mask'], q_mask, training=True, seed=step, sample_offset=0)
        losses, d_h, d_sl, d_el = eng.loss(batch['s_labels'], batch['e_labels'], batch['h_labels'], 1.0, 5.0, inv_batch=inv_batch, mask_sum=mask_sum)
        eng.backward(d_h, d_sl, d_el, grads)
        opt.step(grads)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.eval()
    t1 = time.perf_counter()
    runner.eval_test(model, test_loader, dev, 'test', 1, step)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    model.train()
    t3 = time.perf_counter()
    print('after %d train steps: model.eval() %.3f ms, eval_test %.3f ms, model.train() %.3f ms' % (step, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
