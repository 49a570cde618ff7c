"""Generate golden fixtures under tests/golden/ by running the REFERENCE itself (build container only).

    python oracle/make_golden.py            # needs /root/reference; never runs on the GPU box

The reference (`/root/reference/model/VSLNet_t7.py`, `layers_t7.py`, `util/*_t7.py`) is imported read-only
with one caller-side shim (`transformers.AdamW` no longer exists in transformers 5.x, VSLNet_t7.py:5).
What is written is DATA only: inputs, per-key checksums of the `state_dict` (weights are re-derived from a seed), the outputs of every sub-module on the
forward path (forward hooks), h_score / logits, both losses, every parameter gradient of
`loc + 5.0 * highlight`, and `extract_index`.  No reference source or bytecode is copied.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = '/root/reference'

from oracle import vslnet_oracle as O  # noqa: E402


def load_reference():
    import transformers
    if not hasattr(transformers, 'AdamW'):
        transformers.AdamW = torch.optim.AdamW          # VSLNet_t7.py:5 import shim (reference untouched)
    sys.path.insert(0, REF)
    from model.VSLNet_t7 import VSLNet                   # noqa
    from util import runner_utils_t7, data_loader_t7    # noqa
    return VSLNet, runner_utils_t7, data_loader_t7


CASES = {
    # name: (cfg overrides, B, T, Lq, Lc, ragged)
    'tiny_tf':  (dict(video_feature_dim=64, max_pos_len=32, word_size=52), 3, 24, 7, 6, True),
    'tiny_rnn': (dict(video_feature_dim=64, max_pos_len=32, word_size=52, predictor='rnn'), 3, 24, 7, 6, True),
    'real_tf':  (dict(video_feature_dim=1024, max_pos_len=128, word_size=52), 2, 128, 20, 10, True),
    'long_tf':  (dict(video_feature_dim=64, max_pos_len=256, word_size=52), 2, 200, 12, 5, True),
    # main_t7.py:24: "--char_dim ... 100 for activitynet"
    'chardim100_tf': (dict(video_feature_dim=64, max_pos_len=32, word_size=52, char_dim=100), 3, 24, 7, 12, True),
    # WordEmbedding(word_vectors=None): the trainable nn.Embedding branch (layers_t7.py:36, 43-44)
    'wordtable_tf': (dict(video_feature_dim=64, max_pos_len=32, word_size=52, word_table=True), 3, 24, 7, 6, True),
}


# TRAINING-mode cases (round 4): model.train(), drop_rate 0.2, every nn.Dropout call replaced by a mask drawn from a generator seeded
# with MASK_SEED + call number (oracle.train_mask) -- the placement and ORDER of the 41 (transformer) / 23 (rnn) dropout sites
# (layers_t7.py:45, 63-64, 113, 138, 168-169, 181, 185, 188-189, 227-228) is then pinned by the logits, losses and gradients.
TRAIN_CASES = {
    'train_tf':  (dict(video_feature_dim=64, max_pos_len=32, word_size=52, drop_rate=0.2), 3, 24, 7, 6, True),
    'train_rnn': (dict(video_feature_dim=64, max_pos_len=32, word_size=52, drop_rate=0.2, predictor='rnn'), 3, 24, 7, 6, True),
}

PARAM_SEED = 12345


def run_case(VSLNet, name, spec):
    over, B, T, Lq, Lc, ragged = spec
    cfg = O.make_cfg(**over)
    torch.manual_seed(12345)
    glove = np.zeros((cfg.word_size - 2, cfg.word_dim), np.float32)     # overwritten by load_state_dict
    model = VSLNet(configs=cfg, word_vectors=None if getattr(cfg, 'word_table', False) else glove)
    # weights come from the build's own seeded factory and are loaded into the reference model with
    # strict=True -- this both keeps the fixture small (no weights stored, only checksums) and pins the
    # state_dict schema (names + shapes) against the reference.
    P = O.random_params(cfg, seed=PARAM_SEED)
    model.load_state_dict(P, strict=True)
    model.eval()
    batch = O.synthetic_batch(cfg, B, T, Lq, Lc, seed=3, ragged=ragged)

    taps = {}

    def tap(key):
        def hook(_m, _inp, out):
            taps.setdefault(key, []).append(out.detach().clone() if torch.is_tensor(out) else
                                            [o.detach().clone() for o in out])
        return hook

    hooks = []
    for key, mod in [('video_affine', model.video_affine), ('embedding_net', model.embedding_net),
                     ('word_emb', model.embedding_net.word_emb), ('char_emb', model.embedding_net.char_emb),
                     ('feature_encoder', model.feature_encoder),
                     ('fe_conv_block', model.feature_encoder.conv_block),
                     ('cq_attention', model.cq_attention), ('cq_concat', model.cq_concat),
                     ('highlight_layer', model.highlight_layer)]:
        hooks.append(mod.register_forward_hook(tap(key)))
    if cfg.predictor != 'rnn':
        hooks.append(model.predictor.encoder.register_forward_hook(tap('pred_encoder')))
    else:
        hooks.append(model.predictor.start_encoder.register_forward_hook(tap('pred_start_rnn')))
        hooks.append(model.predictor.end_encoder.register_forward_hook(tap('pred_end_rnn')))

    h, sl, el = model(batch['word_ids'], batch['char_ids'], batch['vfeats'], batch['v_mask'], batch['q_mask'])
    hl = model.compute_highlight_loss(h, batch['h_labels'], batch['v_mask'])
    loc = model.compute_loss(sl, el, batch['s_labels'], batch['e_labels'])
    total = loc + 5.0 * hl
    model.zero_grad()
    total.backward()
    si, ei = model.extract_index(sl, el)
    for hk in hooks:
        hk.remove()

    out = {}
    for k, v in batch.items():
        out['in.' + k] = v.numpy()
    out['param_seed'] = np.array(PARAM_SEED)
    for k, v in model.state_dict().items():
        v64 = v.detach().double()
        out['sdsum.' + k] = np.array([float(v64.sum()), float(v64.abs().sum()), float(v64.flatten()[-1])])
    for n, p in model.named_parameters():
        if p.requires_grad:
            out['grad.' + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    out['out.h_score'] = h.detach().numpy()
    out['out.start_logits'] = sl.detach().numpy()
    out['out.end_logits'] = el.detach().numpy()
    out['out.highlight_loss'] = hl.detach().numpy()
    out['out.loc_loss'] = loc.detach().numpy()
    out['out.start_index'] = si.numpy()
    out['out.end_index'] = ei.numpy()
    for k, lst in taps.items():
        for i, t in enumerate(lst):
            out['tap.%s.%d' % (k, i)] = t.numpy()
    out['cfg'] = np.array(repr(vars(cfg)))
    path = os.path.join(ROOT, 'tests', 'golden', name + '.npz')
    np.savez_compressed(path, **out)
    print(name, 'written', os.path.getsize(path) // 1024, 'KiB', 'loss', float(total))


def run_train_case(VSLNet, name, spec):
    """The reference in TRAINING mode under seeded per-call dropout masks.  The masks are defined in the layout the path's tensors
    have outside the conv block, (B, L, C); the conv block's dropout (layers_t7.py:138) sees (B, C, L), so its masks are transposed."""
    over, B, T, Lq, Lc, ragged = spec
    cfg = O.make_cfg(**over)
    torch.manual_seed(12345)
    glove = np.zeros((cfg.word_size - 2, cfg.word_dim), np.float32)
    model = VSLNet(configs=cfg, word_vectors=glove)
    P = O.random_params(cfg, seed=PARAM_SEED)
    model.load_state_dict(P, strict=True)
    model.train()
    batch = O.synthetic_batch(cfg, B, T, Lq, Lc, seed=4, ragged=ragged)
    conv_drops = {id(model.feature_encoder.conv_block.dropout)}
    if cfg.predictor != 'rnn':
        conv_drops.add(id(model.predictor.encoder.conv_block.dropout))
    calls = [0]
    orig = torch.nn.Dropout.forward

    def patched(self, x):
        if not self.training or self.p == 0.0:
            return x
        n = calls[0]
        calls[0] += 1
        if id(self) in conv_drops:
            return x * O.train_mask(n, tuple(x.transpose(1, 2).shape), self.p).transpose(1, 2)
        return x * O.train_mask(n, tuple(x.shape), self.p)

    torch.nn.Dropout.forward = patched
    try:
        h, sl, el = model(batch['word_ids'], batch['char_ids'], batch['vfeats'], batch['v_mask'], batch['q_mask'])
        hl = model.compute_highlight_loss(h, batch['h_labels'], batch['v_mask'])
        loc = model.compute_loss(sl, el, batch['s_labels'], batch['e_labels'])
        total = loc + 5.0 * hl
        model.zero_grad()
        total.backward()
    finally:
        torch.nn.Dropout.forward = orig
    out = {}
    for k, v in batch.items():
        out['in.' + k] = v.numpy()
    out['param_seed'] = np.array(PARAM_SEED)
    out['n_dropout_calls'] = np.array(calls[0])
    for k, c in sd_checksums(model.state_dict()).items():
        out['sdsum.' + k] = c
    for n, p in model.named_parameters():
        if p.requires_grad:
            out['grad.' + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    out['out.h_score'] = h.detach().numpy()
    out['out.start_logits'] = sl.detach().numpy()
    out['out.end_logits'] = el.detach().numpy()
    out['out.highlight_loss'] = hl.detach().numpy()
    out['out.loc_loss'] = loc.detach().numpy()
    out['cfg'] = np.array(repr(vars(cfg)))
    path = os.path.join(ROOT, 'tests', 'golden', name + '.npz')
    np.savez_compressed(path, **out)
    print(name, 'written', os.path.getsize(path) // 1024, 'KiB', 'loss', float(total.detach()), 'dropout calls', calls[0])


def run_host_helpers(ru, dl):
    """Pins for the host-side helpers the callers own (SURVEY 8c last row)."""
    rs = np.random.RandomState(5)
    lens = torch.tensor(rs.randint(5, 40, size=9), dtype=torch.int64)
    mask = ru.convert_length_to_mask(lens)
    # collate labels: run the reference's train_collate_fn on synthetic records
    data = []
    for i in range(9):
        L = int(lens[i])
        s = int(rs.randint(0, L // 2 + 1))
        e = min(L - 1, s + int(rs.randint(0, L // 2 + 1)))
        feat = rs.randn(L, 16).astype(np.float32)
        nw = int(rs.randint(2, 7))
        w_ids = [int(x) for x in rs.randint(1, 50, size=nw)]
        c_ids = [[int(x) for x in rs.randint(1, 30, size=int(rs.randint(1, 6)))] for _ in range(nw)]
        data.append(({'vid': str(i)}, feat, w_ids, c_ids, s, e))
    _, vfeats, vlens, word_ids, char_ids, s_l, e_l, h_l = dl.train_collate_fn(data)
    out = dict(lens=lens.numpy(), mask=mask.numpy(), vlens=vlens.numpy(), s_labels=s_l.numpy(),
               e_labels=e_l.numpy(), h_labels=h_l.numpy(), word_ids=word_ids.numpy(), char_ids=char_ids.numpy(),
               vfeats=vfeats.numpy())
    # IoU helper pins
    pairs = rs.rand(20, 4).astype(np.float64) * 30
    ious = [ru.calculate_iou(sorted(p[:2]), sorted(p[2:])) for p in pairs]
    out['iou_pairs'] = pairs
    out['ious'] = np.array(ious)
    path = os.path.join(ROOT, 'tests', 'golden', 'host_helpers.npz')
    np.savez_compressed(path, **out)
    print('host_helpers written')


def host_pipeline_records(seed=11, n=13, dv=12):
    """Synthetic dataset records in the reference's format (data_gen.py:178-181) + per-video features.  Deterministic;
    tests/test_data_pipeline.py rebuilds the same inputs from this function (it is pure numpy, no reference import)."""
    rs = np.random.RandomState(seed)
    feats, records = {}, []
    for i in range(n):
        L = int(rs.randint(4, 30))
        feats['v%d' % i] = rs.randn(L, dv).astype(np.float32)
        dur = float(rs.uniform(5, 60))
        st = float(rs.uniform(0, dur * 0.6))
        et = float(min(dur, st + rs.uniform(0.5, dur * 0.5)))
        nw = int(rs.randint(1, 9))
        records.append({'sample_id': i, 'vid': 'v%d' % i, 's_time': st, 'e_time': et, 'duration': dur, 'v_len': L,
                        's_ind': int(rs.randint(0, L)), 'e_ind': 0,
                        'w_ids': [int(x) for x in rs.randint(1, 40, size=nw)],
                        'c_ids': [[int(x) for x in rs.randint(1, 25, size=int(rs.randint(1, 7)))] for _ in range(nw)]})
        records[-1]['e_ind'] = int(rs.randint(records[-1]['s_ind'], L))
    return records, feats


def run_host_pipeline(ru, dl):
    """Pins for the batch pipeline / eval helpers of SURVEY 8(f) rows 1-2: collate functions (data_loader_t7.py:24-81),
    padding (data_util.py:117-159), feature sampling (:59-73), index<->time (:92-114) and the eval metrics
    (runner_utils_t7.py:55-101)."""
    from util import data_util as du
    records, feats = host_pipeline_records()
    ds = dl.Dataset(records, feats)
    out = {}
    items = [ds[i] for i in range(len(ds))]
    _, vf, vl, wi, ci, sl, el, hl = dl.train_collate_fn(items)
    out.update(tr_vfeats=vf.numpy(), tr_vlens=vl.numpy(), tr_word_ids=wi.numpy(), tr_char_ids=ci.numpy(),
               tr_s=sl.numpy(), tr_e=el.numpy(), tr_h=hl.numpy())
    _, vf, vl, wi, ci = dl.test_collate_fn(items[:5])
    out.update(te_vfeats=vf.numpy(), te_vlens=vl.numpy(), te_word_ids=wi.numpy(), te_char_ids=ci.numpy())
    rs = np.random.RandomState(3)
    for k, (L, m) in enumerate([(50, 16), (17, 16), (16, 16), (9, 16), (200, 32), (33, 32)]):
        x = rs.randn(L, 5).astype(np.float32)
        out['samp%d_in' % k] = x
        out['samp%d_out' % k] = np.asarray(du.visual_feature_sampling(x, m))
        out['samp%d_m' % k] = np.array(m)
    t2i, i2t = [], []
    for _ in range(12):
        n = int(rs.randint(3, 40)); dur = float(rs.uniform(4, 90))
        a = float(rs.uniform(0, dur * 0.7)); b = float(min(dur, a + rs.uniform(0.2, dur * 0.5)))
        s, e, _ = du.time_to_index(a, b, n, dur)
        t2i.append([a, b, n, dur, s, e])
        si = int(rs.randint(0, n)); ei = int(rs.randint(si, n))
        st, et = du.index_to_time(si, ei, n, dur)
        i2t.append([si, ei, n, dur, float(st), float(et)])
    out['time_to_index'] = np.array(t2i, dtype=np.float64)
    out['index_to_time'] = np.array(i2t, dtype=np.float64)
    ious = rs.rand(57)
    out['metric_ious'] = ious
    out['metric_acc'] = np.array([ru.calculate_iou_accuracy(list(ious), t) for t in (0.3, 0.5, 0.7)] + [float(np.mean(ious) * 100.0)])
    path = os.path.join(ROOT, 'tests', 'golden', 'host_pipeline.npz')
    np.savez_compressed(path, **out)
    print('host_pipeline written', os.path.getsize(path) // 1024, 'KiB')


REF_CKPT_ARGV = ['--task', 'synthetic', '--predictor', 'transformer', '--max_pos_len', '32', '--video_feature_dim', '64', '--batch_size', '16',
                 '--synthetic_train', '8', '--synthetic_test', '48', '--drop_rate', '0.1']


def run_ref_checkpoint(VSLNet, ru, dl):
    """A checkpoint written BY THE REFERENCE (`torch.save(model.state_dict())`, main_t7.py:125) together with the `configs.json` it writes
    (main_t7.py:81) and the metrics its own `eval_test` (runner_utils_t7.py:71-101) reports for it on the synthetic test split -- the file
    `main.py --mode test` must load (main_t7.py:140-144) and reproduce.  The dataset is what `--task synthetic` builds from the same argv."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location('vsl_repo_main', os.path.join(ROOT, 'main.py'))      # (`import main` would find the reference's TF main.py)
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    from vslnet_amd import data
    cfg = cli.build_parser().parse_args(REF_CKPT_ARGV)
    dataset, feats = data.load_dataset(cfg)
    cfg.char_size, cfg.word_size = dataset['n_chars'], dataset['n_words']
    cfg.num_train_steps = 1
    torch.manual_seed(4321)
    model = VSLNet(configs=cfg, word_vectors=dataset['word_vector'])
    model.eval()
    out_dir = os.path.join(ROOT, 'tests', 'golden', 'ref_checkpoint')
    os.makedirs(out_dir, exist_ok=True)
    torch.save(model.state_dict(), os.path.join(out_dir, 'vslnet_5.t7'))
    with open(os.path.join(out_dir, 'configs.json'), 'w', encoding='utf-8') as f:
        f.write(json.dumps(vars(cfg), indent=4, sort_keys=True))
    loader = dl.get_test_loader(dataset['test_set'], feats, cfg)
    r1i3, r1i5, r1i7, mi, _ = ru.eval_test(model, loader, 'cpu', mode='test')
    with open(os.path.join(out_dir, 'reference_metrics.json'), 'w') as f:
        json.dump({'argv': REF_CKPT_ARGV, 'r1i3': r1i3, 'r1i5': r1i5, 'r1i7': r1i7, 'mIoU': mi, 'n_test': len(dataset['test_set'])}, f, indent=1)
    print('ref_checkpoint written:', os.path.getsize(os.path.join(out_dir, 'vslnet_5.t7')) // 1024, 'KiB', r1i3, r1i5, r1i7, mi)


INIT_SEED = 777


def sd_checksums(sd):
    """per-tensor (sum, abs-sum, last element) in float64: what the fixtures keep of a state_dict"""
    out = {}
    for k, v in sd.items():
        v64 = v.detach().double()
        out[k] = np.array([float(v64.sum()), float(v64.abs().sum()), float(v64.flatten()[-1])])
    return out


def run_init(VSLNet):
    """a19 (VSLNet_t7.py:42-50): the reference constructed under torch.manual_seed(INIT_SEED) -- checksums of its freshly
    initialised state_dict for both predictor heads.  The build's module must reproduce them bit for bit under the same seed."""
    out = {'seed': np.int64(INIT_SEED)}
    for pred in ('transformer', 'rnn', 'transformer_wordtable'):
        cfg = O.make_cfg(video_feature_dim=64, max_pos_len=32, word_size=52, predictor=pred.split('_')[0])
        glove = None if pred.endswith('wordtable') else np.random.RandomState(0).randn(cfg.word_size - 2, cfg.word_dim).astype(np.float32)
        torch.manual_seed(INIT_SEED)
        sd = VSLNet(configs=cfg, word_vectors=glove).state_dict()
        out['keys.' + pred] = np.array(list(sd.keys()))
        for k, c in sd_checksums(sd).items():
            out['%s.%s' % (pred, k)] = c
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'init.npz'), **out)
    print('init.npz written')


def main():
    VSLNet, ru, dl = load_reference()
    if len(sys.argv) > 1 and sys.argv[1] == 'host_pipeline':
        run_host_pipeline(ru, dl)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'init':
        run_init(VSLNet)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'ref_checkpoint':
        run_ref_checkpoint(VSLNet, ru, dl)
        return
    if len(sys.argv) > 2 and sys.argv[1] == 'case':          # one model fixture, the others untouched
        torch.set_num_threads(8)
        if sys.argv[2] in TRAIN_CASES:
            run_train_case(VSLNet, sys.argv[2], TRAIN_CASES[sys.argv[2]])
        else:
            run_case(VSLNet, sys.argv[2], CASES[sys.argv[2]])
        return
    os.makedirs(os.path.join(ROOT, 'tests', 'golden'), exist_ok=True)
    torch.set_num_threads(8)
    for name, spec in CASES.items():
        run_case(VSLNet, name, spec)
    for name, spec in TRAIN_CASES.items():
        run_train_case(VSLNet, name, spec)
    run_host_helpers(ru, dl)
    run_host_pipeline(ru, dl)
    run_init(VSLNet)
    run_ref_checkpoint(VSLNet, ru, dl)


if __name__ == '__main__':
    main()
