"""Batch pipeline + eval helpers against outputs of the reference's own functions
(tests/golden/host_pipeline.npz, written by oracle/make_golden.py: run_host_pipeline from util/data_loader_t7.py,
util/data_util.py and util/runner_utils_t7.py).  Integer outputs must match bit for bit."""
import os

import numpy as np
import torch

from oracle.make_golden import host_pipeline_records       # pure-numpy input generator (no reference import)
from tests.helpers import GOLDEN
from vslnet_amd import data, runner


def _z():
    return np.load(os.path.join(GOLDEN, 'host_pipeline.npz'))


def test_collate_functions_match_reference():
    z = _z()
    records, feats = host_pipeline_records()
    ds = data.VideoQueryDataset(records, feats)
    items = [ds[i] for i in range(len(ds))]
    recs, vf, vl, wi, ci, s, e, h = data.collate_train(items)
    assert [r['sample_id'] for r in recs] == list(range(len(records)))
    for got, key in ((vf, 'tr_vfeats'), (vl, 'tr_vlens'), (wi, 'tr_word_ids'), (ci, 'tr_char_ids'), (s, 'tr_s'), (e, 'tr_e'), (h, 'tr_h')):
        assert got.dtype == torch.from_numpy(z[key]).dtype, key
        assert np.array_equal(got.numpy(), z[key]), key
    _, vf, vl, wi, ci = data.collate_test(items[:5])
    for got, key in ((vf, 'te_vfeats'), (vl, 'te_vlens'), (wi, 'te_word_ids'), (ci, 'te_char_ids')):
        assert np.array_equal(got.numpy(), z[key]), key


def test_highlight_targets_match_first_fixture_too():
    z = np.load(os.path.join(GOLDEN, 'host_helpers.npz'))
    h = data.highlight_targets(z['s_labels'], z['e_labels'], z['vlens'], int(z['vlens'].max()))
    assert np.array_equal(h, z['h_labels'])


def test_feature_resampling():
    z = _z()
    for k in range(6):
        out = data.resample_features(z['samp%d_in' % k], int(z['samp%d_m' % k]))
        assert out.shape == z['samp%d_out' % k].shape
        assert np.allclose(out, z['samp%d_out' % k], rtol=0, atol=1e-6), k
    x = z['samp3_in']
    assert data.resample_features(x, 16) is x                  # short videos are passed through untouched


def test_index_time_round_trips():
    z = _z()
    for a, b, n, dur, s, e in z['time_to_index']:
        gs, ge, ov = data.time_to_index(float(a), float(b), int(n), float(dur))
        assert (gs, ge) == (int(s), int(e))
        assert ov.shape == (int(n), int(n))
    for si, ei, n, dur, st, et in z['index_to_time']:
        gs, ge = data.index_to_time(int(si), int(ei), int(n), float(dur))
        assert float(gs) == st and float(ge) == et


def test_metrics():
    z = _z()
    r = runner.summarise(list(z['metric_ious']), epoch=2, global_step=30)
    assert np.allclose(r[:4], z['metric_acc'], rtol=0, atol=1e-12)
    assert r[4].startswith('Epoch 2, Step 30:\n') and 'mean IoU' in r[4]
    h = np.load(os.path.join(GOLDEN, 'host_helpers.npz'))
    for p, want in zip(h['iou_pairs'], h['ious']):
        assert runner.calculate_iou(sorted(p[:2]), sorted(p[2:])) == want


def test_checkpoint_housekeeping(tmp_path):
    for step in (5, 40, 12, 300, 7):
        (tmp_path / ('vslnet_%d.t7' % step)).write_bytes(b'x')
    assert os.path.basename(runner.get_last_checkpoint(str(tmp_path))) == 'vslnet_300.t7'
    runner.filter_checkpoints(str(tmp_path), max_to_keep=3)
    assert sorted(os.listdir(tmp_path)) == ['vslnet_12.t7', 'vslnet_300.t7', 'vslnet_40.t7']


def test_synthetic_dataset_has_reference_record_format():
    from vslnet_amd.synthetic import make_configs
    cfg = make_configs(video_feature_dim=32, max_pos_len=24, batch_size=4, task='synthetic')
    ds, feats = data.synthetic_dataset(cfg, n_train=9, n_test=3)
    assert set(ds) >= {'train_set', 'val_set', 'test_set', 'word_vector', 'n_words', 'n_chars'}
    r = ds['train_set'][0]
    assert set(r) >= {'sample_id', 'vid', 's_time', 'e_time', 'duration', 's_ind', 'e_ind', 'v_len', 'w_ids', 'c_ids'}
    batch = next(iter(data.get_train_loader(ds['train_set'], feats, cfg)))
    assert batch[1].shape[0] == 4 and batch[1].shape[2] == 32 and batch[7].shape == batch[1].shape[:2]
    assert int(batch[3].max()) < ds['n_words'] and int(batch[4].max()) < ds['n_chars']
    s, e, _ = data.time_to_index(r['s_time'], r['e_time'], r['v_len'], r['duration'])
    assert (s, e) == (r['s_ind'], r['e_ind'])                   # times and indices of a record are consistent


def _small_cfg(batch_size):
    import argparse
    return argparse.Namespace(video_feature_dim=16, max_pos_len=24, batch_size=batch_size, word_dim=8, extend=0.1, seed=3)


def test_resident_split_yields_the_loader_batches():
    """HBM-resident split (device gathers) == DataLoader + collate functions, batch for batch: same shapes (batch-wide
    padding widths), same contents, for the train tuple, the test tuple and the per-rank shards of the fused loop."""
    cfg = _small_cfg(7)                                          # 40 records: ragged last batch
    ds, feats = data.synthetic_dataset(cfg, n_train=40, n_test=13, seed=5)
    order = np.random.RandomState(0).permutation(40)
    ref = torch.utils.data.DataLoader(data.VideoQueryDataset(ds['train_set'], feats), batch_size=cfg.batch_size, sampler=list(order),
                                      collate_fn=lambda b: data.collate_train(b, False, cfg.extend))
    split = data.ResidentSplit(ds['train_set'], feats, cfg, 'cpu', train=True)
    got = list(split.shards(0, 1, order=order))
    assert len(got) == len(ref) == len(split)
    for want, b in zip(ref, got):
        assert [r['sample_id'] for r in want[0]] == [r['sample_id'] for r in b['records']]
        assert np.array_equal(want[2].numpy(), b['lens_global'])
        for w, g in zip(want[1:2] + want[3:], (b['vfeats'], b['word_ids'], b['char_ids'], b['s_labels'], b['e_labels'], b['h_labels'])):
            assert w.shape == g.shape and w.dtype == g.dtype and torch.equal(w, g)
        assert torch.equal(runner.convert_length_to_mask(want[2]), b['v_mask'])
    # two ranks: contiguous halves of every global batch, global widths
    for want, b0, b1 in zip(ref, split.shards(0, 2, order=order), split.shards(1, 2, order=order)):
        assert torch.equal(want[1], torch.cat([b0['vfeats'], b1['vfeats']]))
        assert torch.equal(want[7], torch.cat([b0['h_labels'], b1['h_labels']]))
        assert np.array_equal(b0['lens_global'], b1['lens_global'])
    # the host-loader adapter produces the same dicts
    for b, l in zip(got, data.loader_shards(ref, 'cpu')):
        assert all(torch.equal(b[k], l[k]) for k in ('vfeats', 'v_mask', 'word_ids', 'char_ids', 's_labels', 'e_labels', 'h_labels'))
    # evaluation tuples, in dataset order
    test_ref = data.get_test_loader(ds['test_set'], feats, cfg)
    for want, have in zip(test_ref, data.ResidentSplit(ds['test_set'], feats, cfg, 'cpu', train=False)):
        assert len(have) == 5 and all(torch.equal(w, h) for w, h in zip(want[1:], have[1:]))
    # training order: a permutation drawn from the generator, reproducible
    a = data.ResidentSplit(ds['train_set'], feats, cfg, 'cpu', train=True, generator=torch.Generator().manual_seed(1))
    b = data.ResidentSplit(ds['train_set'], feats, cfg, 'cpu', train=True, generator=torch.Generator().manual_seed(1))
    ids = lambda sp: [r['sample_id'] for bt in sp for r in bt[0]]
    ia = ids(a)
    assert sorted(ia) == list(range(40)) and ia == ids(b) and ia != ids(a)      # second epoch: a new permutation


def test_dataset_is_validated_against_the_engine_limits_up_front():
    """One 129-word query, a 41-character token or an out-of-range id must stop the run at load time with the record named,
    not abort vsl_forward in the middle of an epoch (the reference would raise IndexError from nn.Embedding)."""
    import copy
    import pytest
    from types import SimpleNamespace
    from vslnet_amd import data
    cfg = SimpleNamespace(max_pos_len=256, char_dim=50)
    rec = {'sample_id': 7, 'vid': 'v', 'w_ids': [2, 3, 4], 'c_ids': [[2, 3], [4], [5, 6, 7]], 's_ind': 1, 'e_ind': 2}
    ds = {'n_words': 10, 'n_chars': 9, 'train_set': [rec], 'val_set': None, 'test_set': []}
    data.validate_dataset(ds, cfg)
    for mutate, msg in ((lambda r: r.update(w_ids=[2] * 129, c_ids=[[2]] * 129), 'query words'),
                        (lambda r: r.update(w_ids=[2, 3, 10]), 'word id'),
                        (lambda r: r['c_ids'].__setitem__(0, [2] * 41), 'characters'),
                        (lambda r: r['c_ids'].__setitem__(1, [9]), 'char id')):
        bad = copy.deepcopy(ds)
        mutate(bad['train_set'][0])
        with pytest.raises(ValueError, match=msg):
            data.validate_dataset(bad, cfg)
    with pytest.raises(ValueError, match='char_dim'):
        data.validate_dataset(ds, SimpleNamespace(max_pos_len=128, char_dim=129))
    data.validate_dataset(ds, SimpleNamespace(max_pos_len=128, char_dim=100))      # main_t7.py:24's ActivityNet setting
