"""Round 6: the query branch as two sample-local launches (vslnet_amd/csrc/kernels_query.hip) and the critical-path ledger.

* which path a shape takes: Lq <= 32 -> `query_fwd` / `query_bwd` (no linear_fwd, three conv-block launches per direction instead of four);
  Lq > 32 -> the row-tile launches;
* the sample-local path against the oracle on shapes chosen for ITS edges: Lq = 1, Lq = 32 (a full window), an Embedding width that is not a
  multiple of 16 (the K tail of the chunked linear), char_dim != 50, ragged query lengths with PAD words inside the window, training mode;
* `vsl_profile_launch`: every launch of a step with its stream, its dependencies and sane timestamps (tools/critical_path.py builds
  profiles/r06_critical_path.txt from these records); the tool's chain arithmetic on a hand-made record list (CPU)."""
import io
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_ledger_chain_arithmetic_on_a_hand_made_step():
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import critical_path as cp
    # two streams: a (0 -> 10), b (12 -> 20, behind a), side kernel c on stream 1 (11 -> 40, behind a), d on stream 0 behind b and c (46 -> 50)
    recs = [dict(name='a', stream=0, start_us=100.0, stop_us=110.0, host_us=0.0, deps=[]),
            dict(name='b', stream=0, start_us=112.0, stop_us=120.0, host_us=1.0, deps=[0]),
            dict(name='c', stream=1, start_us=111.0, stop_us=140.0, host_us=2.0, deps=[0]),
            dict(name='d', stream=0, start_us=146.0, stop_us=150.0, host_us=3.0, deps=[1, 2])]
    out = io.StringIO()
    res = cp.ledger(recs, out, ms_per_step=0.05, title='t')
    assert res['launches'] == 4
    assert abs(res['chain_us'] - 50.0) < 1e-9                        # a -> c -> d
    assert abs(res['kernel_us'] - (10.0 + 29.0 + 4.0)) < 1e-9
    assert abs(res['gaps']['join'] - (1.0 + 6.0)) < 1e-9             # c behind a (other stream), d behind c (other stream)
    txt = out.getvalue()
    assert 'critical path: 3 launches' in txt and 'off the chain' in txt


# shapes chosen for the sample-local path's own edges (all Lq <= 32): a one-word query, a full 32-row window, Embedding widths whose last K = 16
# step is half / mostly empty (408 = 25.5 steps, 144 = 9 steps of the chunked linear), wide char embeddings, a trainable word table
QUERY_EDGE_SHAPES = [dict(name='query Lq=%d EW=%d cd=%d' % (Lq, wd + 100, cd), B=B, T=T, Lq=Lq, Lc=Lc, Dv=64, char_dim=cd, char_size=40, word_table=wt, word_dim=wd)
                     for (B, T, Lq, Lc, wd, cd, wt) in [(4, 40, 1, 6, 300, 50, False), (3, 33, 32, 8, 300, 50, False), (5, 24, 20, 10, 308, 50, False),
                                                        (2, 64, 7, 5, 44, 24, False), (3, 50, 13, 12, 300, 100, True), (6, 17, 31, 4, 300, 50, False)]]


@pytest.mark.gpu
@pytest.mark.parametrize('shape', QUERY_EDGE_SHAPES, ids=lambda s: s['name'].replace(' ', '_').replace('=', ''))
def test_sample_local_query_path_against_the_oracle(shape):
    from tests.test_hip_training import check_shape_against_oracle
    check_shape_against_oracle(shape, scaled_bias_floor=True)


@pytest.mark.gpu
def test_which_launches_a_step_takes_and_what_the_ledger_records():
    from vslnet_amd.dp import FlatAdamW, backward_exchange_step
    from vslnet_amd.model.VSLNet import VSLNet
    from vslnet_amd.synthetic import make_configs, synthetic_batch

    def launches(Lq):
        configs = make_configs(video_feature_dim=128, max_pos_len=64, drop_rate=0.2, predictor='transformer')
        torch.manual_seed(configs.seed)
        glove = torch.randn(configs.word_size - 2, configs.word_dim).numpy()
        model = VSLNet(configs, glove).cuda().train()
        flat, grads = model.flat_parameters
        eng = model._engine
        pad_vec, glove_vec = model.embedding_net.word_emb.pad_vec.data, model.embedding_net.word_emb.glove_vec.data
        bt = synthetic_batch(configs, 4, 64, Lq, 8, seed=3)
        opt = FlatAdamW(flat, eng.layout, lr=1e-4, num_train_steps=100, clip_norm=1.0, engine=eng)

        def step(i):
            eng.forward(flat, pad_vec, glove_vec, bt['word_ids'], bt['char_ids'], bt['vfeats'], bt['v_mask'], bt['q_mask'], training=True, seed=i)
            _, d_h, d_sl, d_el = eng.loss(bt['s_labels'], bt['e_labels'], bt['h_labels'], 1.0, 5.0, inv_batch=0.25, mask_sum=float(bt['v_mask'].sum()), lazy=True)
            backward_exchange_step(eng, None, grads, (d_h, d_sl, d_el), opt)        # (as main.train and bench.py)
        step(0)
        eng.profile_select('*')
        step(1)
        torch.cuda.synchronize()
        recs = eng.profile_launches()
        eng.profile_select(None)
        return recs

    recs = launches(20)
    names = [r['name'] for r in recs]
    assert names.count('query_fwd') == 1 and names.count('query_bwd') == 1 and 'linear_fwd' not in names
    assert names.count('convblock_fwd') == 3 and names.count('convblock_bwd') == 3
    assert 'loss' in names and names[-1] == 'adamw' and 'cq_col' not in names       # (the column kernel is folded into its neighbours at T <= 128, Lq <= 32)
    assert names.index('loss') > names.index('embed_bwd')                          # the lazy loss: behind the main stream's last kernel, not in front of head_bwd
    assert len(recs) <= 42, len(recs)
    assert len(set(r['stream'] for r in recs)) == 3
    for i, r in enumerate(recs):
        assert r['stop_us'] > r['start_us'] >= 0.0
        assert all(0 <= d < i for d in r['deps']), (i, r)
        for d in r['deps']:
            assert recs[d]['stop_us'] <= r['start_us'] + 0.5, 'launch %d (%s) started before its dependency %d (%s) stopped' % (i, r['name'], d, recs[d]['name'])
    assert any(len(r['deps']) >= 2 for r in recs)                 # the joins
    os.environ['VSL_FUSED_TAIL'] = '1'                          # the final reduction + clip + AdamW as one launch (off by default)
    try:
        fused = [r['name'] for r in launches(20)]
    finally:
        del os.environ['VSL_FUSED_TAIL']
    assert fused[-1] == 'reduce_adamw' and 'adamw' not in fused and len(fused) == len(names) - 1
    names40 = [r['name'] for r in launches(40)]
    assert 'query_fwd' not in names40 and 'query_bwd' not in names40 and names40.count('convblock_fwd') == 4
