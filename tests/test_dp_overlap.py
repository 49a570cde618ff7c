"""The data-parallel exchange path on the driver's own GPU (round 4).  The reference has no counterpart
(single `--gpu_idx`, main_t7.py:31,66-67), so nothing else would catch an early-gradient event that fires too soon: with N > 1 the
predictor block of the bucket would be all-reduced before it is final and training would be silently wrong.

* `Engine.backward(..., early_event=ev)`: a side stream waits for `ev` and snapshots `grads[split:]`; after the backward the snapshot
  must be bit-identical to the final block -- with the bucket poisoned before every backward (a stale but equal value cannot pass) and a
  spin kernel ahead on the main stream (the side stream reaches its wait long before the backward starts: the widest race window).
* `bench.py` under a one-rank RCCL process group (`VSL_FORCE_DIST=1 VSL_ALLREDUCE=overlap`): the exchange path end to end.
* `Engine()` refuses to run beside a process group with fewer than 8 hardware queues (CPU test, monkeypatched).
"""
import json
import os
import subprocess
import sys

import pytest
import torch

from oracle import vslnet_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POISON = 12345.678


def _setup(cfg, B, T, Lq, Lc, ragged):
    from vslnet_amd.engine import Engine, flat_from_state_dict
    P = O.random_params(cfg, seed=7)
    eng = Engine(cfg)
    flat = flat_from_state_dict(eng, P)
    d = {k: v.cuda().contiguous() for k, v in O.synthetic_batch(cfg, B=B, T=T, Lq=Lq, Lc=Lc, seed=9, ragged=ragged).items()}
    return eng, flat, P, d


SHAPES = [
    # name, cfg overrides, B, T, Lq, Lc, ragged
    ('headline', dict(video_feature_dim=1024, max_pos_len=128), 64, 128, 20, 10, False),
    ('ragged_small', dict(video_feature_dim=64, max_pos_len=64, word_size=52), 3, 37, 7, 6, True),
    ('T256', dict(video_feature_dim=256, max_pos_len=256, word_size=52), 8, 256, 12, 8, True),
    ('rnn', dict(video_feature_dim=64, max_pos_len=64, word_size=52, predictor='rnn'), 4, 48, 7, 6, True),
]


@pytest.mark.gpu
@pytest.mark.parametrize('name,over,B,T,Lq,Lc,ragged', SHAPES, ids=[s[0] for s in SHAPES])
def test_early_gradient_event_fires_after_the_predictor_block_is_final(name, over, B, T, Lq, Lc, ragged):
    cfg = O.make_cfg(drop_rate=0.2, **over)
    eng, flat, P, d = _setup(cfg, B, T, Lq, Lc, ragged)
    split = eng.early_grad_offset()
    n = eng.param_floats
    if cfg.predictor == 'rnn':
        assert split == n                     # no early block: the exchange is one call behind the whole backward
    else:
        names = [nm for nm, off, _, _ in eng.layout if off >= split]
        assert 0 < split < n and names and all(nm.startswith('predictor.') for nm in names)
        assert all(not nm.startswith('predictor.') for nm, off, _, _ in eng.layout if off < split)
    pad, glove = P['embedding_net.word_emb.pad_vec'].cuda(), P['embedding_net.word_emb.glove_vec'].cuda()
    side = torch.cuda.Stream()
    ev = torch.cuda.Event()
    main = torch.cuda.current_stream()
    grads = eng.new_flat()
    snap = torch.empty(max(n - split, 1), device='cuda')
    real = torch.zeros(n, dtype=torch.bool, device='cuda')      # every tensor starts on a 16-byte boundary: the pad floats between them are never written
    for _, off, numel, _ in eng.layout:
        real[off:off + numel] = True
    reference = None
    for rep in range(20):
        eng.forward(flat, pad, glove, d['word_ids'], d['char_ids'], d['vfeats'], d['v_mask'], d['q_mask'], training=True, seed=1000 + rep % 2)
        _, d_h, d_sl, d_el = eng.loss(d['s_labels'], d['e_labels'], d['h_labels'], 1.0, 5.0)
        grads.fill_(POISON)
        snap.fill_(-1.0)
        torch.cuda._sleep(2_000_000 + 500_000 * (rep % 3))       # ~1 ms of spinning ahead of the backward on the main stream
        eng.backward(d_h, d_sl, d_el, grads, early_event=ev)
        side.wait_event(ev)                                       # what dp.OverlappedExchange does
        with torch.cuda.stream(side):
            if split < n:
                snap.copy_(grads[split:], non_blocking=True)
        main.wait_stream(side)
        torch.cuda.synchronize()
        final = grads.clone()
        assert not bool((final[real] == POISON).any()), 'the backward left part of the bucket unwritten'
        assert bool(torch.isfinite(final[real]).all())
        if split < n:
            assert torch.equal(snap, final[split:]), '%s rep %d: grads[split:] changed after the early-gradient event fired' % (name, rep)
        if rep < 2:
            reference = final if rep == 0 else reference
        elif rep % 2 == 0:                                        # same seed as rep 0: the whole bucket is reproducible bit for bit
            assert torch.equal(final, reference)


@pytest.mark.gpu
def test_overlapped_exchange_equals_plain_backward_on_one_rank():
    """dp.OverlappedExchange.backward(skip_exchange=True) takes the event path without needing a process group: same bucket as the
    plain backward, bit for bit."""
    from vslnet_amd.dp import OverlappedExchange
    cfg = O.make_cfg(drop_rate=0.2, video_feature_dim=64, max_pos_len=64, word_size=52)
    eng, flat, P, d = _setup(cfg, 5, 40, 7, 6, True)
    pad, glove = P['embedding_net.word_emb.pad_vec'].cuda(), P['embedding_net.word_emb.glove_vec'].cuda()
    x = OverlappedExchange(eng)
    out = []
    for mode in range(2):
        eng.forward(flat, pad, glove, d['word_ids'], d['char_ids'], d['vfeats'], d['v_mask'], d['q_mask'], training=True, seed=5)
        _, d_h, d_sl, d_el = eng.loss(d['s_labels'], d['e_labels'], d['h_labels'], 1.0, 5.0)
        g = eng.new_flat().fill_(POISON)
        if mode:
            x.backward(d_h, d_sl, d_el, g, skip_exchange=True)
        else:
            eng.backward(d_h, d_sl, d_el, g)
        torch.cuda.synchronize()
        out.append(g.clone())
    assert torch.equal(out[0], out[1])


def _same_loss(over, plain):
    """The two runs differ in ONE place: the plain run clips by the norm the backward's reduction left behind (`norm_from_backward`), the run under a
    communicator by `k_sqsum` over the exchanged bucket -- the same sum of squares in another order, i.e. a clip factor that may differ in its last
    bit, and parameters that then differ by an ulp per step.  The printed loss (5 decimals) is therefore compared to 1e-5 relative + its print
    rounding, not for equality (equality had held until round 5 by rounding luck); a dispatch-order violation of the fused rnn head shows as NaN."""
    assert over == over and abs(over - plain) <= 1e-5 * abs(plain) + 1.5e-5, (over, plain)


def _bench(extra_env, *args):
    env = dict(os.environ)
    env.update(extra_env)
    steps = [] if '--steps' in args else ['--steps', '6']
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *steps, '--warmup', '2', '--regions', '1', '--no-shapes', '--no-cpu-baseline', *args],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_bench_under_a_one_rank_rccl_group_takes_the_overlapped_exchange():
    """`VSL_FORCE_DIST=1 VSL_ALLREDUCE=overlap python bench.py --gpus 1`: RCCL communicator of size 1, both all-reduce calls, the
    early-gradient event, the side stream -- and the same loss as the plain run (same seeds, same batches, identity all-reduce)."""
    plain = _bench({})
    dist_env = {'VSL_FORCE_DIST': '1', 'VSL_ALLREDUCE': 'overlap', 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29611', 'RANK': '0',
                'WORLD_SIZE': '1', 'LOCAL_RANK': '0', 'HSA_ENABLE_IPC_MODE_LEGACY': '0'}
    over = _bench(dist_env)
    assert plain['rccl_ranks'] == 0 and over['rccl_ranks'] == 1
    assert 'allreduce_us' in over and 'step_without_allreduce_ms' in over and over['allreduce'].startswith('two calls')
    _same_loss(over['config']['loss'], plain['config']['loss'])
    assert over['ms_per_step'] < 3.0 * plain['ms_per_step']
    print('[one-rank exchange] plain %.4f ms/step, under RCCL %.4f ms/step (exposed %.1f us)' % (plain['ms_per_step'], over['ms_per_step'], over['allreduce_us']))


@pytest.mark.gpu
def test_fused_rnn_head_at_its_largest_batch_beside_a_communicator():
    """The one-launch rnn head rests on workgroup dispatch order (producers are dispatched before their consumers; bounded spins turn a violation
    into NaNs).  B = 80 is its largest batch: 240 workgroups per direction, here 50 steps in a row with an RCCL communicator's kernels and
    streams in the same process (one rank: `VSL_FORCE_DIST=1`, overlapped two-call exchange path -- for the rnn head a single call).  The
    loss must be finite and identical to the plain run's (same seeds, identity all-reduce)."""
    args = ('--predictor', 'rnn', '--batch', '80', '--steps', '50')
    plain = _bench({}, *args)
    dist_env = {'VSL_FORCE_DIST': '1', 'VSL_ALLREDUCE': 'overlap', 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29613', 'RANK': '0',
                'WORLD_SIZE': '1', 'LOCAL_RANK': '0', 'HSA_ENABLE_IPC_MODE_LEGACY': '0'}
    over = _bench(dist_env, *args)
    assert over['rccl_ranks'] == 1 and over['steps'] == 50
    _same_loss(over['config']['loss'], plain['config']['loss'])


def test_engine_refuses_a_process_group_with_too_few_hardware_queues(monkeypatch):
    """Refused: a multi-rank RCCL ('nccl') group with fewer than 8 hardware queues (or the variable set after HIP came up).  A gloo group, one
    rank, or VSL_ALLOW_FEW_HW_QUEUES=1 only get a warning (the condition is a performance one, embedders must be able to opt out)."""
    import torch.distributed as dist
    import vslnet_amd
    from vslnet_amd import engine
    monkeypatch.setattr(dist, 'is_initialized', lambda: True)
    monkeypatch.setattr(dist, 'get_backend', lambda *a: 'nccl')
    monkeypatch.setattr(dist, 'get_world_size', lambda *a: 2)
    monkeypatch.delenv('VSL_ALLOW_FEW_HW_QUEUES', raising=False)
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', '4')
    with pytest.raises(engine.VslError, match='GPU_MAX_HW_QUEUES'):
        engine.check_hw_queues()
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', '8')
    monkeypatch.setattr(vslnet_amd, 'QUEUES_SET_LATE', False)
    engine.check_hw_queues()                                      # fine
    monkeypatch.setattr(vslnet_amd, 'QUEUES_SET_LATE', True)      # the default arrived after HIP was initialised
    with pytest.raises(engine.VslError, match='set after HIP was initialised'):
        engine.check_hw_queues()
    # the same state, but nothing to lose: one rank / gloo / explicit opt-out -> a warning, once
    for patch in (lambda: monkeypatch.setattr(dist, 'get_world_size', lambda *a: 1), lambda: monkeypatch.setattr(dist, 'get_backend', lambda *a: 'gloo'),
                  lambda: monkeypatch.setenv('VSL_ALLOW_FEW_HW_QUEUES', '1')):
        monkeypatch.setattr(dist, 'get_world_size', lambda *a: 2)
        monkeypatch.setattr(dist, 'get_backend', lambda *a: 'nccl')
        monkeypatch.delenv('VSL_ALLOW_FEW_HW_QUEUES', raising=False)
        patch()
        monkeypatch.setattr(engine, '_HW_QUEUE_WARNED', False)
        with pytest.warns(UserWarning, match='GPU_MAX_HW_QUEUES'):
            engine.check_hw_queues()
        engine.check_hw_queues()                                  # second call: silent
    monkeypatch.setattr(dist, 'is_initialized', lambda: False)    # no process group: nothing to check
    engine.check_hw_queues()
