"""The critical-path ledger of ONE headline training step, from HIP events -- no profiler attached.

`vsl_profile_select("*")` puts a timing start / stop event on every launch's own dispatch packet and records, per launch, its stream, the
host's enqueue time and the launches it was ordered behind (same-stream predecessor + every cross-stream ordering point).  This tool
runs the bench loop, profiles one step in the middle of a back-to-back run (so the host is as far ahead as it is in `bench.py`), and
prints:

  * every launch: stream, start / stop offset, duration, the dependency that released it (the one that stopped last) and the gap to it;
  * the longest dependency chain ending in the step's last kernel: kernel time on the chain, same-stream gaps, cross-stream join gaps,
    and gaps where the HOST enqueued the launch after its dependencies had already finished (host-late).

    python tools/critical_path.py [--steps-before 30] [--out profiles/r06_critical_path.txt] [bench shape flags]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def ledger(recs, out, ms_per_step=None, title=''):
    n = len(recs)
    t0 = min(r['start_us'] for r in recs)
    for r in recs:
        r['start_us'] -= t0
        r['stop_us'] -= t0
    # host clock -> device clock: the host enqueues a launch before it starts; the tightest pair gives the offset (launch latency included)
    off = min(r['start_us'] - r['host_us'] for r in recs)
    for i, r in enumerate(recs):
        deps = [d for d in r['deps'] if 0 <= d < n]
        r['rel'] = max(deps, key=lambda d: recs[d]['stop_us']) if deps else -1       # the dependency that released it
        r['ready_us'] = recs[r['rel']]['stop_us'] if deps else 0.0
        r['gap_us'] = r['start_us'] - r['ready_us']
        r['host_dev_us'] = r['host_us'] + off                                          # earliest moment the device could have started it
        r['kind'] = 'first' if not deps else ('same' if recs[r['rel']]['stream'] == r['stream'] else 'join')
        if deps and r['host_dev_us'] > r['ready_us'] + 0.5:
            r['kind'] = 'host'
    span = max(r['stop_us'] for r in recs)
    out.write('# %s\n' % title)
    out.write('# one training step, %d launches on %d streams, first start -> last stop %.1f us' % (n, len(set(r['stream'] for r in recs)), span))
    if ms_per_step:
        out.write(' ; the same loop un-instrumented: %.1f us per step' % (1e3 * ms_per_step))
    out.write('\n# idx  stream   start    stop     dur   released by (idx: name)            gap  kind   host-enqueue (device clock)\n')
    for i, r in enumerate(recs):
        rel = '%3d: %-18s' % (r['rel'], recs[r['rel']]['name'][:18]) if r['rel'] >= 0 else '  -                    '
        out.write('%4d  s%d  %8.1f %8.1f %6.1f   %s %6.1f  %-5s %8.1f   %s\n' % (i, r['stream'], r['start_us'], r['stop_us'], r['stop_us'] - r['start_us'], rel,
                                                                              r['gap_us'], r['kind'], r['host_dev_us'], r['name']))
    # the chain that ends in the last kernel to stop
    last = max(range(n), key=lambda i: recs[i]['stop_us'])
    chain = []
    i = last
    while i >= 0:
        chain.append(i)
        i = recs[i]['rel']
    chain.reverse()
    kt = sum(recs[i]['stop_us'] - recs[i]['start_us'] for i in chain)
    gaps = {'same': 0.0, 'join': 0.0, 'host': 0.0, 'first': 0.0}
    cnt = {'same': 0, 'join': 0, 'host': 0, 'first': 0}
    for i in chain:
        gaps[recs[i]['kind']] += recs[i]['gap_us']
        cnt[recs[i]['kind']] += 1
    out.write('\n# critical path: %d launches, %.1f us = %.1f kernel time + %.1f same-stream boundaries (%d) + %.1f cross-stream joins (%d) + %.1f host-late (%d)\n'
              % (len(chain), recs[last]['stop_us'] - recs[chain[0]]['start_us'], kt, gaps['same'], cnt['same'], gaps['join'], cnt['join'], gaps['host'], cnt['host']))
    out.write('# idx  stream   start    stop     dur     gap  kind   name\n')
    for i in chain:
        r = recs[i]
        out.write('%4d  s%d  %8.1f %8.1f %6.1f  %6.1f  %-5s  %s\n' % (i, r['stream'], r['start_us'], r['stop_us'], r['stop_us'] - r['start_us'], r['gap_us'], r['kind'], r['name']))
    # kernels off the chain: their slack (how much later they could have stopped without moving anything that waited for them)
    on = set(chain)
    users = {}
    for i, r in enumerate(recs):
        for d in r['deps']:
            if 0 <= d < n:
                users.setdefault(d, []).append(i)
    out.write('\n# off the chain (slack = earliest start of a launch ordered behind it - its stop)\n')
    for i, r in enumerate(recs):
        if i in on:
            continue
        u = users.get(i, [])
        slack = min(recs[j]['start_us'] for j in u) - r['stop_us'] if u else float('nan')
        out.write('%4d  s%d  %8.1f %8.1f %6.1f  slack %6.1f  %s\n' % (i, r['stream'], r['start_us'], r['stop_us'], r['stop_us'] - r['start_us'], slack, r['name']))
    return dict(chain_us=recs[last]['stop_us'] - recs[chain[0]]['start_us'], kernel_us=kt, gaps=gaps, launches=n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--T', type=int, default=128)
    ap.add_argument('--dv', type=int, default=1024)
    ap.add_argument('--lq', type=int, default=20)
    ap.add_argument('--lc', type=int, default=10)
    ap.add_argument('--predictor', default='transformer')
    ap.add_argument('--steps-before', type=int, default=30)
    ap.add_argument('--steps-after', type=int, default=10)
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    import torch
    from vslnet_amd.model.VSLNet import VSLNet
    from vslnet_amd.synthetic import make_configs, synthetic_batch
    from vslnet_amd.dp import FlatAdamW, backward_exchange_step
    B, T, Dv, Lq, Lc = args.batch, args.T, args.dv, args.lq, args.lc
    configs = make_configs(video_feature_dim=Dv, max_pos_len=max(T, Lq), drop_rate=0.2, predictor=args.predictor)
    torch.manual_seed(configs.seed)
    glove = torch.randn(configs.word_size - 2, configs.word_dim).numpy()
    model = VSLNet(configs, glove).cuda().train()
    flat, grads = model.flat_parameters
    eng = model._engine
    pad_vec, glove_vec = model.embedding_net.word_emb.pad_vec.data, model.embedding_net.word_emb.glove_vec.data
    batches = [synthetic_batch(configs, B, T, Lq, Lc, seed=100 + 1000 * k) for k in range(10)]
    mask_sum = float(batches[0]['v_mask'].sum().item())
    opt = FlatAdamW(flat, eng.layout, lr=configs.init_lr, num_train_steps=100000, clip_norm=configs.clip_norm, engine=eng)

    def step(i):
        b = batches[i % len(batches)]
        eng.forward(flat, pad_vec, glove_vec, b['word_ids'], b['char_ids'], b['vfeats'], b['v_mask'], b['q_mask'], training=True, seed=i)
        _, d_h, d_sl, d_el = eng.loss(b['s_labels'], b['e_labels'], b['h_labels'], 1.0, configs.highlight_lambda, inv_batch=1.0 / B, mask_sum=mask_sum, lazy=True)
        backward_exchange_step(eng, None, grads, (d_h, d_sl, d_el), opt)

    for i in range(10):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(40):
        step(10 + i)
    torch.cuda.synchronize()
    plain = (time.perf_counter() - t0) / 40 * 1e3
    # the profiled step sits inside a back-to-back run: the host is as far ahead of the device as it is in bench.py's timed region
    for i in range(args.steps_before):
        step(100 + i)
    eng.profile_select('*')
    step(200)
    # the following steps are recorded too (the queue stays as busy behind the profiled step as in front of it) and cut off below
    for i in range(args.steps_after):
        step(201 + i)
    torch.cuda.synchronize()
    recs = eng.profile_launches()
    eng.profile_select(None)
    per = len(recs) // (1 + args.steps_after)
    one = recs[:per]
    nxt = recs[per]['start_us'] - recs[0]['start_us'] if len(recs) > per else float('nan')
    title = 'critical-path ledger, HIP events on the dispatch packets (tools/critical_path.py): B=%d T=%d Dv=%d Lq=%d %s, drop 0.2, optimizer inside' % (B, T, Dv, Lq, args.predictor)
    out = open(args.out, 'w') if args.out else sys.stdout
    res = ledger(one, out, ms_per_step=plain, title=title)
    out.write('\n# step period with the events attached (first launch of this step -> first launch of the next): %.1f us ; un-instrumented loop: %.1f us per step\n' % (nxt, 1e3 * plain))
    if args.out:
        out.close()
        print(open(args.out).read())
    print('chain %.1f us, kernel %.1f, gaps %s, %d launches, plain %.1f us' % (res['chain_us'], res['kernel_us'], res['gaps'], res['launches'], 1e3 * plain), file=sys.stderr)


if __name__ == '__main__':
    main()
