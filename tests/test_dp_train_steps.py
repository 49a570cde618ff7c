"""main.train's per-step exchange on more than one rank, on CPU (gloo): the unit main.py calls every step -- dp.backward_and_exchange, then the
replicated FlatAdamW update -- driven over two steps by 2 and 3 ranks, with an uneven batch (B = 7) and a LAST batch that has fewer samples than
there are ranks (the empty-shard branch), for the transformer head (two calls: predictor block first) and the rnn head (no early block: one call),
through `OverlappedExchange` and through VSL_ALLREDUCE=single's one call.  The compute engine is a stand-in that produces the flat gradient bucket
with the CPU oracle (test infrastructure); call order, sizes, zero buckets, normalisers and the update are the product's (vslnet_amd/dp.py).
No reference counterpart: the reference is single-device (main_t7.py:31,66-67)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from oracle import vslnet_oracle as O
from vslnet_amd import dp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleEngine:
    """What dp.py needs of an engine, on the CPU: `device`, `early_grad_offset()`, `backward(d_h, d_sl, d_el, grads)`.  The gradient of the
    rank's shard (global normalisers, like vsl_loss) comes from the oracle's autograd; `layout` = (name, offset, numel, shape) of the bucket."""
    device = torch.device('cpu')

    def __init__(self, cfg, predictor):
        self.cfg = cfg
        P = O.random_params(cfg, seed=5)
        self.names = [k for k in P if k not in O.FROZEN]
        self.frozen = {k: v for k, v in P.items() if k in O.FROZEN}
        self.layout, off = [], 0
        for k in self.names:
            self.layout.append((k, off, P[k].numel(), tuple(P[k].shape)))
            off += P[k].numel()
        self.numel = off
        self.flat = torch.cat([P[k].reshape(-1) for k in self.names]).clone()
        first = next((o for k, o, _, _ in self.layout if k.startswith('predictor.')), self.numel)
        self.split = first if predictor == 'transformer' else self.numel        # rnn head: no early block (Engine.early_grad_offset)
        self.batch = self.norm = None

    def early_grad_offset(self):
        return self.split

    def params(self):
        P = dict(self.frozen)
        for k, o, n, shp in self.layout:
            P[k] = self.flat[o:o + n].view(shp).clone().requires_grad_(True)
        return P

    def forward_loss(self, batch, inv_batch, mask_sum):
        self.batch, self.norm = batch, (inv_batch, mask_sum)
        return torch.zeros(4), None, None, None

    def backward(self, d_h, d_sl, d_el, grads, early_event=None):
        P, b, (inv_b, msum) = self.params(), self.batch, self.norm
        h, sl, el = O.forward(P, self.cfg, b['word_ids'], b['char_ids'], b['vfeats'], b['v_mask'], b['q_mask'])
        ce = torch.nn.functional.cross_entropy(sl, b['s_labels'], reduction='sum') + torch.nn.functional.cross_entropy(el, b['e_labels'], reduction='sum')
        y = b['h_labels'].float()
        w = torch.where(y == 0.0, y + 1.0, 2.0 * y)
        hl = (torch.nn.functional.binary_cross_entropy(h, y, reduction='none') * w * b['v_mask']).sum() / (msum + 1e-12)
        (ce * inv_b + 5.0 * hl).backward()
        grads.copy_(torch.cat([(P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])).reshape(-1) for k in self.names]))
        return grads


BATCHES = [(7, 9), (2, 10)]          # (samples, seed): 7 = uneven shards on 2 and 3 ranks; 2 < 3 ranks = a rank without rows


def _train(rank, world, predictor, mode):
    """two steps of main.train's fused path; returns the final flat parameters"""
    torch.set_num_threads(2)
    cfg = O.make_cfg(video_feature_dim=32, max_pos_len=32, word_size=52, predictor=predictor)
    eng = OracleEngine(cfg, predictor)
    grads = torch.zeros(eng.numel)
    opt = dp.FlatAdamW(eng.flat, eng.layout, lr=1e-3, num_train_steps=10, clip_norm=1.0)
    xchg = dp.OverlappedExchange(eng) if world > 1 and mode == 'overlap' else None
    empty_seen = 0
    for B, seed in BATCHES:
        full = O.synthetic_batch(cfg, B=B, T=20, Lq=6, Lc=5, seed=seed, ragged=True)
        inv_batch, mask_sum = dp.global_normalisers(full['lens'].tolist())
        shard = dp.shard_batch(full, rank, world)
        if shard['vfeats'].shape[0] == 0:
            seeds, empty_seen = None, empty_seen + 1
        else:
            _, *seeds = eng.forward_loss(shard, inv_batch, mask_sum)
        touched = dp.backward_and_exchange(eng, xchg, grads, seeds)
        assert touched == (world > 1), (touched, world)
        opt.step(grads, from_backward=not touched)
    return eng.flat.clone(), empty_seen, eng.split, eng.numel


def _worker(rank, world, port, predictor, mode, out):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    flat, empty_seen, split, numel = _train(rank, world, predictor, mode)
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    empties = torch.tensor([empty_seen])
    dist.all_reduce(empties)
    if rank == 0:
        out.put((flat.numpy().copy(), all(torch.equal(flat, g) for g in gathered), int(empties.item()), split, numel))   # (numpy: no shared-memory handles)
    dist.destroy_process_group()


def _run(world, predictor, mode):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, predictor, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    res = (torch.from_numpy(res[0]),) + tuple(res[1:])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize('predictor', ['transformer', 'rnn'])
def test_three_ranks_uneven_batches_train_like_one_process(predictor):
    ref, _, split, numel = _train(0, 1, predictor, 'single')
    assert (split < numel) == (predictor == 'transformer')            # rnn: the single-call branch of OverlappedExchange / exchange()
    for mode in ('overlap', 'single'):
        flat, replicas_equal, empties, _, _ = _run(3, predictor, mode)
        assert replicas_equal, 'every rank must hold the same parameters after the replicated update'
        assert empties == 1, 'the second batch (2 samples, 3 ranks) must leave exactly one rank without rows'
        err = float((flat - ref).abs().max())
        assert err <= 2e-6 * float(ref.abs().max()) + 1e-7, (mode, err)  # (sums over ranks in another order than over samples: rounding only)


def test_two_ranks_overlapped_exchange_gives_the_bits_of_the_single_call():
    """two ranks: a + b has one order, so the two-call exchange (predictor block first) and the one-call exchange must agree BIT FOR BIT in the
    parameters after two updates (nothing ran main.train's default exchange on more than one rank)."""
    a = _run(2, 'transformer', 'overlap')
    b = _run(2, 'transformer', 'single')
    assert a[1] and b[1]
    assert torch.equal(a[0], b[0])
