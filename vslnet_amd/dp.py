"""Data-parallel training of the VSLNet path: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" for the CPU tests).  The reference has no distributed code at all (single `--gpu_idx`,
main_t7.py:31,66-67), so there is no call pattern to mirror; this is the scheme SURVEY.md 8(e) derives:

  * the global minibatch is split contiguously, B / N samples per rank; every shard is padded to the GLOBAL max T / Lq
    (the logits depend on the padded length -- SURVEY section 5 "pad-sensitivity");
  * losses use the GLOBAL normalisers (1 / B_global for the two CrossEntropy means, sum(v_mask) over the whole batch for
    the highlight loss) -- both known on the host before the step -- so per-rank gradients are plain partial sums;
  * ONE exchange (sum) of the flat fp32 gradient bucket per step (0.5 - 1.1 M floats = 2 - 4.4 MB: latency-bound), issued as TWO
    all-reduce calls: the predictor block (final about 55 % into the backward; the library fires an event, vsl_io.early_grads_event)
    goes out on a side stream while the rest of the backward runs, the remainder follows the backward (`OverlappedExchange`);
  * identical clip-by-global-norm + AdamW on every rank (replicated 2.7 MB of weights).
"""
import math

import torch


def shard_slice(global_batch, rank, world):
    """Contiguous split (sizes differ by at most one when B % N != 0)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return slice(lo, lo + base + (1 if rank < rem else 0))


def shard_batch(batch, rank, world):
    """Slice every per-sample tensor of a collated batch (dict of tensors with leading dim B).  The padded widths are left
    untouched: the shard keeps the global max T / Lq / Lc."""
    B = next(iter(batch.values())).shape[0]
    s = shard_slice(B, rank, world)
    return {k: (v[s].contiguous() if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B else v) for k, v in batch.items()}


def global_normalisers(lens_global):
    """(1 / B_global, sum of the global video mask) from the host-side clip counts of the WHOLE batch
    (vfeat_lens of train_collate_fn, data_loader_t7.py:33-34) -- no collective needed."""
    return 1.0 / float(len(lens_global)), float(sum(int(x) for x in lens_global))


def allreduce_flat_(flat_grads, group=None):
    """The one exchange step: sum the flat gradient bucket over all ranks, in place."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    return flat_grads


def two_segment_allreduce_(flat_grads, split, group=None):
    """The exchange of `OverlappedExchange` without streams or events (CPU / gloo tests): two all-reduce calls, [split:] first."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if split < flat_grads.numel():
            dist.all_reduce(flat_grads[split:], op=dist.ReduceOp.SUM, group=group)
        if split > 0:
            dist.all_reduce(flat_grads[:split], op=dist.ReduceOp.SUM, group=group)
    return flat_grads


class OverlappedExchange:
    """backward + the gradient exchange of one step.  grads[split:] (the predictor block) is all-reduced on a side stream as soon as
    the library reports it final, grads[:split] on the caller's stream behind the whole backward; the caller's stream then waits for
    the side stream.  Both calls go to the same communicator in the same order on every rank."""

    def __init__(self, engine, group=None):
        self.engine, self.group = engine, group
        self.split = engine.early_grad_offset()
        # (a CPU engine -- the gloo tests' stand-in -- has no streams: the same two calls, one after the other)
        self.on_gpu = torch.device(engine.device).type == 'cuda'
        self.side = torch.cuda.Stream(device=engine.device) if self.on_gpu else None
        self.event = torch.cuda.Event(enable_timing=False) if self.on_gpu else None

    def backward(self, d_h, d_sl, d_el, grads, skip_exchange=False):
        import torch.distributed as dist
        if not self.on_gpu:
            self.engine.backward(d_h, d_sl, d_el, grads)
            return grads if skip_exchange else two_segment_allreduce_(grads, self.split, self.group)
        cur = torch.cuda.current_stream(self.engine.device)
        overlapped = not skip_exchange and self.split < grads.numel()
        self.engine.backward(d_h, d_sl, d_el, grads, early_event=self.event if overlapped else None)
        if skip_exchange:
            return grads
        if overlapped:
            self.side.wait_event(self.event)
            with torch.cuda.stream(self.side):
                dist.all_reduce(grads[self.split:], op=dist.ReduceOp.SUM, group=self.group)
            if self.split > 0:
                dist.all_reduce(grads[:self.split], op=dist.ReduceOp.SUM, group=self.group)
            cur.wait_stream(self.side)
        else:                                             # rnn head: no early block -- [split:] is empty, ONE call (two_segment_allreduce_ does the same)
            dist.all_reduce(grads, op=dist.ReduceOp.SUM, group=self.group)
        return grads


    def exchange(self, grads):
        """The exchange alone, for a rank whose shard of the batch is empty (it has no backward to overlap with): the same two calls in the
        same order on the same communicator as `backward` issues on the other ranks."""
        two_segment_allreduce_(grads, self.split, self.group)
        return grads


def backward_and_exchange(engine, xchg, grads, seeds):
    """One training step's backward + gradient exchange, exactly as `main.train` runs it on every rank (the unit the multi-rank tests drive).
      seeds = (d_h, d_sl, d_el) from `engine.loss`, or None for a rank whose shard of this batch is EMPTY (the last batch of an epoch can hold
              fewer samples than there are ranks -- TACoS: 10146 % 16 = 2): such a rank contributes a zero bucket and joins the exchange with the
              same calls, in the same order, on the same communicator as the ranks that have rows;
      xchg  = an `OverlappedExchange` (two calls, the predictor block early) or None (ONE call behind the backward; VSL_ALLREDUCE=single).
    Returns True when the bucket was MODIFIED after the backward wrote it (an exchange ran): the optimizer must then take its own norm of the
    bucket (`FlatAdamW.step(from_backward=False)`) -- the backward's sum of squares belongs to the local gradient, not to the summed one."""
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(xchg.group if xchg is not None else None) > 1
    if seeds is None:
        grads.zero_()
        if xchg is not None:
            xchg.exchange(grads)
        else:
            allreduce_flat_(grads)
        return True
    if xchg is not None:
        xchg.backward(seeds[0], seeds[1], seeds[2], grads)
        return True
    engine.backward(seeds[0], seeds[1], seeds[2], grads)
    allreduce_flat_(grads)
    return multi


def backward_exchange_step(engine, xchg, grads, seeds, opt):
    """One training step from the loss seeds on: backward, gradient exchange, clip + AdamW -- what `main.train` and `bench.py` run.
    One process with rows in the batch: the update rides in the backward's last launch (`Engine.backward(fused_step=...)`: final reduction,
    global norm and AdamW as one kernel).  More than one rank: `backward_and_exchange`, then the two-kernel step on the SUMMED bucket."""
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(xchg.group if xchg is not None else None) > 1
    if not multi and xchg is None and seeds is not None and opt.engine is not None and getattr(engine, 'fused_tail', False):
        engine.backward(seeds[0], seeds[1], seeds[2], grads, fused_step=opt.fused_step())
        return
    touched = backward_and_exchange(engine, xchg, grads, seeds)
    opt.step(grads, from_backward=not touched)                # the backward's own norm only for an untouched bucket


class FlatAdamW:
    """clip_grad_norm_(1.0) + AdamW + linear decay (main_t7.py:111-113, VSLNet_t7.py:8-17) on the FLAT buckets.
    Weight decay 0.01 except for names containing bias / layer_norm / LayerNorm.  With an `engine` (GPU) the step is the
    library's fused two-kernel `vsl_adamw_step` (no host synchronisation: the global norm never leaves the device);
    without one (CPU / gloo tests) the same arithmetic runs as torch ops, which is also what the GPU test checks the
    kernels against.  Update rule = torch.optim.AdamW's (eps 1e-6); `hf_order=True` reproduces the historical
    transformers.AdamW the reference imports (VSLNet_t7.py:5): eps outside the bias correction, decoupled decay applied after
    the update."""

    def __init__(self, flat, layout, lr, num_train_steps, warmup_proportion=0.0, clip_norm=1.0, betas=(0.9, 0.999), eps=1e-6,
                 weight_decay=0.01, engine=None, hf_order=False):
        self.hf_order = bool(hf_order)
        self.flat, self.lr0, self.N, self.clip = flat, lr, float(num_train_steps), clip_norm
        self.warm = float(num_train_steps) * warmup_proportion
        self.b1, self.b2, self.eps, self.weight_decay = betas[0], betas[1], eps, weight_decay
        self.m, self.v = torch.zeros_like(flat), torch.zeros_like(flat)
        self.engine = engine
        self.wd = None
        if engine is None:
            self.wd = torch.zeros_like(flat)
            for name, off, numel, _ in layout:
                if not any(k in name for k in ('bias', 'layer_norm', 'LayerNorm')):
                    self.wd[off:off + numel] = weight_decay
        self.t = 0

    def lr(self):
        n = self.t
        if n < self.warm:
            return self.lr0 * n / max(1.0, self.warm)
        return self.lr0 * max(0.0, (self.N - n) / max(1.0, self.N - self.warm))

    def fused_step(self, grad_norm_out=None):
        """This optimizer step as the `fused_step` argument of `Engine.backward` (one process, no exchange): the update is applied by the
        backward's last launch, so there is no `step()` call for it.  Advances the step count like `step()` does."""
        if self.engine is None:
            raise RuntimeError('fused_step needs the GPU engine')
        lr = self.lr()
        self.t += 1
        return dict(flat=self.flat, exp_avg=self.m, exp_avg_sq=self.v, lr=lr, step=self.t, betas=(self.b1, self.b2), eps=self.eps,
                    weight_decay=self.weight_decay, clip_norm=self.clip, hf_order=self.hf_order, grad_norm_out=grad_norm_out)

    @torch.no_grad()
    def step(self, grads, from_backward=False):
        """`from_backward=True`: `grads` is exactly what the engine's last backward() wrote (one process, no exchange, nothing applied to
        it since): the clip's norm then comes out of the backward's own final reduction (Engine.adamw_step)."""
        lr = self.lr()
        self.t += 1
        if self.engine is not None:
            self.engine.adamw_step(self.flat, grads, self.m, self.v, lr, self.t, (self.b1, self.b2), self.eps, self.weight_decay,
                                   self.clip, hf_order=self.hf_order, norm_from_backward=from_backward)
            return
        if self.clip:
            gn = torch.linalg.vector_norm(grads)
            grads = grads * torch.clamp(self.clip / (gn + 1e-6), max=1.0)
        if not self.hf_order:
            self.flat.mul_(1 - lr * self.wd)
        self.m.mul_(self.b1).add_(grads, alpha=1 - self.b1)
        self.v.mul_(self.b2).addcmul_(grads, grads, value=1 - self.b2)
        bc1, bc2 = 1 - self.b1 ** self.t, 1 - self.b2 ** self.t
        if self.hf_order:
            self.flat.addcdiv_(self.m, self.v.sqrt().add_(self.eps), value=-lr * math.sqrt(bc2) / bc1)
            self.flat.sub_(self.flat * self.wd, alpha=lr)
            return
        denom = (self.v.sqrt() / math.sqrt(bc2)).add_(self.eps)
        self.flat.addcdiv_(self.m, denom, value=-lr / bc1)
