"""What the gradient exchange should cost when an 8-GPU node first runs it -- a prediction to check SCALE_rNN.json's `allreduce_us` against.

Measured here (one rank is enough to see the launch and stream hand-off cost RCCL adds between two kernels of the compute stream):
    MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python tools/allreduce_cost.py [out.json]
for the two bucket sizes of the path (2.7 MB: transformer head at Dv = 1024; 4.4 MB: Dv = 4096 -- SURVEY 8e) and for the two-call form of
dp.OverlappedExchange (predictor block + rest).  Modelled on top (no multi-GPU box in this pool): a ring all-reduce over xGMI moves
2 (N - 1) / N of the bucket per GPU over ONE link per direction (MI355X_MICROARCH.md: 7 links x ~153 GB/s, point to point) in 2 (N - 1)
steps; RING_EFF of the link rate is assumed reachable at these sizes, ALPHA_US per step.  The exposed cost per training step is what the
backward does not hide: the predictor block (~55 % into the backward) overlaps, the rest of the bucket follows the backward."""
import json
import sys
import time

import torch
import torch.distributed as dist

LINK_GBPS, RING_EFF, ALPHA_US = 153.0, 0.6, 3.0


def measure(n_floats, calls):
    xs = [torch.zeros(n, device='cuda') for n in n_floats]
    y = torch.zeros(1 << 20, device='cuda')
    for _ in range(10):
        for x in xs:
            dist.all_reduce(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        y.add_(1.0)
        for x in xs:
            dist.all_reduce(x)
        y.add_(1.0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(calls):
        y.add_(1.0)
        y.add_(1.0)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return ((t1 - t0) - (t2 - t1)) / calls * 1e6


def ring_us(nbytes, N):
    return 0.0 if N < 2 else 2.0 * (N - 1) / N * nbytes / (LINK_GBPS * RING_EFF * 1e3) + 2 * (N - 1) * ALPHA_US


def main():
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    out = {'model': {'link_GBps': LINK_GBPS, 'ring_efficiency': RING_EFF, 'alpha_us_per_ring_step': ALPHA_US,
                     'formula': 'fixed(N=1, measured) + 2 (N-1)/N * bytes / (link * eff) + 2 (N-1) * alpha'}, 'buckets': []}
    measure([4096], 100)                                            # communicator warm-up (the first timed loop otherwise pays lazy initialisation)
    for name, n, early in (('transformer head, Dv=1024 (configs[1], [3], [4])', 675_000, 0.44), ('transformer head, Dv=4096 (configs[2])', 1_100_000, 0.27)):
        one = measure([n], 200)
        two = measure([int(n * early), n - int(n * early)], 200)
        row = {'bucket': name, 'floats': n, 'MB': round(n * 4 / 1e6, 2), 'fixed_us_one_call_one_rank': round(one, 1),
               'fixed_us_two_calls_one_rank': round(two, 1), 'early_block_fraction': early, 'predicted': {}}
        for N in (2, 4, 8):
            total = one + ring_us(n * 4, N)
            late = two / 2 + ring_us(n * 4 * (1 - early), N)          # the call behind the backward; the early block hides behind ~45 % of it
            row['predicted'][str(N)] = {'allreduce_us_single_call': round(total, 1), 'allreduce_us_overlapped_exposed': round(late, 1)}
        out['buckets'].append(row)
        print(row, file=sys.stderr)
    dist.destroy_process_group()
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 1:
        open(sys.argv[1], 'w').write(txt + '\n')
    print(txt)


if __name__ == '__main__':
    main()
