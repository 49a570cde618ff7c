"""Fixed cost of one torch.distributed all_reduce between two kernels of the compute stream (single rank is enough to see
the stream hand-offs): MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python tools/allreduce_cost.py"""
import sys
import time

import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
for n in (16, 700000):
    x = torch.zeros(n, device='cuda')
    y = torch.zeros(1 << 20, device='cuda')
    for _ in range(10):
        dist.all_reduce(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        y.add_(1.0); dist.all_reduce(x); y.add_(1.0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(200):
        y.add_(1.0); y.add_(1.0)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%d floats: all_reduce adds %.1f us per call' % (n, ((t1 - t0) - (t2 - t1)) / 200 * 1e6), file=sys.stderr)
dist.destroy_process_group()
