"""cpu_baseline calibration: the oracle restatement ("port") timed beside the IMPORTED reference on the same cores, same
workload, in the BUILD container (the reference cannot travel to the GPU box).  Writes profiles/r04_cpu_calibration.json, which bench.py
attaches to its `cpu_baseline` object as `calibration`.

    python tools/calibrate_cpu_baseline.py            # needs /root/reference; B=16 T=128 Dv=1024 Lq=20 Lc=10 drop 0.2, 8 threads
"""
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vslnet_oracle as O  # noqa: E402
from oracle.make_golden import load_reference  # noqa: E402

THREADS, B, T, DV, LQ, LC, DROP = 8, 16, 128, 1024, 20, 10, 0.2


def timed(step, warm=3, n=10):
    for _ in range(warm):
        step()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


def main():
    torch.set_num_threads(THREADS)
    VSLNet, _, _ = load_reference()
    cfg = O.make_cfg(video_feature_dim=DV, max_pos_len=T, drop_rate=DROP)
    b = O.synthetic_batch(cfg, B, T, LQ, LC, seed=0)
    sd = O.random_params(cfg, seed=1)
    model = VSLNet(configs=cfg, word_vectors=np.zeros((cfg.word_size - 2, cfg.word_dim), np.float32))
    model.load_state_dict(sd, strict=True)
    model.train()

    def ref_step():                                   # the timed region of main_t7.py:103-110
        model.zero_grad()
        h, sl, el = model(b['word_ids'], b['char_ids'], b['vfeats'], b['v_mask'], b['q_mask'])
        total = model.compute_loss(sl, el, b['s_labels'], b['e_labels']) + 5.0 * model.compute_highlight_loss(h, b['h_labels'], b['v_mask'])
        total.backward()
    P = {k: v.clone().requires_grad_(k not in O.FROZEN) for k, v in sd.items()}

    def oracle_step():                                # bench.py cpu_baseline()
        for p in P.values():
            p.grad = None
        total, _ = O.total_loss(P, cfg, b, training=True)
        total.backward()
    # interleaved: ref, oracle, ref, oracle
    r1, o1, r2, o2 = timed(ref_step), timed(oracle_step), timed(ref_step), timed(oracle_step)
    ref, orc = min(r1, r2), min(o1, o2)
    out = {'reference_ms_per_step': round(ref * 1e3, 1), 'oracle_ms_per_step': round(orc * 1e3, 1), 'ratio_reference_over_oracle': round(ref / orc, 3),
           'threads': THREADS, 'workload': 'B=%d T=%d Dv=%d Lq=%d Lc=%d drop_rate=%.1f transformer, train mode, fwd + both losses + bwd' % (B, T, DV, LQ, LC, DROP),
           'where': 'build container (the reference is imported from /root/reference; it does not exist on the GPU box)',
           'script': 'tools/calibrate_cpu_baseline.py', 'runs_ms': [round(x * 1e3, 1) for x in (r1, o1, r2, o2)]}
    path = os.path.join(ROOT, 'profiles', 'r04_cpu_calibration.json')
    json.dump(out, open(path, 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
