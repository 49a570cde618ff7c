"""Random shapes with Lq <= 32 (the sample-local query kernels' domain) against the oracle: forward logits, losses, every gradient.
usage: python tools/dbg/r06_fuzz_query.py [n] [seed]"""
import sys
import traceback

import numpy as np

sys.path.insert(0, '.')
from tests.test_hip_training import check_shape_against_oracle  # noqa: E402


def main(n=40, seed=606):
    rs = np.random.RandomState(seed)
    bad = 0
    for i in range(n):
        wd = int(rs.choice([300, 300, 44, 52, 156, 308, 412]))
        shape = dict(name='qfuzz %d' % i, B=int(rs.randint(1, 9)), T=int(rs.choice([4, 16, 31, 32, 33, 64, 97, 128])), Lq=int(rs.randint(1, 33)),
                     Lc=int(rs.choice([4, 5, 10, 17, 24])), Dv=int(rs.choice([4, 36, 64, 100])), char_dim=int(rs.choice([50, 50, 8, 64, 100])),
                     char_size=int(rs.choice([40, 17, 97])), word_table=bool(rs.randint(0, 3) == 0), word_dim=wd)
        try:
            check_shape_against_oracle(shape, scaled_bias_floor=True)
            print('ok   ', shape, flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print('FAIL ', shape, repr(e)[:600], flush=True)
            traceback.print_exc(limit=2)
    print('%d / %d shapes failed' % (bad, n))
    return bad


if __name__ == '__main__':
    sys.exit(1 if main(*(int(a) for a in sys.argv[1:3])) else 0)
