"""Random-shape parity sweep on the GPU: forward logits, losses, every gradient and extract_index against the oracle
(the body of tests/test_hip_training.py::test_baseline_shapes_against_oracle) for shapes the fixed test list does not hold:
ragged row counts, odd feature widths, one-word queries, single clips ...   usage: python tools/fuzz_parity.py [n] [seed]"""
import sys
import traceback

import numpy as np

sys.path.insert(0, '.')
from tests.test_hip_training import test_baseline_shapes_against_oracle as check  # noqa: E402
from tests.test_hip_rnn import test_rnn_head_against_oracle as check_rnn  # noqa: E402


def main(n=24, seed=0):
    rs = np.random.RandomState(seed)
    bad = 0
    for i in range(n):
        shape = dict(name='fuzz %d' % i, B=int(rs.randint(1, 7)), T=int(rs.choice([4, 7, 16, 31, 32, 33, 40, 50, 64, 97, 128, 160, 256])),
                     Lq=int(rs.choice([1, 2, 3, 8, 20, 31, 32, 33, 47, 64, 65, 82, 96, 97, 111, 128])), Lc=int(rs.choice([4, 5, 10, 17, 24, 25, 40])),
                     Dv=int(rs.choice([4, 36, 64, 100, 500, 1024])), char_dim=int(rs.choice([50, 50, 8, 64, 65, 100, 128])),
                     char_size=int(rs.choice([40, 40, 17, 97, 200])), word_table=bool(rs.randint(0, 3) == 0))
        try:
            if i % 4 == 3:                      # every fourth shape goes through the rnn head (chunk-pipelined LSTMs, Dv = 64)
                # (that test fixes max_pos_len = 128 and has no structural-zero gate for one-word queries)
                shape = dict(name=shape['name'] + ' rnn', B=shape['B'] * 4 - 1, T=min(shape['T'], 128), Lq=min(max(shape['Lq'], 2), 128), Lc=shape['Lc'])
                check_rnn(shape)
            else:
                check(shape)
            print('ok   ', shape, flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print('FAIL ', shape, repr(e)[:600], flush=True)
            traceback.print_exc(limit=2)
    print('%d / %d shapes failed' % (bad, n))
    return bad


if __name__ == '__main__':
    sys.exit(1 if main(*(int(a) for a in sys.argv[1:3])) else 0)
