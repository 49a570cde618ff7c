import sys
sys.path.insert(0, '.')
from tests.test_hip_rnn import test_rnn_head_against_oracle as check_rnn
bad = 0
for shape in [dict(name='T1', B=3, T=1, Lq=4, Lc=6), dict(name='T2', B=2, T=2, Lq=3, Lc=5), dict(name='T3', B=5, T=3, Lq=8, Lc=10),
              dict(name='T5', B=1, T=5, Lq=2, Lc=4), dict(name='T6', B=4, T=6, Lq=9, Lc=7), dict(name='T9', B=7, T=9, Lq=20, Lc=10),
              dict(name='B80', B=80, T=24, Lq=6, Lc=5), dict(name='B81 (chunked)', B=81, T=40, Lq=6, Lc=5)]:
    try:
        check_rnn(shape)
        print('ok  ', shape, flush=True)
    except Exception as e:
        bad += 1
        print('FAIL', shape, repr(e)[:400], flush=True)
print(bad, 'failed')
