"""ms per step of the headline shape at a few batch lengths (same box): how far the lengths that are not multiples of the 32-row tile are
from the whole-tile fast path.  python tools/dbg/r06_ragged.py [T ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
Ts = [int(t) for t in sys.argv[1:]] or [128, 117, 96, 100, 64, 75]
for rep in range(2):
    for T in Ts:
        ms, pps, loss = bench.time_shape('transformer', 64, T, 1024, 20, 10, 0.2, 'f32', 30, 6, 10)
        print('T=%4d  %.4f ms/step  %9.1f pairs/s  %.3f us per row-tile-equivalent (ms / ceil(T/32))' % (T, ms, pps, 1e3 * ms / ((T + 31) // 32)), flush=True)
