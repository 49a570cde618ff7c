"""Per-kernel MFMA utilisation from a rocprofv3 rocpd database collected with
`rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`:
  SQ_VALU_MFMA_BUSY_CYCLES = cycles a SIMD's matrix pipe is busy, summed over the chip (64 per v_mfma_f32_32x32x2_f32,
  MI355X_MICROARCH.md; checked: k_vproj_fwd = 524288 MFMAs * 64 = 33.55 M), GRBM_GUI_ACTIVE = busy shader-clock cycles of the
  dispatch SUMMED OVER THE 8 XCDs (value / 8 = kernel-trace duration * clock).
  util = busy / ((active / 8) * 256 CUs * 4 SIMDs).
Counter collection serialises the dispatches, so these are isolated-kernel figures (no cross-stream overlap)."""
import sqlite3
import sys


def main(db_path, out=sys.stdout):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute('select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name'))
    by = {}
    for k, c, n, avg in rows:
        by.setdefault(k, {})[c] = (n, avg)
    out.write('# source: %s\n%-58s %6s %16s %14s %10s\n' % (db_path, 'kernel', 'calls', 'mfma_busy_cyc', 'gui_active(x8)', 'mfma_util'))
    tab = []
    for k, d in by.items():
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'GRBM_GUI_ACTIVE' in d:
            n, busy = d['SQ_VALU_MFMA_BUSY_CYCLES']
            act = d['GRBM_GUI_ACTIVE'][1]
            tab.append((busy * n, k, n, busy, act, busy / max(act / 8.0 * 1024.0, 1.0)))
    for _, k, n, busy, act, util in sorted(tab, reverse=True)[:40]:
        out.write('%-58s %6d %16.0f %14.0f %9.1f%%\n' % (k[:58], n, busy, act, 100 * util))
    tb, ta = sum(t[3] * t[2] for t in tab), sum(t[4] * t[2] for t in tab)
    out.write('# all kernels: busy %.3e cycles, active %.3e cycles (serialised), util %.1f%%\n' % (tb, ta, 100 * tb / max(ta / 8.0 * 1024.0, 1.0)))


if __name__ == '__main__':
    main(sys.argv[1])
