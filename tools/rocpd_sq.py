"""Per-kernel SQ issue / stall breakdown from a rocprofv3 rocpd database collected with
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
(MI355X_MICROARCH.md: WAIT_ANY = wave parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing; the three
are disjoint and sum to about WAVE_CYCLES; all in quad-cycles summed over waves)."""
import sqlite3
import sys


def main(db_path, out=sys.stdout):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute('select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name'))
    ker = {}
    for k, c, n, v in rows:
        ker.setdefault(k, {})[c] = v
        ker[k]['_n'] = n
    names = ['SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS', 'SQ_LDS_BANK_CONFLICT']
    out.write('# source: %s\n%-52s %6s %12s' % (db_path, 'kernel', 'calls', 'wave_cyc/call') + ''.join(' %9s' % n[3:12] for n in names) + '   (fractions of SQ_WAVE_CYCLES)\n')
    for k, d in sorted(ker.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
        wc = d.get('SQ_WAVE_CYCLES', 0)
        if not wc:
            continue
        out.write('%-52s %6d %12.0f' % (k[:52], d['_n'], wc / d['_n']) + ''.join(' %8.1f%%' % (100.0 * d.get(n, 0) / wc) for n in names) + '\n')


if __name__ == '__main__':
    main(sys.argv[1])
