"""Dump the per-kernel summary (calls, total / avg / min / max duration, share) of a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes DIR/NAME_results.db on ROCm 7.2) as text."""
import sqlite3
import sys


def main(db_path, out=sys.stdout, top=60):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute('select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), '
                            'max(vgpr_count), max(lds_size) from kernels group by name order by 3 desc'))
    tot = sum(r[2] for r in rows)
    out.write('# source: %s\n# total kernel time %.1f us over %d dispatches\n' % (db_path, tot / 1e3, sum(r[1] for r in rows)))
    out.write('%-64s %7s %12s %10s %10s %10s %6s %5s %7s\n' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'pct', 'vgpr', 'lds'))
    for r in rows[:top]:
        out.write('%-64s %7d %12.1f %10.2f %10.2f %10.2f %6.1f %5s %7s\n' %
                  (r[0][:64], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, r[6], r[7]))


if __name__ == '__main__':
    main(sys.argv[1])
