"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --pmc X -d DIR -o NAME`).
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch.  Per MI355X_MICROARCH.md (HBM section), on gfx950
FETCH_SIZE counts exactly HALF of the bytes of a wide coalesced streaming read -> the `fetch_x2` column doubles it;
WRITE_SIZE is uncalibrated (reported as is)."""
import sqlite3
import sys


def main(db_path, out=sys.stdout):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute('select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection '
                            'group by kernel_name, counter_name order by 5 desc'))
    out.write('# source: %s\n%-60s %-12s %7s %14s %14s\n' % (db_path, 'kernel', 'counter', 'calls', 'avg_KiB', 'avg_KiB_x2(FETCH)'))
    for k, c, n, avg, _ in rows[:40]:
        out.write('%-60s %-12s %7d %14.1f %14.1f\n' % (k[:60], c, n, avg, 2 * avg if c == 'FETCH_SIZE' else avg))


if __name__ == '__main__':
    main(sys.argv[1])
