"""Print the per-stream kernel timeline of ONE benchmark step (the kernels between two k_adamw launches: a step ends with its update; since
round 4 a step opens with three k_pack launches on three streams) from a rocprofv3 rocpd database: start/end (us, relative to the step
start), stream index, kernel, duration."""
import sqlite3
import sys


def main(db_path, step=12, out=sys.stdout):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute('select name, start, end, stream_id from kernels order by start'))
    ends = [i for i, r in enumerate(rows) if 'k_adamw' in r[0]]
    i0, i1 = ends[step] + 1, ends[step + 1] + 1
    t0 = rows[i0][1]
    st = rows[i0:i1]
    streams = sorted(set(r[3] for r in st))
    out.write('# %s  step %d: span %.1f us, %d kernels, %d streams\n' % (db_path, step, (max(r[2] for r in st) - t0) / 1e3, len(st), len(streams)))
    for r in st:
        nm = r[0].split('(')[0].replace('vsl::k_', '')
        out.write('%8.1f %8.1f  s%-2d %-18s %6.1f\n' % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, streams.index(r[3]), nm[:18], (r[2] - r[1]) / 1e3))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12)
