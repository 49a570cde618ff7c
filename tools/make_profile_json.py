"""Build profiles/<tag>_profile.json -- what bench.py quotes as the STATIC half of its roofline object -- from the rocprofv3
passes of one `tools/collect_evidence.sh` call: the kernel trace (average launch duration per kernel group) and the two PMC
passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes).  Launches per step are counted against
the k_adamw dispatches of the same database (one per step), so they cannot drift from the stats file.
usage: make_profile_json.py <stats.db> <fetch.db> <write.db> <out.json> <tag>"""
import json
import sqlite3
import sys

GROUP = {  # kernel function -> launcher group of vsl_profile_* (api.hip LAUNCH names); default = name without "k_"
    'k_attn_bwd_fused': 'attn_bwd', 'k_attn_bwd_long': 'attn_bwd',
    'k_cq_bwd_a': 'cq_bwd', 'k_cq_bwd_b': 'cq_bwd', 'k_cq_bwd_c': 'cq_bwd',
    'k_loss_a': 'loss', 'k_loss_b': 'loss', 'k_loss_c': 'loss', 'k_loss_fused': 'loss', 'k_wgrad2': 'wgrad', 'k_wgrad3': 'wgrad', 'k_wgrad4': 'wgrad', 'k_sqsum': 'adamw',
    'k_vproj_fwd3': 'vproj_fwd', 'k_attn_block_fwd': 'attn_block_fwd',
}


def short(name):
    return name.split('(')[0].split('<')[0].split('::')[-1].strip().split(' ')[-1]


def durations(db):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, n, total in cur.execute('select name, count(*), sum(end-start) from kernels group by name'):
        k = short(name)
        e = out.setdefault(k, [0, 0.0])
        e[0] += n
        e[1] += total / 1e3
    return out


def counters(db, counter):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, n, total in cur.execute('select kernel_name, count(*), sum(value) from counters_collection where counter_name=? '
                                      'group by kernel_name', (counter,)):
        e = out.setdefault(short(name), [0, 0.0])
        e[0] += n
        e[1] += total * 1024.0                                # rocprofv3 reports KiB
    return out


def main(stats_db, fetch_db, write_db, out_path, tag):
    d, f, w = durations(stats_db), counters(fetch_db, 'FETCH_SIZE'), counters(write_db, 'WRITE_SIZE')
    nsteps = d['k_adamw'][0]                                    # one update per step (k_pack: three launches per step since round 4)
    groups = {}
    for k, (n, us) in d.items():
        if not k.startswith('k_'):
            continue
        g = groups.setdefault(GROUP.get(k, k[2:]), {'dispatches': 0, 'us': 0.0, 'fetch': 0.0, 'write': 0.0, 'fd': 0, 'wd': 0})
        g['dispatches'] += n
        g['us'] += us
        if k in f:
            g['fetch'] += f[k][1]
            g['fd'] += f[k][0]
        if k in w:
            g['write'] += w[k][1]
            g['wd'] += w[k][0]
    res = {}
    for name, g in groups.items():
        e = {'launches_per_step': round(g['dispatches'] / nsteps, 2), 'avg_us': round(g['us'] / g['dispatches'], 2),
             'us_per_step': round(g['us'] / nsteps, 1)}
        if g['fd'] and g['wd']:
            e.update(fetch_bytes_x2=int(2 * g['fetch'] / g['fd']), write_bytes=int(g['write'] / g['wd']),
                     traffic_bytes=int(2 * g['fetch'] / g['fd'] + g['write'] / g['wd']), launches_counted=g['fd'])
        res[name] = e
    json.dump({'_note': 'Per kernel group of bench.py (B=64 T=128 Dv=1024 Lq=20): average launch duration from rocprofv3 --kernel-trace, '
                        'HBM-side bytes per launch from separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes.  FETCH_SIZE doubled per '
                        'MI355X_MICROARCH.md (gfx950 counts half of wide coalesced reads); WRITE_SIZE uncalibrated, as reported; both '
                        'include Infinity-Cache hits.',
               'tag': tag, 'steps_profiled': nsteps, 'shape': [64, 128, 1024, 20],
               'sources': ['profiles/%s_kernel_stats.txt' % tag, 'profiles/%s_pmc_fetch_size.txt' % tag, 'profiles/%s_pmc_write_size.txt' % tag],
               'groups': res}, open(out_path, 'w'), indent=1)


if __name__ == '__main__':
    main(*sys.argv[1:6])
