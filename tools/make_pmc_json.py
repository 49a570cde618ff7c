"""Build profiles/<round>_pmc_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the
MI355X guide prescribes) and the kernel-trace pass: HBM bytes per LAUNCH of each bench.py kernel group.
usage: make_pmc_json.py <fetch.db> <write.db> <steps+warmup+1 of the profiled bench run> <out.json>"""
import json
import sqlite3
import sys

GROUP = {  # kernel function -> launcher group of vsl_profile_* (api.hip LAUNCH names)
    'k_attn_bwd_fused': 'attn_bwd', 'k_attn_bwd_dq': 'attn_bwd', 'k_attn_bwd_dkv': 'attn_bwd',
    'k_cq_bwd_a': 'cq_bwd', 'k_cq_bwd_b': 'cq_bwd', 'k_cq_bwd_c': 'cq_bwd', 'k_cq_bwd_d': 'cq_bwd',
    'k_loss_a': 'loss', 'k_loss_b': 'loss', 'k_loss_c': 'loss',
}
LAUNCHES_PER_STEP = {'wgrad': 11, 'attn_bwd': 4, 'cq_bwd': 1, 'loss': 1}


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, n, total in cur.execute('select kernel_name, count(*), sum(value) from counters_collection where counter_name=? '
                                      'group by kernel_name', (counter,)):
        short = name.split('(')[0].split('::')[-1].strip()
        out[short] = (n, total * 1024.0)                      # rocprofv3 reports KiB
    return out


def main(fetch_db, write_db, nsteps, out_path):
    nsteps = int(nsteps)
    f, w = per_kernel(fetch_db, 'FETCH_SIZE'), per_kernel(write_db, 'WRITE_SIZE')
    groups = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith('k_'):
            continue
        g = GROUP.get(k, k[2:])
        e = groups.setdefault(g, {'fetch': 0.0, 'write': 0.0, 'dispatches': 0})
        e['fetch'] += f.get(k, (0, 0.0))[1]
        e['write'] += w.get(k, (0, 0.0))[1]
        e['dispatches'] += f.get(k, (0, 0.0))[0]
    res = {}
    for g, e in groups.items():
        launches = LAUNCHES_PER_STEP.get(g, None)
        n = launches * nsteps if launches else e['dispatches']
        res[g] = {'fetch_bytes_x2': int(2 * e['fetch'] / n), 'write_bytes': int(e['write'] / n),
                  'traffic_bytes': int((2 * e['fetch'] + e['write']) / n), 'launches_counted': n}
    json.dump({'_note': 'HBM bytes per launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of bench.py, '
                        'B=64 T=128 Dv=1024). FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts half of wide coalesced reads); '
                        'WRITE_SIZE uncalibrated, as reported.', 'groups': res}, open(out_path, 'w'), indent=1)


if __name__ == '__main__':
    main(*sys.argv[1:5])
