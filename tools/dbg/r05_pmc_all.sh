#!/bin/bash
# per-kernel SQ instruction counters of one bench run (own rocprofv3 pass per counter set): bash tools/dbg/r05_pmc_all.sh <tag> "<counters>"
R=$GRAFT_REPO_ROOT; TAG=$1; CTR="$2"; shift 2
mkdir -p $R/gpurun_out/r05; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
env "$@" rocprofv3 --kernel-trace --pmc $CTR -d /tmp/pmc_$TAG -o p -- python $R/bench.py --steps 6 --warmup 3 --regions 1 --no-shapes --no-cpu-baseline > /dev/null 2> $R/gpurun_out/r05/pmc_$TAG.err
python - <<PY > $R/gpurun_out/r05/pmc_$TAG.txt
import sqlite3, glob, collections
db = glob.glob('/tmp/pmc_$TAG/**/*.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute('select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name'))
tab = collections.defaultdict(dict); calls = {}
for k, c, n, sm in rows:
    tab[k][c] = sm; calls[k] = n
ctrs = sorted({c for k in tab for c in tab[k]})
nsteps = 9.0   # steps + warmup of the run (every launch counted)
print('%-52s %6s' % ('kernel (sums per STEP, millions)', 'calls') + ''.join('%16s' % c[:15] for c in ctrs))
tot = collections.Counter()
for k in sorted(tab, key=lambda k: -tab[k].get(ctrs[0], 0)):
    print('%-52s %6.1f' % (k[:52], calls[k] / nsteps) + ''.join('%16.3f' % (tab[k].get(c, 0) / nsteps / 1e6) for c in ctrs))
    for c in ctrs: tot[c] += tab[k].get(c, 0) / nsteps / 1e6
print('%-52s %6s' % ('TOTAL', '') + ''.join('%16.3f' % tot[c] for c in ctrs))
PY
head -45 $R/gpurun_out/r05/pmc_$TAG.txt
