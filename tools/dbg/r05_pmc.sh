#!/bin/bash
# SQ counters of the conv-block kernels (own run per counter set): bash tools/dbg/r05_pmc.sh <tag> "<counters>" [env...]
R=$GRAFT_REPO_ROOT; TAG=$1; CTR="$2"; shift 2
mkdir -p $R/gpurun_out/r05; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
env "$@" rocprofv3 --kernel-trace --pmc $CTR -d /tmp/pmc_$TAG -o p -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/r05/pmc_$TAG.err
python - <<PY > $R/gpurun_out/r05/pmc_$TAG.txt
import sqlite3, glob
db = glob.glob('/tmp/pmc_$TAG/**/*.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute('select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name order by 1, 2'))
for k, c, n, avg in rows:
    if 'convblock' in k or 'attn_block' in k or 'wgrad' in k: print('%-50s %-28s %5d %14.1f' % (k[:50], c, n, avg))
PY
cat $R/gpurun_out/r05/pmc_$TAG.txt
