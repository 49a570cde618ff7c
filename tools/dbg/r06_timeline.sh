#!/bin/bash
# rocprofv3 kernel trace of the headline step -> gpurun_out/r06/<tag>_timeline.txt + kernel stats
cd $GRAFT_REPO_ROOT; R=$PWD; tag=${1:-t}; shift
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p2 && rocprofv3 --kernel-trace --stats -d /tmp/p2 -o s -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-shapes --regions 1 "$@" > /tmp/out.txt 2>/tmp/err.txt < /dev/null
cd $R
tail -5 /tmp/err.txt; tail -2 /tmp/out.txt
db=$(find /tmp/p2 -name "*.db" | head -1)
python tools/rocpd_timeline.py $db 15 > gpurun_out/r06/${tag}_timeline.txt
python tools/rocpd_stats.py $db > gpurun_out/r06/${tag}_kernel_stats.txt 2>&1
