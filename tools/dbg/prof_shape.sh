#!/bin/bash
# rocprofv3 kernel stats + one-step timeline of one bench shape: bash tools/dbg/prof_shape.sh <tag> [bench args]
R=$PWD; TAG=$1; shift
mkdir -p $R/gpurun_out/r04; rm -rf /tmp/ps_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ps_$TAG -o s -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline "$@" > /dev/null 2> $R/gpurun_out/r04/ps_$TAG.err
DB=$(find /tmp/ps_$TAG -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $R/gpurun_out/r04/${TAG}_kernel_stats.txt
python $R/tools/rocpd_timeline.py $DB 8 > $R/gpurun_out/r04/${TAG}_timeline.txt
