#!/bin/bash
# round 6: first run of the sample-local query forward: parity suite, then same-box A/B against the row-tile launches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 -x -k "parity or training or fuzz or module" > gpurun_out/r06/q1_tests.log 2>&1 < /dev/null
tail -40 gpurun_out/r06/q1_tests.log
bash tools/dbg/r05_ab_env.sh 3 "VSL_QUERY_FUSED=0" - > gpurun_out/r06/q1_ab.txt 2>&1 < /dev/null
cat gpurun_out/r06/q1_ab.txt
