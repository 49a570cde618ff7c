#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
tag=${1:-q17}; shift
timeout 900 python -m pytest tests -m gpu -q -x -k "training or module or parity" > gpurun_out/r06/${tag}_tests.log 2>&1 < /dev/null
tail -2 gpurun_out/r06/${tag}_tests.log
bash tools/dbg/r05_ab_env.sh 4 "$@" > gpurun_out/r06/${tag}_ab.txt 2>&1 < /dev/null
cat gpurun_out/r06/${tag}_ab.txt
python tools/critical_path.py --out gpurun_out/r06/${tag}_critical_path.txt > /dev/null 2> gpurun_out/r06/${tag}_err.txt; tail -1 gpurun_out/r06/${tag}_err.txt
