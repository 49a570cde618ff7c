#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
b() { echo -n "$1: "; env $2 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-optimizer 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(d['value'],d['ms_per_step'])"; }
{
for rep in 1 2; do
b base_noopt "VSL_WGRAD5=0"
b fwd_only "VSL_DBG_NOBWD=1"
b fwd_noq "VSL_DBG_NOBWD=1 VSL_DBG_NOQFWD=1"
b fwd_nopred "VSL_DBG_NOBWD=1 VSL_DBG_NOPFWD=1"
b fwd_noq_nopred "VSL_DBG_NOBWD=1 VSL_DBG_NOPFWD=1 VSL_DBG_NOQFWD=1"
b fwd_1stream "VSL_DBG_NOBWD=1 VSL_MULTI_STREAM=0"
b all_1stream "VSL_WGRAD5=0 VSL_MULTI_STREAM=0"
done
} > gpurun_out/r04/ceil2.txt 2>&1
cat gpurun_out/r04/ceil2.txt
