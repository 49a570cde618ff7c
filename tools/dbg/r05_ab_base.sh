#!/bin/bash
# same-box A/B of the committed revision's library (tools/build_base.py <rev>) against the working tree's, N interleaved pairs
cd $GRAFT_REPO_ROOT; N=${1:-3}; shift
for rep in $(seq $N); do
  for v in base new; do
    if [ $v = base ]; then export VSLNET_HIP_LIB=$PWD/vslnet_amd/lib/libvslnet_hip_base.so; else unset VSLNET_HIP_LIB; fi
    echo -n "[$v] "; timeout 300 python bench.py --steps 60 --warmup 8 --regions 1 --no-shapes --no-cpu-baseline "$@" < /dev/null 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(d['value'],d['ms_per_step'])"
  done
done
