#!/bin/bash
# same-box A/B of library builds on another bench shape: bash tools/dbg/r05_ab_libs_args.sh N "bench args" lib1 lib2 ...   ("tree" = the tree's)
cd $GRAFT_REPO_ROOT; N=$1; ARGS="$2"; shift 2
for rep in $(seq $N); do
  for v in "$@"; do
    if [ "$v" = "tree" ]; then unset VSLNET_HIP_LIB; else export VSLNET_HIP_LIB=$PWD/vslnet_amd/lib/libvslnet_hip_$v.so; fi
    echo -n "[$v] "; timeout 300 python bench.py --steps 20 --warmup 5 --regions 1 --no-shapes --no-cpu-baseline $ARGS < /dev/null 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(d['value'],d['ms_per_step'])"
  done
done
