#!/bin/bash
# same-box A/B of (library, environment) pairs: bash tools/dbg/r06_ab_mixed.sh N "lib|ENV=.. ENV=.." ...   (lib: tree or a suffix of libvslnet_hip_<suffix>.so; env may be empty)
cd $GRAFT_REPO_ROOT; N=$1; shift
for rep in $(seq $N); do
  for v in "$@"; do
    lib=${v%%|*}; e=${v#*|}
    if [ "$lib" = "tree" ]; then L=""; else L="VSLNET_HIP_LIB=$PWD/vslnet_amd/lib/libvslnet_hip_$lib.so"; fi
    echo -n "[$v] "; env $L $e timeout 300 python bench.py --steps 60 --warmup 8 --regions 1 --no-shapes --no-cpu-baseline < /dev/null 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(d['value'],d['ms_per_step'])"
  done
done
