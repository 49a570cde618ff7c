#!/bin/bash
# same-box A/B of environment switches, three interleaved pairs: tools/dbg/r05_ab.sh "NAME=VAL ..." "NAME=VAL ..." [bench args]
cd $GRAFT_REPO_ROOT
A="$1"; B="$2"; shift 2
for rep in 1 2 3; do
  for v in "$A" "$B"; do
    echo -n "[$v] "; env $v timeout 300 python bench.py --steps 60 --warmup 8 --no-cpu-baseline "$@" < /dev/null 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(d['value'],d['ms_per_step'])"
  done
done
