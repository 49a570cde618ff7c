#!/bin/bash
# same-box A/B of environment settings, N interleaved rounds: bash tools/dbg/r05_ab_env.sh N "A=1 B=2" "A=0" ...   ("-" = no setting)
cd $GRAFT_REPO_ROOT; N=$1; shift
for rep in $(seq $N); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    echo -n "[$v] "; env $e timeout 300 python bench.py --steps 60 --warmup 8 --regions 1 --no-shapes --no-cpu-baseline < /dev/null 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(d['value'],d['ms_per_step'])"
  done
done
