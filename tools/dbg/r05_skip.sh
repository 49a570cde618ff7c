#!/bin/bash
# wall-clock contribution of each launch group: the step with that group's launches skipped (garbage results, timing only; needs the `ko` build of
# tools/dbg/README.md: LAUNCH() honours VSL_KO_SKIP=name,name).  bash tools/dbg/r05_skip.sh group1 group2 ...
cd $GRAFT_REPO_ROOT; export VSLNET_HIP_LIB=$PWD/vslnet_amd/lib/libvslnet_hip_ko.so
for g in none "$@" none; do
  echo -n "[skip $g] "; VSL_KO_SKIP=$g timeout 300 python bench.py --steps 60 --warmup 8 --regions 1 --no-shapes --no-cpu-baseline < /dev/null 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(d['ms_per_step'])"
done
