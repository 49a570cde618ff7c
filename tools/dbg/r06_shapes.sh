#!/bin/bash
# the driver's `shapes` entries of one library build: bash tools/dbg/r06_shapes.sh [suffix of vslnet_amd/lib/libvslnet_hip<suffix>.so]
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = "-" ]; then lib=""; else lib="VSLNET_HIP_LIB=$PWD/vslnet_amd/lib/libvslnet_hip$v.so"; fi
  echo "== $v"
  env $lib python bench.py --steps 20 --warmup 5 --regions 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'])
for s in d.get('shapes', []): print(s.get('tag'), s.get('ms_per_step'), s.get('step_mfma_frac'))"
done
