#!/bin/bash
# same-box A/B: baseline .so (tools/build_base.py) vs the working build, interleaved; prints pairs/s and ms/step
cd "$(dirname "$0")/.."
for v in base new base new; do
  if [ $v = base ]; then export VSLNET_HIP_LIB=$PWD/vslnet_amd/lib/libvslnet_hip_base.so; else unset VSLNET_HIP_LIB; fi
  echo -n "$v: "; python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-optimizer "$@" 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(d['value'],d['ms_per_step'])"
done
