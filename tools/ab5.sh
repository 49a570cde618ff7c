#!/bin/bash
# same-box A/B, 5 interleaved pairs, with optimizer; prints ms/step of each run and the medians
cd "$(dirname "$0")/.."
B=(); N=()
for i in 1 2 3 4 5; do
  export VSLNET_HIP_LIB=$PWD/vslnet_amd/lib/libvslnet_hip_base.so
  B+=($(python bench.py --steps 40 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys;print(json.load(sys.stdin)['ms_per_step'])"))
  unset VSLNET_HIP_LIB
  N+=($(python bench.py --steps 40 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys;print(json.load(sys.stdin)['ms_per_step'])"))
done
echo "base: ${B[*]}"; echo "new:  ${N[*]}"
python - "${B[*]}" "${N[*]}" <<'PY'
import sys, statistics
b=[float(x) for x in sys.argv[1].split()]; n=[float(x) for x in sys.argv[2].split()]
print('median base %.4f new %.4f  (%.2f %%)' % (statistics.median(b), statistics.median(n), 100*(statistics.median(n)/statistics.median(b)-1)))
PY
