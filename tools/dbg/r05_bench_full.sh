#!/bin/bash
# the driver-style bench line (all regions, shapes, cpu baseline) into gpurun_out/r05/<tag>_bench.json
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05; TAG=${1:-try}
T0=$(date +%s); python bench.py --steps 20 --warmup 10 > gpurun_out/r05/${TAG}_bench.json 2> gpurun_out/r05/${TAG}_bench.err < /dev/null
echo "wall $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/r05/${TAG}_bench.err
python - <<PY
import json
d = json.load(open('gpurun_out/r05/${TAG}_bench.json'))
print(d['value'], d['ms_per_step'], d['region_ms'], d['roofline'].get('step_mfma_frac'))
for s in d.get('shapes', []): print(s['tag'], s['pairs_per_s'], s['ms_per_step'], s['step_mfma_frac'])
print('cpu', d.get('cpu_baseline', {}).get('value'))
PY
