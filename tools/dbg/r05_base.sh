#!/bin/bash
# round-5 baseline on one box: three bench runs, phase stamps, per-kernel rocprofv3 stats
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for i in 1 2 3; do timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline < /dev/null 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print('bench',d['value'],d['ms_per_step'])"; done > gpurun_out/r05/base_bench.txt 2>&1
( export VSL_DEBUG_TIMING=1 VSL_MULTI_STREAM=0 VSLNET_HIP_LIB=$PWD/vslnet_amd/lib/libvslnet_hip_stamps.so; timeout 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline < /dev/null > gpurun_out/r05/base_stamps.log 2>&1 )
bash tools/dbg/prof_shape.sh base < /dev/null; cp gpurun_out/r04/base_kernel_stats.txt gpurun_out/r04/base_timeline.txt gpurun_out/r05/ 2>/dev/null
hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o /tmp/valu 2>/dev/null && /tmp/valu > gpurun_out/r05/valu_rates.txt 2>&1
cat gpurun_out/r05/base_bench.txt; grep cycles gpurun_out/r05/base_stamps.log | head -20; head -12 gpurun_out/r05/base_kernel_stats.txt | cut -c1-150; cat gpurun_out/r05/valu_rates.txt
