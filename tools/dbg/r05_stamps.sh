#!/bin/bash
# phase stamps (stamps build, one stream) + rocprofv3 kernel stats of the current build: bash tools/dbg/r05_stamps.sh <tag> [filter]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05; TAG=$1; F=${2:-convblock}
( export VSL_DEBUG_TIMING=1 VSL_MULTI_STREAM=0 VSLNET_HIP_LIB=$PWD/vslnet_amd/lib/libvslnet_hip_stamps.so; timeout 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline < /dev/null > gpurun_out/r05/${TAG}_stamps.log 2>&1 )
grep cycles gpurun_out/r05/${TAG}_stamps.log | grep "$F\|L0:" | tail -8
bash tools/dbg/prof_shape.sh $TAG < /dev/null; cp gpurun_out/r04/${TAG}_kernel_stats.txt gpurun_out/r04/${TAG}_timeline.txt gpurun_out/r05/
grep "$F\|total kernel" gpurun_out/r05/${TAG}_kernel_stats.txt | cut -c1-150
