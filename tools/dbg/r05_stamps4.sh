#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
( export VSL_DEBUG_TIMING=1 VSL_MULTI_STREAM=0 VSLNET_HIP_LIB=$PWD/vslnet_amd/lib/libvslnet_hip_stamps.so; timeout 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline < /dev/null 2>&1 | grep "convblock_fwd\|L0:" | tail -4 )
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_training.py -m gpu -x -q < /dev/null 2>&1 | tail -3
bash tools/dbg/r05_ab.sh VSL_CB2=0 VSL_CB2=1
bash tools/dbg/prof_shape.sh cb2c < /dev/null; grep "convblock" gpurun_out/r04/cb2c_kernel_stats.txt | cut -c1-150
