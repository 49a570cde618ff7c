#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for lib in stamps stamps_NOEPI stamps_NOMFMA; do for wv in 8 16; do
  echo "== $lib waves $wv"
  ( export VSL_CB2_WAVES=$wv VSL_DEBUG_TIMING=1 VSL_MULTI_STREAM=0 VSLNET_HIP_LIB=$PWD/vslnet_amd/lib/libvslnet_hip_$lib.so; timeout 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline < /dev/null 2>&1 | grep "convblock_fwd\|L0:" | tail -2 )
done; done
