#!/bin/bash
cd $GRAFT_REPO_ROOT
for lib in stamps stamps_NOEPIDC2_NOMFMA stamps_NOAREAD stamps_NOAREADDC2_NOEPI; do
  echo "== $lib"
  ( export VSL_CB2_WAVES=8 VSL_DEBUG_TIMING=1 VSL_MULTI_STREAM=0 VSLNET_HIP_LIB=$PWD/vslnet_amd/lib/libvslnet_hip_$lib.so; timeout 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline < /dev/null 2>&1 | grep "convblock_fwd\|L0:" | head -6 | tail -4 )
done
