#!/bin/bash
cd $GRAFT_REPO_ROOT
for lib in stamps stamps_NOSTORE stamps_NOHALO stamps_NORED stamps_NOGW stamps_NOSTOREDCBB_NOHALODCBB_NOREDDCBB_NOGW; do
  echo "== $lib"
  ( export VSL_DEBUG_TIMING=1 VSL_MULTI_STREAM=0 VSLNET_HIP_LIB=$PWD/vslnet_amd/lib/libvslnet_hip_$lib.so; timeout 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --regions 1 --no-shapes < /dev/null 2>&1 | grep "L3:" | tail -2 )
done
