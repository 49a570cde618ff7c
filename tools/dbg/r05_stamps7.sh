#!/bin/bash
# phase stamps of several stamps builds: bash tools/dbg/r05_stamps7.sh <grep pattern> lib1 lib2 ...   (lib = suffix of libvslnet_hip_<lib>.so)
cd $GRAFT_REPO_ROOT; PAT="$1"; shift
for lib in "$@"; do
  echo "== $lib"
  ( export VSL_DEBUG_TIMING=1 VSL_MULTI_STREAM=0 VSLNET_HIP_LIB=$PWD/vslnet_amd/lib/libvslnet_hip_$lib.so; timeout 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --regions 1 --no-shapes < /dev/null 2>&1 | grep "$PAT" | tail -4 )
done
