#!/bin/bash
# conv-block forward stamps of several stamp libraries / wave counts: bash tools/dbg/r05_stamps2.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for lib in stamps stamps_seq; do for wv in 8 16; do
  echo "== $lib waves $wv"
  ( export VSL_CB2_WAVES=$wv VSL_DEBUG_TIMING=1 VSL_MULTI_STREAM=0 VSLNET_HIP_LIB=$PWD/vslnet_amd/lib/libvslnet_hip_$lib.so; timeout 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline < /dev/null 2>&1 | grep "convblock_fwd\|L0:" | tail -4 )
done; done
for lib in "" _seq; do for wv in 8 16; do
  echo -n "== lib$lib waves $wv: "; ( export VSL_CB2_WAVES=$wv VSLNET_HIP_LIB=$PWD/vslnet_amd/lib/libvslnet_hip$lib.so; timeout 300 python bench.py --steps 60 --warmup 8 --no-cpu-baseline < /dev/null 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(d['value'],d['ms_per_step'])" )
done; done
