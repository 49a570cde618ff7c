export VSL_DEBUG_TIMING=1 VSL_MULTI_STREAM=0
export VSLNET_HIP_LIB=$PWD/vslnet_amd/lib/libvslnet_hip_stamps.so
timeout 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/stamps.log 2>&1
grep "wgrad4" gpurun_out/stamps.log | head -5
unset VSLNET_HIP_LIB VSL_DEBUG_TIMING
timeout 600 bash tools/prof_serial.sh > gpurun_out/r3f_serial.log 2>&1; grep "wgrad\|total kernel" gpurun_out/stats_serial.txt
unset VSL_MULTI_STREAM
