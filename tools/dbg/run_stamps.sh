export VSL_DEBUG_TIMING=1 VSL_MULTI_STREAM=0
for v in amps 1 2 3; do
export VSLNET_HIP_LIB=$PWD/vslnet_amd/lib/libvslnet_hip_st$v.so
timeout 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/stamps_$v.log 2>&1
echo "variant $v"; grep "embed_bwd" gpurun_out/stamps_$v.log | head -2
done
