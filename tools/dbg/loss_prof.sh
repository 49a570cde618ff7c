R=$PWD; mkdir -p $R/gpurun_out/r04; cd /tmp && export TMPDIR=/tmp
for tag in new base; do
  if [ $tag = base ]; then export VSLNET_HIP_LIB=$R/vslnet_amd/lib/libvslnet_hip_base.so; else unset VSLNET_HIP_LIB; fi
  rm -rf /tmp/lp_$tag
  rocprofv3 --kernel-trace --stats -d /tmp/lp_$tag -o s -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline > /dev/null 2> /dev/null
  python $R/tools/rocpd_stats.py $(find /tmp/lp_$tag -name "*.db" | head -1) | grep -i "loss" | cut -c1-150
done
