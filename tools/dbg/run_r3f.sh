timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3f_tests.log 2>&1; tail -5 gpurun_out/r3f_tests.log
( export VSLNET_HIP_LIB=$PWD/vslnet_amd/lib/libvslnet_hip_stamps.so VSL_DEBUG_TIMING=1 VSL_MULTI_STREAM=0
timeout 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/stamps.log 2>&1
grep "embed_" gpurun_out/stamps.log | head -4 )
timeout 600 bash tools/prof_serial.sh > gpurun_out/r3f_serial.log 2>&1; grep "embed\|reduce\|total kernel" gpurun_out/stats_serial.txt
for i in 1 2; do timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(d['value'],d['ms_per_step'])"; done
