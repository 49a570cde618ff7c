timeout 1500 python -m pytest tests -m gpu -x -q < /dev/null > gpurun_out/quick_tests.log 2>&1; tail -2 gpurun_out/quick_tests.log
timeout 400 bash tools/prof_serial.sh > gpurun_out/x_serial.log 2>&1 < /dev/null; grep "k_pack\|total kernel" gpurun_out/stats_serial.txt | cut -c1-130
for i in 1 2 3; do timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null < /dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(d['value'],d['ms_per_step'])"; done
