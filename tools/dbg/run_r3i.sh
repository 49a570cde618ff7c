for rep in 1 2; do for n in 0 3 2 1; do export VSL_WGRAD_CUS=$n; echo -n "wgrad_cus=$n/4: "; timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null < /dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(d['value'],d['ms_per_step'])"; done; done > gpurun_out/cumask.log 2>&1
cat gpurun_out/cumask.log
