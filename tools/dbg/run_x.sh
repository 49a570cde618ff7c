run() { echo -n "$1: "; shift; env "$@" timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null < /dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(d['value'],d['ms_per_step'])"; }
for rep in 1 2; do
run default A=1
run attn_waves8 VSL_ATTN_WAVES=8
run lds_spread VSL_LDS_SPREAD=86016
run queues4 GPU_MAX_HW_QUEUES=4
run queues16 GPU_MAX_HW_QUEUES=16
done
