run() { echo -n "$1: "; shift; env "$@" timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null < /dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(d['value'],d['ms_per_step'])"; }
for rep in 1 2 3; do
run "old embed, late " VSL_EMBED_FWD2=0 VSL_EARLY_QUERY=0
run "new embed, early" VSL_EMBED_FWD2=1 VSL_EARLY_QUERY=1
run "new embed, late " VSL_EMBED_FWD2=1 VSL_EARLY_QUERY=0
done
