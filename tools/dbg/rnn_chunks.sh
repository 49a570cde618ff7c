# sweep of the rnn head's time-chunk count (VSL_DBG_CHUNKS: debug build only)
mkdir -p gpurun_out/r04
for shape in "--batch 16" "--batch 64" "--batch 16 --T 256" "--batch 16 --T 64"; do
for n in 1 2 3 4 5 6 8; do
echo -n "$shape chunks=$n: "
VSL_DBG_CHUNKS=$n python bench.py --steps 30 --warmup 10 --no-cpu-baseline --predictor rnn $shape 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
