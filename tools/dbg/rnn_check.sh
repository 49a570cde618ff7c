mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q -k "rnn or lstm or fallback" 2>&1 | tail -5
for args in "--batch 16" "--batch 64" "--batch 16 --T 256"; do
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --predictor rnn $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:60], d['value'], d['ms_per_step'])"
done
