#!/bin/bash
# one bench line per BASELINE shape that fits one GPU (incl. optimizer): value, ms/step, fp32-MFMA roofline fraction
cd "$(dirname "$0")/.."
run() { echo -n "$1: "; shift; python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(d['value'],d['ms_per_step'],d['roofline'].get('step_mfma_frac'))"; }
run "cfg2 B=64 T=128 Dv=1024"
run "cfg3 B=32 T=256 Dv=4096" --batch 32 --T 256 --dv 4096
run "cfg4 B=32 T=256 Dv=1024" --batch 32 --T 256
run "cfg5 B=16 T=1024 Dv=1024" --batch 16 --T 1024
run "cfg1 rnn B=16" --predictor rnn --batch 16
run "rnn B=64" --predictor rnn --batch 64
