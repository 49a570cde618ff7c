python -m pytest tests/test_hip_parity.py tests/test_hip_training.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(d['value'],d['ms_per_step'])"; done
bash tools/bench_shapes.sh 2>&1 | sed -n 2,6p
