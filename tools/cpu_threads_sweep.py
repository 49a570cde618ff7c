"""Thread sweep of bench.py's cpu_baseline leg (the pinned CPU oracle, oracle/vslnet_oracle.py) on this box's host cores:
    python tools/cpu_threads_sweep.py [threads ...]      (default 16 32 64 128)
Prints pairs/s at B = 16 and B = 64 per thread count; bench.py uses the best one found on the GPU boxes (profiles/r03_notes.md)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vslnet_amd.synthetic import make_configs  # noqa: E402

if __name__ == '__main__':
    threads = [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128]
    configs = make_configs(video_feature_dim=1024, max_pos_len=128, drop_rate=0.2, predictor='transformer')
    for t in threads:
        r = bench.cpu_baseline(configs, 128, 20, 10, threads=t, budget_s=16.0)
        print(json.dumps({'threads': r['cores'], 'pairs_per_s_b16': r['value'], 'pairs_per_s_b64': r['value_b64'], 'cpu': r['cpu_model'],
                          'physical_cores': r['physical_cores']}), flush=True)
