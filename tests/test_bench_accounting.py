"""bench.py's accounting, on CPU: the algorithmic FLOPs per pair are SURVEY.md 8(d)'s values for every BASELINE config (the figure
`roofline.step_mfma_frac` rests on), the per-kernel work table is consistent with it, and the workload names map to BASELINE.json's configs."""
import json
import os
import types

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_algorithmic_flops_match_survey_8d():
    # (predictor, T, Dv) -> (forward, forward + backward) MFLOP per pair, SURVEY.md 8(d) "Values"; Lq = 20, Lc = 10
    want = {('rnn', 128, 1024): (204.0, 578.5), ('transformer', 128, 1024): (222.6, 634.3), ('transformer', 256, 4096): (694.4, 1814.6),
            ('transformer', 256, 1024): (493.0, 1412.0), ('transformer', 1024, 1024): (3376.1, 9859.8)}
    for (pred, T, Dv), (f, fb) in want.items():
        gf, gfb = bench.alg_flops_per_pair(T, Dv, 20, 10, predictor=pred)
        assert abs(gf / 1e6 - f) < 0.06 and abs(gfb / 1e6 - fb) < 0.06, (pred, T, Dv, gf / 1e6, gfb / 1e6)


def test_weight_gradient_work_is_the_sum_of_its_jobs():
    B, T, Dv, Lq, d = 64, 128, 1024, 20, 128
    R, Rq = B * T, B * Lq
    flops, nbytes = bench.kernel_work('wgrad', B, T, Dv, Lq)
    # 8 (128 x 128) weights per encoder application (3 video-length + 1 query-length), two (128, 256) heads, cqa (128, 512), cat half (128, 128),
    # embedding (128, 400), VisualProjection (128, Dv): dW = G^T A is 2 * rows * N * K
    jobs = 2 * d * ((3 * R + Rq) * 8 * d + 2 * R * 2 * d + R * 4 * d + R * d + Rq * 400 + R * Dv)
    assert flops == jobs
    assert abs(flops / 10 - 1147247001) < 1e3            # per launch, the figure in the committed bench lines
    assert nbytes > 0


def test_workload_names_follow_baseline_json():
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    assert 'T=128' in base['configs'][1] and 'batch 64' in base['configs'][1]
    ns = types.SimpleNamespace(predictor='transformer', batch=64, T=128, dv=1024)
    assert bench.workload_name(ns).startswith('configs[1]')
    assert bench.workload_name(types.SimpleNamespace(predictor='rnn', batch=16, T=128, dv=1024)).startswith('configs[0]')
    assert bench.workload_name(types.SimpleNamespace(predictor='transformer', batch=32, T=256, dv=4096)).startswith('configs[2]')
    assert bench.PEAK_MFMA_F32 == 157.3e12 and bench.PEAK_HBM == 8.0e12
    assert os.path.exists(os.path.join(ROOT, 'profiles', bench.PROFILE_JSON))


def test_other_shapes_cover_the_remaining_baseline_configs_and_the_bf16_mode():
    """The `shapes` list bench.py appends behind the headline regions (driver-observed numbers for every BASELINE config):
    its entries are the per-GPU shapes of configs[0], [2], [3], [4] and the bf16 mode of configs[1]."""
    tags = [t for t, *_ in bench.OTHER_SHAPES]
    assert tags == ['configs[0]', 'configs[2]', 'configs[3]/GPU', 'configs[4]/GPU', 'configs[1] --dtype bf16', 'ragged T=117']
    for tag, kw, steps, warmup, nres in bench.OTHER_SHAPES:
        ns = types.SimpleNamespace(predictor=kw['predictor'], batch=kw['batch'], T=kw['T'], dv=kw['dv'])
        if tag.startswith('ragged'):          # round 6: a batch length off the 32-row tile (what a collated Charades batch looks like): no BASELINE config
            assert kw['T'] % 32 != 0 and bench.workload_name(ns).startswith('custom shape')
        else:
            assert bench.workload_name(ns).startswith(tag.split('/')[0].split(' ')[0]), (tag, bench.workload_name(ns))
        assert steps >= 10 and warmup >= 3 and nres >= 3
        # rotated resident batches of every shape exceed the 256 MiB Infinity Cache only where that is cheap; they must at least differ step to step
        assert kw.get('dtype', 'f32') in ('f32', 'bf16')
    src = open(os.path.join(ROOT, 'bench.py')).read()
    for key in ("'regions'", "'region_ms'", "'shapes'", "'pairs_per_s'", "'step_mfma_frac'"):
        assert key in src, key
    # the headline fields stay the median region's: value = B * world * steps / dt with dt the median of region_dt
    assert 'dt = sorted(region_dt)[(len(region_dt) - 1) // 2]' in src
