"""main.py: flag surface of the reference CLI (main_t7.py:13-45) on CPU; one short end-to-end training + test-mode run on
the GPU (synthetic learnable dataset, both optimizer paths, checkpoint round trip)."""
import json
import os

import pytest
import torch

import main as cli


def test_flags_and_defaults_match_reference_cli():
    ns = cli.build_parser().parse_args([])
    want = dict(save_dir='datasets_t7', task='charades', fv='new', max_pos_len=128, word_size=None, char_size=None, word_dim=300,
                video_feature_dim=1024, char_dim=50, dim=128, highlight_lambda=5.0, num_heads=8, drop_rate=0.2, predictor='rnn',
                gpu_idx='0', seed=12345, mode='train', epochs=100, batch_size=16, num_train_steps=None, init_lr=0.0001,
                clip_norm=1.0, warmup_proportion=0.0, extend=0.1, period=100, model_dir='ckpt_t7', model_name='vslnet', suffix=None,
                data='resident')
    for k, v in want.items():
        assert getattr(ns, k) == v, k
    assert cli.build_parser().parse_args(['--hidden_size', '64']).dim == 64              # TF spelling (main.py:27)
    ns = cli.build_parser().parse_args(['--task', 'tacos', '--fv', 'org', '--max_pos_len', '256', '--predictor', 'transformer', '--suffix', 'x'])
    assert cli.model_home(ns) == os.path.join('ckpt_t7', 'vslnet_tacos_org_256_transformer_x', 'model')


def test_unknown_task_and_missing_dataset_raise_value_error(tmp_path):
    from vslnet_amd import data
    ns = cli.build_parser().parse_args(['--task', 'nope'])
    with pytest.raises(ValueError, match='Unknown task'):
        data.load_dataset(ns)
    ns = cli.build_parser().parse_args(['--task', 'charades', '--save_dir', str(tmp_path)])
    with pytest.raises(ValueError, match='not found'):
        data.load_dataset(ns)


@pytest.mark.gpu
@pytest.mark.parametrize('optimizer,predictor,source', [('fused', 'transformer', 'resident'), ('torch', 'transformer', 'resident'),
                                                        ('fused', 'rnn', 'resident'), ('fused', 'transformer', 'loader')])
def test_train_then_test_mode_on_synthetic_task(tmp_path, optimizer, predictor, source):
    argv = ['--task', 'synthetic', '--predictor', predictor, '--max_pos_len', '32', '--video_feature_dim', '64', '--batch_size', '16',
            '--epochs', '6', '--init_lr', '0.002', '--drop_rate', '0.1', '--period', '10', '--synthetic_train', '256', '--synthetic_test', '64',
            '--model_dir', str(tmp_path), '--optimizer', optimizer, '--data', source]
    lines = []
    out = cli.run(argv + ['--mode', 'train'], log=lines.append)
    losses = [v for _, v in out['history'] if isinstance(v, float)]
    evals = [v for _, v in out['history'] if isinstance(v, dict)]
    assert out['steps'] == 6 * 16 and len(evals) == 12
    assert losses[-1] < 0.6 * losses[0], losses                       # it learns
    assert evals[-1]['mIoU'] > evals[0]['mIoU'] + 5.0, evals          # and localises better than at the start
    files = os.listdir(out['model_dir'])
    assert 'configs.json' in files and 'eval_results.txt' in files and 1 <= sum(f.endswith('.t7') for f in files) <= 3
    cfg = json.load(open(os.path.join(out['model_dir'], 'configs.json')))
    assert cfg['num_train_steps'] == 96 and cfg['word_size'] == 200
    res = cli.run(argv + ['--mode', 'test'], log=lines.append)
    from vslnet_amd import runner
    best = os.path.basename(runner.get_last_checkpoint(out['model_dir']))
    step = int(best.split('_')[1].split('.')[0])
    at_step = [v for s, v in out['history'] if isinstance(v, dict) and s == step][-1]
    assert abs(res['mIoU'] - at_step['mIoU']) < 1e-3                  # the checkpoint reproduces the metrics it was saved with
    sd = torch.load(os.path.join(out['model_dir'], best), map_location='cpu')
    assert 'predictor.start_block.0.conv1d.weight' in sd and 'embedding_net.word_emb.glove_vec' in sd


@pytest.mark.gpu
def test_test_mode_loads_a_checkpoint_written_by_the_reference(tmp_path):
    """tests/golden/ref_checkpoint/: `torch.save(model.state_dict())` done BY THE REFERENCE (main_t7.py:125) on a randomly initialised model, and the
    metrics its own `eval_test` reports for it on the synthetic test split (oracle/make_golden.py: run_ref_checkpoint).  The `configs.json` beside it
    is THIS repo's argument namespace for the same flags (the reference's main_t7.py cannot be imported here -- nltk -- so it is not the file
    main_t7.py:81 would write); what the reference contributes is the `state_dict` and the metrics.  `main.py --mode test` (main_t7.py:132-149) must
    load the file and reproduce the metrics: with untrained weights that pins the loader, the key schema and `extract_index`, not a trained model."""
    import shutil
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_checkpoint')
    want = json.load(open(os.path.join(src, 'reference_metrics.json')))
    ns = cli.build_parser().parse_args(want['argv'] + ['--model_dir', str(tmp_path)])
    home = cli.model_home(ns)
    os.makedirs(home)
    shutil.copy(os.path.join(src, 'vslnet_5.t7'), home)
    shutil.copy(os.path.join(src, 'configs.json'), home)
    res = cli.run(want['argv'] + ['--model_dir', str(tmp_path), '--mode', 'test'], log=lambda *a: None)
    for k in ('r1i3', 'r1i5', 'r1i7', 'mIoU'):
        assert abs(res[k] - want[k]) < 1e-3, (k, res[k], want[k])


def test_rnn_predictor_has_the_reference_state_dict_entries():
    from vslnet_amd.model.VSLNet import VSLNet
    from vslnet_amd.synthetic import make_configs
    import numpy as np
    sd = VSLNet(make_configs(predictor='rnn', word_size=20), np.zeros((18, 300), dtype=np.float32)).state_dict()
    for enc in ('start_encoder', 'end_encoder'):
        for n, shp in (('weight_ih_l0', (512, 128)), ('weight_hh_l0', (512, 128)), ('bias_ih_l0', (512,)), ('bias_hh_l0', (512,))):
            assert tuple(sd['predictor.%s.lstm.%s' % (enc, n)].shape) == shp
    assert not any(k.startswith('predictor.encoder.') or 'layer_norm' in k and k.startswith('predictor.') for k in sd)
