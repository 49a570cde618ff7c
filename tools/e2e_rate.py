"""End-to-end training rate of main.py (data pipeline + H2D + step), synthetic dataset of the headline shape.
Two runs that differ only in the number of epochs isolate the per-step cost from start-up / dataset synthesis."""
import sys
import time

sys.path.insert(0, '.')
import main  # noqa: E402


def run(epochs, extra):
    argv = ['--task', 'synthetic', '--predictor', 'transformer', '--mode', 'train', '--batch_size', '64', '--max_pos_len', '128',
            '--synthetic_train', '2048', '--synthetic_test', '64', '--epochs', str(epochs), '--period', '100000',
            '--model_dir', '/tmp/e2e_ckpt'] + extra
    t0 = time.time()
    main.run(argv, log=lambda *a: None)
    return time.time() - t0


if __name__ == '__main__':
    extra = sys.argv[1:]
    run(1, extra)                       # warm-up (module load, plan build)
    a, b = run(2, extra), run(8, extra)
    steps = 6 * (2048 // 64)
    ms = (b - a) / steps * 1e3
    print('end-to-end: %.3f ms/step, %.0f pairs/s (incl. 2 evaluations of 64 test pairs per epoch)' % (ms, 64 / ms * 1e3))
