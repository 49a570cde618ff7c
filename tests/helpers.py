"""Shared helpers for the test-suite: golden fixture loading (tests/golden/*.npz, made by oracle/make_golden.py)."""
import ast
import os
from types import SimpleNamespace

import numpy as np
import torch

from oracle import vslnet_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    cfg = SimpleNamespace(**ast.literal_eval(str(z['cfg'])))
    P = O.random_params(cfg, seed=int(z['param_seed']))
    # the fixture stores checksums of the weights the reference actually ran with: regenerated weights must match
    for k, v in P.items():
        ref = z['sdsum.' + k]
        v64 = v.double()
        got = np.array([float(v64.sum()), float(v64.abs().sum()), float(v64.flatten()[-1])])
        assert np.allclose(got, ref, rtol=1e-12, atol=1e-12), 'regenerated weights differ from fixture: ' + k
    batch = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('in.')}
    return cfg, P, batch, z


def grad_tol(g_ref):
    """per-tensor gradient gate 1e-4 * ||g||_inf + 1e-6."""
    return 1e-4 * float(np.abs(g_ref).max()) + 1e-6
