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


def relu_flips(eng, B, T, Lq, predictor='transformer'):
    """(number of ReLU decisions on which the HIP forward and the oracle forward disagree, the HIP path's decisions as a list
    of bool tensors in the oracle's call order).  Sites: conv layers of every encoder application + both span heads.  Call
    after O.record_relu_signs(); O.forward(...) and an eng.forward of the same inputs.
    A disagreement needs a pre-activation within the ~1e-5 forward noise of zero, but it changes the gradient of that element
    discontinuously (the tolerances of SURVEY 8c assume none).  Tests that find one re-run the oracle on the GPU path's branch
    (O.force_relu_signs) and keep the strict gate."""
    from oracle import vslnet_oracle as O
    sites = list(O.RELU_SIGNS)
    masks = hip_relu_masks(eng, B, T, Lq, predictor)
    assert [s for s, _ in sites] == ['conv'] * (len(masks) - 2) + ['head_start', 'head_end'], [s for s, _ in sites]
    flips = sum(int((m != sg).sum()) for m, (_, sg) in zip(masks, sites))
    return flips, masks


# ---------------------------------------------------------------------------------------------------------------------------
# How far from zero may a ReLU pre-activation be when two correct fp32 implementations disagree about its sign?
#
# z = sum_k a_k w_k + b is a dot product of length K (128 in the conv layers: a = depthwise7(LN(x)); 256 in the span heads).  Two
# implementations that add the same products in different orders differ by at most gamma_K * S each from the exact sum, S = sum_k
# |a_k w_k| + |b|, gamma_K = K u / (1 - K u), u = 2^-24 (Higham, Accuracy and Stability of Numerical Algorithms, (3.5)) -- and they do
# not see the same a either: a itself is the end of a chain of fp32 reductions (7 taps, the two LayerNorm moments over 128 channels, the
# dot product that produced x: 128 .. Dv terms, the residual stream of up to 4 layers x 4 encoder applications), each contributing an
# error of the same form relative to ITS OWN absolute sum, which the contraction with w carries into z.  In units of u * S:
#       |z_1 - z_2| / (u S)  <=  2 (K + 7 + 2 * 128 + K_in) * A,        K_in <= max(128, Dv) = 4096 at configs[2],
# A = amplification of the inputs' relative error through LN (1 / sigma of a row, O(1) here).  That worst case (~1e4) is never approached:
# rounding errors add like a random walk, sqrt instead of linear, so the expected scale is 2 sqrt(K + 7 + 256 + K_in) ~ 40 .. 130.  The
# gate: a decision may be overridden only where |z| <= RELU_NOISE_KAPPA * u * S.  Round 3 set KAPPA = 150 (the random-walk scale with
# Dv = 4096, rounded up); measured over the whole GPU suite (`pytest -s` prints every value) the largest override is 3.9 x (u * S) = |z| 2.0e-7
# with the split-bf16 kernels, 3.0 x (1.7e-7) with the fp32-input MFMA kernels of round 2 -- so round 4 sets KAPPA = 16: four times the
# largest value ever seen, and a kernel regression that moved pre-activations by 5 x today's noise now fails the gradient-parity tests
# instead of being absorbed by forced branches (VERDICT r3 / ADVICE r3).  The margin is printed with every check.  The absolute deviation is
# still printed but not asserted: the old gate (2e-5, not tied to S) vetoed a legitimate re-ordering in round 2 (the 2-K-group
# VisualProjection: 3.3e-5 on a row with a large S) while being 100x looser than needed on ordinary rows.
# ---------------------------------------------------------------------------------------------------------------------------
RELU_NOISE_KAPPA = 16.0
RELU_MAX_OVERRIDES_FRAC = 1e-4     # of a site's decisions
RELU_MAX_OVERRIDES_ABS = 2         # ... but a tiny site may have one or two


def assert_forced_relu_inside_noise(O, context=None):
    """After a forward with forced ReLU branches: every overridden pre-activation must lie inside its own fp32 noise."""
    r = O.forced_relu_noise_ratio()
    print('[relu-noise] %s: largest overridden pre-activation %.3e = %.1f x (u * S), gate %.0f (margin %.1f x)'
          % (context if isinstance(context, str) else '', O.forced_relu_deviation(), r, RELU_NOISE_KAPPA, RELU_NOISE_KAPPA / max(r, 1e-9)))
    assert r <= RELU_NOISE_KAPPA, (context, 'forced ReLU branch at %.1f x (u * S), |z| up to %.3e' % (r, O.forced_relu_deviation()))
    # ... and there must be FEW of them: a pre-activation lands inside its noise band with probability ~ KAPPA * u * S / spread(z) ~ 1e-6, so a
    # site of n decisions may see max(RELU_MAX_OVERRIDES_ABS, RELU_MAX_OVERRIDES_FRAC * n) overrides -- a kernel that flipped thousands of
    # near-zero decisions inside the band would pass the magnitude gate above
    for site, (nbad, n) in enumerate(O.forced_relu_counts()):
        assert nbad <= max(RELU_MAX_OVERRIDES_ABS, RELU_MAX_OVERRIDES_FRAC * n), (context, 'ReLU site %d: %d of %d decisions overridden' % (site, nbad, n))


def hip_relu_masks(eng, B, T, Lq, predictor='transformer'):
    """The ReLU decisions the last HIP forward saved, as bool tensors in the oracle's call order."""
    import torch
    encs = [('venc', T)] * 4 + [('qenc', Lq)] * 4
    if predictor == 'transformer':
        encs += [('p1', T)] * 4 + [('p2', T)] * 4
    masks = []
    for idx, (enc, L) in enumerate(encs):
        words = eng.ws_view('relu_%s_%d' % (enc, idx % 4), (B * L * 4,)).view(torch.int32).cpu().view(B * L, 4)
        masks.append(((words.unsqueeze(-1) >> torch.arange(32, dtype=torch.int32)) & 1).bool().view(B, L, 128))
    for name in ('s', 'e'):
        masks.append(eng.ws_view('hid_' + name, (B, T, 128)).cpu() > 0)
    return masks


# ---------------------------------------------------------------------------------------------------------------------------
# the HIP path's dropout masks, recomputed on the host (vslnet_amd/csrc/common.hpp: drop_keep_scale, api.hip: Ctx::drop)
# ---------------------------------------------------------------------------------------------------------------------------
_M32 = np.uint64(0xFFFFFFFF)


def _fmix32(h):
    h = h & _M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    h ^= h >> np.uint64(16)
    return h


def hip_site_seed(seed, site):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    x = (seed & 0xFFFFFFFF) ^ (((seed >> 32) * 0x9E3779B1) & 0xFFFFFFFF) ^ ((site * 0x85EBCA77 + 0x165667B1) & 0xFFFFFFFF)
    return int(_fmix32(np.array([x], dtype=np.uint64))[0])


def hip_site_key(seed, site):
    """second per-site word (api.hip Ctx::drop): xor-ed in between the two multiply rounds of the element hash"""
    k = np.array([hip_site_seed(seed, site) ^ 0x68E31DA4], dtype=np.uint64)
    k ^= k >> np.uint64(15)
    k = (k * np.uint64(0x2C1B3C6D)) & _M32
    k ^= k >> np.uint64(12)
    k = (k * np.uint64(0x297A2D39)) & _M32
    k ^= k >> np.uint64(15)
    return int(k[0])


def _drop_hash(idx, seed, key):
    """common.hpp drop_hash"""
    h = ((idx * np.uint64(0x9E3779B1)) + np.uint64(seed)) & _M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h ^= (h >> np.uint64(13)) ^ np.uint64(key)
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    return h                                             # (no final xor-shift: the decision is a compare against p * 2^32)


# site ids in the order the oracle reaches its dropout calls (api.hip: SITE_* = 64.., encoder pass `app` * 16 + 0..8)
HIP_DROP_SITES = [64, 65, 66] + list(range(0, 9)) + list(range(16, 25)) + [67, 68] + list(range(32, 41)) + list(range(48, 57))


def hip_dropout(seed):
    """-> callable for O.force_dropout: multiplier of element i (row-major index in the tensor the reference applies
    nn.Dropout to) at the k-th dropout call = keep(drop_hash(i, seed_site, key_site)) / (1 - p)."""
    def mask(call_no, shape, p):
        site = HIP_DROP_SITES[call_no]
        n = int(np.prod(shape))
        idx = np.arange(n, dtype=np.uint64)
        if site < 64 and site % 16 == 5 and shape[-1] > 256:
            # the attention probabilities of L > 256 (k_attn_fwd / k_attn_bwd_long, common.hpp drop_hash_odd): keys 2 j, 2 j + 1 of a (b, h, q) row
            # share drop_hash(row * ceil(L / 2) + j); the odd key compares the hash rotated by 16 bits
            L = np.uint64(shape[-1])
            row, key = idx // L, idx % L
            h = _drop_hash((row * ((L + np.uint64(1)) // np.uint64(2)) + key // np.uint64(2)) & _M32, hip_site_seed(seed, site), hip_site_key(seed, site))
            h = np.where(key % np.uint64(2) == 1, ((h >> np.uint64(16)) | (h << np.uint64(16))) & _M32, h)
        else:
            h = _drop_hash(idx, hip_site_seed(seed, site), hip_site_key(seed, site))
        thresh = np.uint64(min(4294967295.0, float(np.float32(p)) * 4294967296.0))
        scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
        return torch.from_numpy(np.where(h >= thresh, scale, np.float32(0.0)).astype(np.float32).reshape(shape))
    return mask
