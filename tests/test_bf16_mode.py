"""bf16 THROUGHPUT mode (BASELINE configs[1] says "bf16"; the reference is fp32 end to end): bfloat16 features in HBM and a
bf16-MFMA VisualProjection (bf16 features x bf16-rounded weight, fp32 accumulation), everything downstream fp32.
Tolerances of this mode, stated here and nowhere claimed for the parity path:
  * against the fp32 oracle evaluated on the SAME rounded inputs (features and video_affine weight rounded to bf16): the products
    of two bf16 numbers are exact in fp32, so only the summation order differs -> the parity gates apply (logits 1e-4, gradients
    1e-4 * |g|inf + 1e-6);
  * against the fp32 oracle on the unrounded inputs: logits within 8e-2 (rounding the features alone moves logits by
    2.4e-2 at the real shape)."""
import numpy as np
import pytest
import torch

from oracle import vslnet_oracle as O
from tests.helpers import assert_forced_relu_inside_noise
from tests.helpers import relu_flips

pytestmark = pytest.mark.gpu


def _round_bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)
