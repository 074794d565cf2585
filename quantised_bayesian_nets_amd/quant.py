"""Quantisation constants and host-side scalar derivations.

Mirrors reference src/utils.py:18-20 (UINT_BOUNDS / INT_BOUNDS), src/quant_utils.py:120-121 (the bit-width
asserts) and src/models/stochastic/bbb/quantized/__init__.py:1-2 (NOISE_SCALE / NOISE_ZERO_POINT).
The float/double mix of each derived scalar is the one ATen / FBGEMM use ("Arithmetic contracts").
"""
import numpy as np

from ._lib import SampleParams

UINT_BOUNDS = {8: [0, 255], 7: [0, 127], 6: [0, 63], 5: [0, 31], 4: [0, 15], 3: [0, 7], 2: [0, 3]}
INT_BOUNDS = {8: [-128, 127], 7: [-64, 63], 6: [-32, 31], 5: [-16, 15], 4: [-8, 7], 3: [-4, 3], 2: [-2, 1]}
NOISE_SCALE = float(0.02362204724)
NOISE_ZERO_POINT = int(0)


def check_bits(args):
    """Same contract as quant_utils.prepare_model (quant_utils.py:120-121)."""
    assert 2 <= args.activation_precision and args.activation_precision <= 7
    assert 2 <= args.weight_precision and args.weight_precision <= 8


def activation_hi(args):
    return UINT_BOUNDS[args.activation_precision][1]


def make_sample_params(s_w, z_w, s_sigma, z_sigma, s_mul, z_mul, s_add, z_add, weight_precision):
    f32 = np.float32
    p = SampleParams()
    p.inv_noise_scale = f32(1.0) / f32(NOISE_SCALE)
    p.mul_multiplier = f32(np.float64(s_sigma) * np.float64(NOISE_SCALE) / np.float64(s_mul))
    p.z_sigma = int(z_sigma)
    p.z_mul = int(z_mul)
    p.s_w = f32(s_w)
    p.nzs_w = f32(-int(z_w)) * f32(s_w)
    p.s_mul = f32(s_mul)
    p.nzs_mul = f32(-int(z_mul)) * f32(s_mul)
    p.inv_s_add = f32(1.0) / f32(s_add)
    p.z_add = int(z_add)
    p.w_lo, p.w_hi = INT_BOUNDS[weight_precision]
    return p
