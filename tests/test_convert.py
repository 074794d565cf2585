"""QAT -> int8 conversion (quantised_bayesian_nets_amd/convert.py) against what the reference's own convert() produced
for the same QAT layers (tests/golden/make_golden_convert.py).  CPU-only host logic; everything must match exactly."""
import os
import types

import numpy as np
import pytest
import torch

from quantised_bayesian_nets_amd import convert as cv

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "convert_layers_a7w8.npz")


def _layers():
    d = np.load(GOLDEN)
    names = sorted({"/".join(k.split("/")[:2]) for k in d.files})
    return d, names


def _observer(a):
    return cv.ObserverState(np.float32(a[0]), np.float32(a[1]), int(a[2]), int(a[3]))


@pytest.mark.parametrize("idx", range(10))
def test_convert_layer_matches_reference(idx):
    d, names = _layers()
    p = names[idx] + "/"
    bn = None
    if p + "bn.running_mean" in d.files:
        bn = dict(running_mean=d[p + "bn.running_mean"], running_var=d[p + "bn.running_var"], eps=float(d[p + "bn.eps"]),
                  weight=d[p + "bn.weight"], bias=d[p + "bn.bias"])
    bias = d[p + "bias"] if p + "bias" in d.files else None
    got = cv.convert_layer(d[p + "mu"], d[p + "rho"], bias, _observer(d[p + "obs.weight"]), _observer(d[p + "obs.std"]),
                           _observer(d[p + "obs.act"]), _observer(d[p + "obs.add"]), _observer(d[p + "obs.mul"]), bn)
    expect = {k[len(p + "expect/"):]: d[k] for k in d.files if k.startswith(p + "expect/")}
    assert set(expect) == set(got), (sorted(expect), sorted(got))
    for k, v in expect.items():
        g = np.asarray(got[k])
        if np.issubdtype(np.asarray(v).dtype, np.integer) or v.dtype == np.int8:
            assert np.array_equal(g, v), (p, k)
        else:
            assert np.array_equal(g.astype(np.float64), np.asarray(v).astype(np.float64)), (p, k, g, v)


def test_convert_qat_module_duck_typing():
    """convert_qat_module reads a reference-style QAT module object (attributes only)."""
    d, names = _layers()
    p = [n for n in names if n.endswith("layers.3.0.stem.3") and n.startswith("w8")][0] + "/"

    def fq(a):
        o = types.SimpleNamespace(min_val=torch.tensor(np.float32(a[0])), max_val=torch.tensor(np.float32(a[1])), quant_min=int(a[2]), quant_max=int(a[3]))
        return types.SimpleNamespace(activation_post_process=o, quant_min=int(a[2]), quant_max=int(a[3]))

    bn = types.SimpleNamespace(running_mean=torch.from_numpy(d[p + "bn.running_mean"]), running_var=torch.from_numpy(d[p + "bn.running_var"]),
                               eps=float(d[p + "bn.eps"]), weight=torch.from_numpy(d[p + "bn.weight"]), bias=torch.from_numpy(d[p + "bn.bias"]))
    mod = types.SimpleNamespace(weight=torch.from_numpy(d[p + "mu"]), std=torch.from_numpy(d[p + "rho"]), bias=None, bn=bn,
                                weight_fake_quant=fq(d[p + "obs.weight"]), std_fake_quant=fq(d[p + "obs.std"]),
                                activation_post_process=fq(d[p + "obs.act"]),
                                add_weight=types.SimpleNamespace(activation_post_process=fq(d[p + "obs.add"])),
                                mul_noise=types.SimpleNamespace(activation_post_process=fq(d[p + "obs.mul"])))
    got = cv.convert_qat_module(mod)
    assert np.array_equal(got["weight"], d[p + "expect/weight"]) and np.array_equal(got["std"], d[p + "expect/std"])
    assert got["zero_point"] == int(d[p + "expect/zero_point"])


def test_from_float_accepts_a_qat_module():
    """Conv2d.from_float on a reference-style QAT module: the reference's own swap seam (quant_utils.py:62-99)."""
    import quantised_bayesian_nets_amd as q
    d, names = _layers()
    p = [n for n in names if n.endswith("layers.4.0.shortcut.0") and n.startswith("w8")][0] + "/"

    def fq(a):
        o = types.SimpleNamespace(min_val=torch.tensor(np.float32(a[0])), max_val=torch.tensor(np.float32(a[1])), quant_min=int(a[2]), quant_max=int(a[3]))
        return types.SimpleNamespace(activation_post_process=o, quant_min=int(a[2]), quant_max=int(a[3]))

    bn = types.SimpleNamespace(running_mean=torch.from_numpy(d[p + "bn.running_mean"]), running_var=torch.from_numpy(d[p + "bn.running_var"]),
                               eps=float(d[p + "bn.eps"]), weight=torch.from_numpy(d[p + "bn.weight"]), bias=torch.from_numpy(d[p + "bn.bias"]))
    mod = types.SimpleNamespace(weight=torch.from_numpy(d[p + "mu"]), std=torch.from_numpy(d[p + "rho"]), bias=None, bn=bn,
                                weight_fake_quant=fq(d[p + "obs.weight"]), std_fake_quant=fq(d[p + "obs.std"]),
                                activation_post_process=fq(d[p + "obs.act"]),
                                add_weight=types.SimpleNamespace(activation_post_process=fq(d[p + "obs.add"])),
                                mul_noise=types.SimpleNamespace(activation_post_process=fq(d[p + "obs.mul"])),
                                in_channels=24, out_channels=48, kernel_size=(1, 1), stride=(2, 2), padding=(0, 0), dilation=(1, 1),
                                groups=1, padding_mode="zeros", args=types.SimpleNamespace(activation_precision=7, weight_precision=8))
    layer = q.Conv2d.from_float(mod)
    assert np.array_equal(layer.weight.int_repr(), d[p + "expect/weight"])
    assert layer.zero_point == int(d[p + "expect/zero_point"]) and abs(layer.scale - float(d[p + "expect/scale"])) == 0
    assert np.array_equal(layer.bias_.numpy(), d[p + "expect/bias_"])
