#!/usr/bin/env python3
"""Golden vectors for the QAT -> int8 conversion.  RUNS ONLY IN THE BUILD CONTAINER.

Rebuilds the same reference `conv_resnet_bbb` as make_golden.py up to the calibrated QAT model, dumps -- for a few real
layers of different kinds -- the QAT-side inputs of the conversion (mu, rho, bias, BatchNorm statistics, the min/max state
of the five observers), then lets the REFERENCE convert the model and records the converted layer state.
Output: tests/golden/convert_layers_a7w8.npz (data only)."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402

ref_shim.install()
import torch  # noqa: E402

LAYERS = ["layers.0", "layers.3.0.stem.3", "layers.4.0.shortcut.0", "layers.4.0.stem.0", "layers.9"]


def obs(fq):
    o = fq.activation_post_process
    return np.array([float(o.min_val), float(o.max_val), fq.quant_min, fq.quant_max], np.float64)


def main():
    import make_golden as mg
    import src.quant_utils as qu
    from src.models import ModelFactory
    from src.models.stochastic.bbb.conv import Conv2d as Conv2dBBB
    from src.models.stochastic.bbb.linear import Linear as LinearBBB
    for w_bits in (8, 4):
        args = types.SimpleNamespace(sigma_prior=-2.0, activation_precision=7, weight_precision=w_bits, model="conv_resnet_bbb",
                                     q=True, at=True, samples=4, task="classification")
        torch.manual_seed(mg.PARAM_SEED)
        model = ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, args)
        g = torch.Generator().manual_seed(mg.PARAM_SEED)
        for m in model.modules():
            if isinstance(m, (Conv2dBBB, LinearBBB)):
                fan_in = m.weight[0].numel()
                m.weight.data = torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5
                m.std.data.fill_(-3.0)
            elif isinstance(m, torch.nn.BatchNorm2d):
                m.weight.data = torch.rand(m.weight.shape, generator=g) + 0.5
                m.bias.data.zero_()
        qu.prepare_model(model, args)
        gx = torch.Generator().manual_seed(mg.INPUT_SEED)
        x_cal = torch.randn(32, 3, 32, 32, generator=gx)
        torch.manual_seed(mg.PARAM_SEED + 100)
        model.train(); model(x_cal); model.eval()
        with torch.no_grad():
            for _ in range(3):
                model(x_cal)
        out = {}
        mods = dict(model.named_modules())
        for n in LAYERS:
            m = mods[n]
            p = f"w{w_bits}/{n}/"
            out[p + "kind"] = np.array(type(m).__name__)
            out[p + "mu"] = m.weight.detach().numpy().copy()
            out[p + "rho"] = m.std.detach().numpy().copy()
            if m.bias is not None:
                out[p + "bias"] = m.bias.detach().numpy().copy()
            if hasattr(m, "bn"):
                out[p + "bn.running_mean"] = m.bn.running_mean.numpy().copy()
                out[p + "bn.running_var"] = m.bn.running_var.numpy().copy()
                out[p + "bn.weight"] = m.bn.weight.detach().numpy().copy()
                out[p + "bn.bias"] = m.bn.bias.detach().numpy().copy()
                out[p + "bn.eps"] = np.float64(m.bn.eps)
            out[p + "obs.weight"] = obs(m.weight_fake_quant)
            out[p + "obs.std"] = obs(m.std_fake_quant)
            out[p + "obs.act"] = obs(m.activation_post_process)
            out[p + "obs.add"] = obs(m.add_weight.activation_post_process)
            out[p + "obs.mul"] = obs(m.mul_noise.activation_post_process)
        qu.convert(model)
        st = mg.flat_state(model)
        for n in LAYERS:
            for k, v in st.items():
                if k.startswith(n + ".") and not k.startswith(n + ".std_prior"):
                    out[f"w{w_bits}/{n}/expect/" + k[len(n) + 1:]] = v
        if w_bits == 8:
            big = dict(out)
    big.update(out)
    path = os.path.join(HERE, "convert_layers_a7w8.npz")
    np.savez_compressed(path, **big)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()
