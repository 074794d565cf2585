#!/usr/bin/env python3
"""Golden vectors for the small int8 Bayes-by-backprop graphs: `conv_lenet_bbb` (models_bbb.py:98-143) and
`linear_bbb` with q=True (models_bbb.py:32-95).  RUNS ONLY IN THE BUILD CONTAINER.  Same recipe as make_golden.py:
prepare_model -> 1 train + 3 eval calibration forwards -> convert -> reference forward with the build's Philox eps
injected.  Outputs: tests/golden/lenet_bbb_a7w8.npz, tests/golden/mlp_bbb_a7w8.npz (data only)."""
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

import make_golden as mg  # noqa: E402
from oracle import oracle as orc  # noqa: E402

SEED = 3


def build(model_name, input_size, output_size, task, xcal):
    from src.models import ModelFactory
    import src.quant_utils as qu
    from src.models.stochastic.bbb.conv import Conv2d as Conv2dBBB
    from src.models.stochastic.bbb.linear import Linear as LinearBBB
    args = types.SimpleNamespace(sigma_prior=-2.0, activation_precision=7, weight_precision=8, model=model_name, q=True, at=True,
                                 samples=3, task=task)
    torch.manual_seed(1)
    model = ModelFactory.get_model(model_name, input_size, output_size, True, args)
    g = torch.Generator().manual_seed(1)
    for m in model.modules():
        if isinstance(m, (Conv2dBBB, LinearBBB)):
            fan_in = m.weight[0].numel()
            m.weight.data = torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5
            m.std.data.fill_(-3.0)
            if m.bias is not None:
                m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.1
    qu.prepare_model(model, args)
    torch.manual_seed(101)
    model.train(); model(xcal); model.eval()
    with torch.no_grad():
        for _ in range(3):
            model(xcal)
    qu.convert(model)
    model.eval()
    return model, args


def run(model, x, names, S, rec_names):
    mods = dict(model.named_modules())
    shapes = [tuple(mods[n].weight.shape) for n in names]
    rec, hooks = {}, []

    def mk(name):
        def hook(_m, _i, o):
            a = o.int_repr().numpy() if o.is_quantized else o.numpy()
            a = np.ascontiguousarray(a.transpose(0, 2, 3, 1)) if a.ndim == 4 else a
            rec[name + ".out"] = np.minimum(a, 127).astype(np.uint8)
        return hook

    for n in rec_names:
        hooks.append(mods[n].register_forward_hook(mk(n)))
    inj = mg.Injector(shapes)
    outs = []
    with torch.no_grad():
        for s in range(S):
            inj.arm(SEED, s)
            with inj:
                outs.append(model(x))
            if s == 0:
                for h in hooks:
                    h.remove()
    return outs, rec


def main():
    g = torch.Generator().manual_seed(5)
    # ---- LeNet BBB int8
    model, args = build("conv_lenet_bbb", [1, 1, 28, 28], 10, "classification", torch.rand(32, 1, 28, 28, generator=g))
    names = ["layers.0", "layers.2", "layers.5", "layers.7"]
    x = torch.rand(4, 1, 28, 28, generator=g)
    outs, rec = run(model, x, names, 3, names + ["quant", "layers.1", "layers.3"])
    probs = np.stack([o.numpy() for o in outs])
    st = mg.flat_state(model)
    net = orc.Int8LeNetBBBOracle(st, 7, 8)
    orec = {}
    p0 = net.forward(x.numpy(), SEED, 0, record=orec)
    bad = sum(int((orec[k].reshape(v.shape) != v).sum()) for k, v in rec.items())
    rel = np.abs(p0 - probs[0]).max() / probs[0].max()
    print(f"oracle vs reference (LeNet BBB int8): {bad} mismatching integer elements; probs max rel err {rel:.2e}")
    assert bad == 0 and rel < 1e-5
    out = {"x": x.numpy(), "probs": probs, "mean_probs": probs.mean(0), "meta.philox_seed": np.int64(SEED)}
    out.update({"state/" + k: v for k, v in st.items()})
    out.update({"rec/" + k: v for k, v in rec.items()})
    np.savez_compressed(os.path.join(HERE, "lenet_bbb_a7w8.npz"), **out)
    print("wrote lenet_bbb_a7w8.npz", os.path.getsize(os.path.join(HERE, "lenet_bbb_a7w8.npz")) / 1e6, "MB")

    # ---- MLP BBB int8 (regression heads)
    model, args = build("linear_bbb", [13], 1, "regression", torch.randn(64, 13, generator=g))
    names = ["layers.0", "layers.2", "layers.4", "mu", "log_var"]
    x = torch.randn(200, 13, generator=g)
    outs, rec = run(model, x, names, 3, names + ["quant"])
    mu = np.stack([o[0].numpy() for o in outs]); var = np.stack([o[1].numpy() for o in outs])
    st = mg.flat_state(model)
    net = orc.Int8MLPBBBOracle(st, 7, 8)
    orec = {}
    m0, v0 = net.forward(x.numpy(), SEED, 0, record=orec)
    bad = sum(int((orec[k].reshape(v.shape) != v).sum()) for k, v in rec.items())
    rel = max(np.abs(m0 - mu[0]).max() / max(np.abs(mu[0]).max(), 1e-9), np.abs(v0 - var[0]).max() / np.abs(var[0]).max())
    print(f"oracle vs reference (MLP BBB int8): {bad} mismatching integer elements; max rel err {rel:.2e}")
    assert bad == 0 and rel < 1e-5
    out = {"x": x.numpy(), "mu": mu, "var": var, "meta.philox_seed": np.int64(SEED)}
    out.update({"state/" + k: v for k, v in st.items()})
    out.update({"rec/" + k: v for k, v in rec.items()})
    np.savez_compressed(os.path.join(HERE, "mlp_bbb_a7w8.npz"), **out)
    print("wrote mlp_bbb_a7w8.npz", os.path.getsize(os.path.join(HERE, "mlp_bbb_a7w8.npz")) / 1e6, "MB")


if __name__ == "__main__":
    main()
