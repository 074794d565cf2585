#!/usr/bin/env python3
"""Golden vectors for BASELINE config 2: MNIST-shaped LeNet, MC-Dropout, int8 A7/W8.  RUNS ONLY IN THE BUILD CONTAINER.

Imports the real reference (ref_shim), builds `conv_lenet_mc`, prepare_model (QAT) -> calibration forwards -> convert,
then runs the reference's stochastic forward with the Bernoulli masks INJECTED from the build's Philox uniform stream
(Tensor.bernoulli_ patched; draw order = layers.1, layers.4, layers.9) and records every layer output of sample 0 and
the per-sample / mean probabilities.  Output: tests/golden/lenet_mc_a7w8.npz (data only)."""
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

from oracle import oracle as orc  # noqa: E402

SEED, B, S, P = 3, 6, 3, 0.2


def main():
    from src.models import ModelFactory
    import src.quant_utils as qu
    args = types.SimpleNamespace(p=P, activation_precision=7, weight_precision=8, model="conv_lenet_mc", q=True, at=True,
                                 samples=S, task="classification")
    torch.manual_seed(1)
    model = ModelFactory.get_model("conv_lenet_mc", [1, 1, 28, 28], 10, True, args)
    g = torch.Generator().manual_seed(1)
    for m in model.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)):
            fan_in = m.weight[0].numel()
            m.weight.data = torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5
    qu.prepare_model(model, args)
    xcal = torch.rand(32, 1, 28, 28, generator=g)           # MNIST is un-normalised [0,1] (src/data.py:18)
    model.train()
    for _ in range(3):
        model(xcal)
    model.eval()
    with torch.no_grad():
        model(xcal)
    qu.convert(model)
    model.eval()

    state = {}
    for i in (0, 3, 7, 10):
        m = model.layers[i]
        w = m.weight()
        state[f"layers.{i}.weight"] = w.int_repr().numpy()
        state[f"layers.{i}.weight.q_scale"] = np.float64(w.q_scale())
        state[f"layers.{i}.weight.q_zero_point"] = np.int64(w.q_zero_point())
        state[f"layers.{i}.scale"] = np.float64(m.scale)
        state[f"layers.{i}.zero_point"] = np.int64(m.zero_point)
        assert m.bias() is None
    for i in (1, 4, 9):
        m = model.layers[i]
        state[f"layers.{i}.mul_mask.scale"] = np.float64(m.mul_mask.scale)
        state[f"layers.{i}.mul_mask.zero_point"] = np.int64(m.mul_mask.zero_point)
        state[f"layers.{i}.p"] = m.p.detach().numpy()
        state[f"layers.{i}.multiplier"] = m.multiplier.detach().numpy()
    state["quant.scale"] = model.quant.scale.numpy()
    state["quant.zero_point"] = model.quant.zero_point.numpy()

    x = torch.rand(B, 1, 28, 28, generator=g)
    shapes = [(B, 20), (B, 50), (B, 500)]
    keep = np.float32(1.0) - np.float32(P)
    queue = []
    orig = torch.Tensor.bernoulli_

    def bernoulli_(t, p=0.5, *, generator=None):
        m = queue.pop(0)
        assert tuple(t.shape) == m.shape
        t.copy_(torch.from_numpy(m))
        return t

    rec, hooks = {}, []

    def mk(name):
        def hook(_m, _i, o):
            a = o.int_repr().numpy()
            a = np.ascontiguousarray(a.transpose(0, 2, 3, 1)) if a.ndim == 4 else a
            rec[name + ".out"] = np.minimum(a, 127).astype(np.uint8)
        return hook

    for i in (0, 1, 2, 3, 4, 5, 7, 9, 10):
        hooks.append(model.layers[i].register_forward_hook(mk(f"layers.{i}")))
    hooks.append(model.quant.register_forward_hook(mk("quant")))
    probs = []
    torch.Tensor.bernoulli_ = bernoulli_
    try:
        with torch.no_grad():
            for s in range(S):
                queue[:] = [(orc.fill_uniform(int(np.prod(sh)), SEED, di, s) < keep).astype(np.float32).reshape(sh) for di, sh in enumerate(shapes)]
                probs.append(model(x).numpy().copy())
                if s == 0:
                    for h in hooks:
                        h.remove()
    finally:
        torch.Tensor.bernoulli_ = orig
    probs = np.stack(probs)
    mean = torch.stack([torch.from_numpy(p) for p in probs], dim=1).mean(dim=1).numpy()

    net = orc.Int8LeNetMCOracle(state, 7)
    orec = {}
    p0 = net.forward(x.numpy(), SEED, 0, record=orec)
    bad = 0
    for k, v in rec.items():
        o = orec[k]
        if k == "layers.5.out" or k == "layers.2.out":
            pass
        bad += int((o.reshape(v.shape) != v).sum())
    rel = np.abs(p0 - probs[0]).max() / probs[0].max()
    print(f"oracle vs reference (LeNet MC-Dropout): {bad} mismatching integer elements; probs max rel err {rel:.2e}")
    assert bad == 0 and rel < 1e-5
    out = {"x": x.numpy(), "probs": probs, "mean_probs": mean, "meta.philox_seed": np.int64(SEED), "meta.a_bits": np.int64(7),
           "meta.w_bits": np.int64(8), "meta.p": np.float32(P)}
    out.update({"state/" + k: v for k, v in state.items()})
    out.update({"rec/" + k: v for k, v in rec.items()})
    path = os.path.join(HERE, "lenet_mc_a7w8.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()
