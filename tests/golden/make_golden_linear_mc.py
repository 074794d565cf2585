#!/usr/bin/env python3
"""Golden vectors for `linear_mc` (the MC-Dropout regression MLP), int8 A7/W8.  RUNS ONLY IN THE BUILD CONTAINER.

Imports the real reference (ref_shim), builds `linear_mc` (mcdropout/models_mc.py:10-73; factory name
src/models/__init__.py:25-26), prepare_model (QAT) -> calibration forwards -> convert, then runs the reference's
stochastic forward with the Bernoulli masks INJECTED from the build's Philox uniform stream (Tensor.bernoulli_ patched;
draw order = layers.2, layers.5, mu.0, log_var.0) and records every layer output of sample 0, the per-sample (mu, var)
and the regression reduction of experiments/utils.py:348-353.  Output: tests/golden/mlp_mc_a7w8.npz (data only)."""
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

SEED, B, S, P, IN_DIM = 3, 200, 4, 0.2, 13
LINEARS = ("layers.0", "layers.3", "layers.6", "mu.1", "log_var.1")
DROPS = ("layers.2", "layers.5", "mu.0", "log_var.0")


def main():
    from src.models import ModelFactory
    import src.quant_utils as qu
    args = types.SimpleNamespace(p=P, activation_precision=7, weight_precision=8, model="linear_mc", q=True, at=True,
                                 samples=S, task="regression")
    torch.manual_seed(1)
    model = ModelFactory.get_model("linear_mc", [IN_DIM], 1, True, args)
    g = torch.Generator().manual_seed(1)
    for m in model.modules():
        if isinstance(m, torch.nn.Linear):
            fan_in = m.weight.shape[1]
            m.weight.data = torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5     # SURVEY 8(d) config C1 init
            m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.1
    qu.prepare_model(model, args)
    xcal = torch.randn(256, IN_DIM, generator=g)
    torch.manual_seed(101)
    model.train()
    for _ in range(3):
        model(xcal)
    model.eval()
    with torch.no_grad():
        model(xcal)
    qu.convert(model)
    model.eval()

    mods = dict(model.named_modules())
    state = {}
    for n in LINEARS:
        m = mods[n]
        w = m.weight()
        state[n + ".weight"] = w.int_repr().numpy()
        state[n + ".weight.q_scale"] = np.float64(w.q_scale())
        state[n + ".weight.q_zero_point"] = np.int64(w.q_zero_point())
        state[n + ".bias"] = m.bias().detach().numpy().copy()
        state[n + ".scale"] = np.float64(m.scale)
        state[n + ".zero_point"] = np.int64(m.zero_point)
    for n in DROPS:
        m = mods[n]
        state[n + ".mul_mask.scale"] = np.float64(m.mul_mask.scale)
        state[n + ".mul_mask.zero_point"] = np.int64(m.mul_mask.zero_point)
        state[n + ".p"] = m.p.detach().numpy()
        state[n + ".multiplier"] = m.multiplier.detach().numpy()
    state["quant.scale"] = model.quant.scale.numpy()
    state["quant.zero_point"] = model.quant.zero_point.numpy()

    x = torch.randn(B, IN_DIM, generator=g)
    shapes = [(B, 100)] * 4
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
            rec[name + ".out"] = np.minimum(o.int_repr().numpy(), 127).astype(np.uint8)     # clamp_activation follows every module
        return hook

    for n in LINEARS + DROPS:
        hooks.append(mods[n].register_forward_hook(mk(n)))
    hooks.append(model.quant.register_forward_hook(mk("quant")))
    mus, vars_ = [], []
    torch.Tensor.bernoulli_ = bernoulli_
    try:
        with torch.no_grad():
            for s in range(S):
                queue[:] = [(orc.fill_uniform(int(np.prod(sh)), SEED, di, s) < keep).astype(np.float32).reshape(sh) for di, sh in enumerate(shapes)]
                mu, var = model(x)
                assert not queue
                mus.append(mu.numpy().copy()); vars_.append(var.numpy().copy())
                if s == 0:
                    for h in hooks:
                        h.remove()
    finally:
        torch.Tensor.bernoulli_ = orig
    mu_t = [torch.from_numpy(m) for m in mus]
    var_t = [torch.from_numpy(v) for v in vars_]
    mean = torch.stack(mu_t, dim=1).mean(dim=1)                                           # experiments/utils.py:351
    pvar = torch.stack(mu_t, dim=1).var(dim=1) + torch.stack(var_t, dim=1).mean(dim=1)    # :352

    net = orc.Int8MLPMCOracle(state, 7)
    orec = {}
    m0, v0 = net.forward(x.numpy(), SEED, 0, record=orec)
    bad = sum(int((orec[k].reshape(v.shape) != v).sum()) for k, v in rec.items())
    rel = max(np.abs(m0 - mus[0]).max() / max(np.abs(mus[0]).max(), 1e-9), np.abs(v0 - vars_[0]).max() / np.abs(vars_[0]).max())
    kept = [float((rec[n + ".out"] != state[n + ".mul_mask.zero_point"]).mean()) for n in DROPS]
    print(f"oracle vs reference (MLP MC-Dropout int8): {bad} mismatching integer elements over {sum(v.size for v in rec.values())}; "
          f"max rel err {rel:.2e}; non-zero share behind each dropout {kept}")
    assert bad == 0 and rel < 1e-5
    out = {"x": x.numpy(), "mu": np.stack(mus), "var": np.stack(vars_), "mean": mean.numpy(), "pred_var": pvar.numpy(),
           "meta.philox_seed": np.int64(SEED), "meta.in_dim": np.int64(IN_DIM), "meta.p": np.float32(P)}
    out.update({"state/" + k: v for k, v in state.items()})
    out.update({"rec/" + k: v for k, v in rec.items()})
    path = os.path.join(HERE, "mlp_mc_a7w8.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()
