#!/usr/bin/env python3
"""Golden vectors for BASELINE config 0: UCI-regression-shaped 3x100 MLP, Bayes-by-backprop fp32, 10 MC samples.
RUNS ONLY IN THE BUILD CONTAINER.  Imports the real reference (`linear_bbb`, float, eval mode: bbb/linear.py:42-50),
injects the build's Philox eps into Tensor.normal_ (draw order layers.0, layers.2, layers.4, mu, log_var) and records
per-sample (mu, var) and the reduction of experiments/utils.py:348-353.  Output: tests/golden/mlp_bbb_f32.npz (in_dim 13, 1000 rows, 10
samples: BASELINE config 0 as the benchmark runs it) and mlp_bbb_f32_in{1,4,6,8,11}.npz (SURVEY 8(d) C1's other input widths, 250 rows,
4 samples).  Every fixture also records the reference's distance from itself on another CPU code path (altref.py): `refspread.*`."""
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

import altref  # noqa: E402
from oracle import oracle as orc  # noqa: E402

SEED = 3


def main(B, S, IN_DIM, out_name):
    from src.models import ModelFactory
    args = types.SimpleNamespace(sigma_prior=-2.0, model="linear_bbb", q=False, at=False, samples=S, task="regression")
    torch.manual_seed(1)
    model = ModelFactory.get_model("linear_bbb", [IN_DIM], 1, False, args)
    g = torch.Generator().manual_seed(1)
    names = ["layers.0", "layers.2", "layers.4", "mu", "log_var"]
    mods = dict(model.named_modules())
    for n in names:
        m = mods[n]
        fan_in = m.weight.shape[1]
        m.weight.data = torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5     # SURVEY 8(d) config C1 init
        m.std.data.fill_(-3.0)
        m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.1
    model.eval()
    x = torch.randn(B, IN_DIM, generator=g)
    state = {}
    for n in names:
        m = mods[n]
        state[n + ".weight"] = m.weight.detach().numpy().copy()
        state[n + ".std"] = m.std.detach().numpy().copy()
        state[n + ".bias"] = m.bias.detach().numpy().copy()
    shapes = [tuple(mods[n].weight.shape) for n in names]
    queue = []
    orig = torch.Tensor.normal_

    def normal_(t, mean=0, std=1, *, generator=None):
        e = queue.pop(0)
        assert tuple(t.shape) == e.shape
        t.copy_(torch.from_numpy(e))
        return t

    mus, vars_ = [], []
    torch.Tensor.normal_ = normal_
    try:
        with torch.no_grad():
            for s in range(S):
                queue[:] = [orc.fill_normal(int(np.prod(sh)), SEED, lid, s).reshape(sh) for lid, sh in enumerate(shapes)]
                mu, var = model(x)
                mus.append(mu.numpy().copy()); vars_.append(var.numpy().copy())
    finally:
        torch.Tensor.normal_ = orig
    mu_t = [torch.from_numpy(m) for m in mus]
    var_t = [torch.from_numpy(v) for v in vars_]
    mean = torch.stack(mu_t, dim=1).mean(dim=1)                                           # experiments/utils.py:351
    var = torch.stack(mu_t, dim=1).var(dim=1) + torch.stack(var_t, dim=1).mean(dim=1)     # :352
    if altref.alt_out_path():                  # this process is the reference on the other code path: outputs only
        np.savez(altref.alt_out_path(), mu=np.stack(mus), var=np.stack(vars_))
        return
    alt = altref.run_alt(os.path.abspath(__file__), out_name)
    sp_mu, sp_var = altref.spread(np.stack(mus), alt["mu"]), altref.spread(np.stack(vars_), alt["var"])
    print(f"in_dim {IN_DIM}: reference AVX-512 vs reference AVX2: mu max abs {sp_mu[0]:.2e} (range {np.abs(np.stack(mus)).max():.2f}), var max rel {sp_var[1]:.2e}")
    net = orc.F32MLPOracle(state)
    o_mu, o_var = net.forward(x.numpy(), SEED, 0)
    rel = max(np.abs(o_mu - mus[0]).max() / np.abs(mus[0]).max(), np.abs(o_var - vars_[0]).max() / np.abs(vars_[0]).max())
    print(f"oracle vs reference (float BBB MLP): max rel err {rel:.2e}")
    assert rel < 1e-5
    out = {"x": x.numpy(), "mu": np.stack(mus), "var": np.stack(vars_), "mean": mean.numpy(), "pred_var": var.numpy(),
           "meta.philox_seed": np.int64(SEED), "meta.in_dim": np.int64(IN_DIM),
           "refspread.mu_abs": np.float64(sp_mu[0]), "refspread.var_rel": np.float64(sp_var[1])}
    out.update({"state/" + k: v for k, v in state.items()})
    path = os.path.join(HERE, out_name)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")


CASES = [(1000, 10, 13, "mlp_bbb_f32.npz")] + [(250, 4, d, "mlp_bbb_f32_in%d.npz" % d) for d in (1, 4, 6, 8, 11)]

if __name__ == "__main__":
    for case in CASES:
        if altref.alt_tag() in (None, case[3]):
            main(*case)
