#!/usr/bin/env python3
"""Golden vectors for the FLOAT MC-Dropout graphs (SURVEY rows a6 / a7 with q=False).  RUNS ONLY IN THE BUILD CONTAINER.
Imports the real reference (`linear_mc`, `conv_lenet_mc`, `conv_resnet_mc` with q=False in eval mode: mcdropout/models_mc.py:10-226,
dropout.py:15-40 with FloatFunctional), injects the build's Philox Bernoulli masks into Tensor.bernoulli_ in draw order and records the
per-sample outputs and the MC reduction (experiments/utils.py:342-355).  Each forward is also run on another CPU code path of the reference (the conv graphs: oneDNN
off; the MLP, whose sgemm does not go through oneDNN: the whole script again on the AVX2 kernels, altref.py): the distance of the
reference from itself is recorded as the tolerance floor (`refspread.*`).
Output: tests/golden/mlp_mc_f32.npz, lenet_mc_f32.npz, resnet_mc_f32.npz (inputs + expected outputs only)."""
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

SEED, P = 3, 0.2


def run(model_name, in_shape, out_size, B, S, out, logit_gain=1.0):
    from src.models import ModelFactory
    from src.models.stochastic.mcdropout.dropout import BernoulliDropout
    regression = model_name == "linear_mc"
    args = types.SimpleNamespace(p=P, model=model_name, q=False, at=False, samples=S, task="regression" if regression else "classification",
                                 activation_precision=7, weight_precision=8)
    torch.manual_seed(1)
    model = ModelFactory.get_model(model_name, in_shape, out_size, False, args)
    g = torch.Generator().manual_seed(1)
    for m in model.modules():                                # SURVEY 8(d) initialisation
        if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)):
            fan_in = m.weight[0].numel()
            m.weight.data = torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5
            if m.bias is not None:
                m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.1
        elif isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = torch.rand(m.weight.shape, generator=g) + 0.5
            m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.1
            m.running_mean = torch.randn(m.running_mean.shape, generator=g) * 0.1
            m.running_var = torch.rand(m.running_var.shape, generator=g) + 0.5
    if not regression:
        last = [m for m in model.modules() if isinstance(m, torch.nn.Linear)][-1]
        last.weight.data *= logit_gain                      # keep the softmax away from saturation (a saturated output tests nothing)
    model.eval()
    if regression:
        x = torch.randn(B, in_shape[0], generator=g)
    elif len(in_shape) == 4:
        x = torch.randn(B, *in_shape[1:], generator=g)
    else:
        x = torch.rand(B, *in_shape, generator=g)
    state = {k: v.detach().numpy().copy() for k, v in model.state_dict().items() if v.dtype.is_floating_point}
    shapes, hooks = [], []
    orig = torch.Tensor.bernoulli_

    def discover(t, p=0.5, *, generator=None):
        shapes.append(tuple(t.shape))
        return orig(t, p)

    torch.Tensor.bernoulli_ = discover
    try:
        with torch.no_grad():
            model(x)                                         # discover the mask-draw order and shapes
    finally:
        torch.Tensor.bernoulli_ = orig
    n_drop = sum(1 for m in model.modules() if isinstance(m, BernoulliDropout))
    assert len(shapes) == n_drop, (len(shapes), n_drop)
    keep = np.float32(1.0) - np.float32(P)
    queue = []

    def bernoulli_(t, p=0.5, *, generator=None):
        m = queue.pop(0)
        assert tuple(t.shape) == m.shape
        t.copy_(torch.from_numpy(m))
        return t

    def arm(s):
        queue[:] = [(orc.fill_uniform(int(np.prod(sh)), SEED, di, s) < keep).astype(np.float32).reshape(sh) for di, sh in enumerate(shapes)]

    outs, outs_aten = [], []
    torch.Tensor.bernoulli_ = bernoulli_
    try:
        with torch.no_grad():
            for s in range(S):
                arm(s)
                outs.append(model(x))
                assert not queue
                arm(s)
                with torch.backends.mkldnn.flags(enabled=False):
                    outs_aten.append(model(x))
    finally:
        torch.Tensor.bernoulli_ = orig
    net = orc.F32MCOracle(state)
    res = {"x": x.numpy(), "meta.philox_seed": np.int64(SEED), "meta.p": np.float32(P), "meta.n_dropouts": np.int64(n_drop)}
    if regression:
        mu = np.stack([o[0].numpy() for o in outs]); var = np.stack([o[1].numpy() for o in outs])
        if altref.alt_out_path():
            np.savez(altref.alt_out_path(), mu=mu, var=var)
            return
        alt = altref.run_alt(os.path.abspath(__file__), out)
        mu_a, var_a = alt["mu"], alt["var"]
        mu_t, var_t = [o[0] for o in outs], [o[1] for o in outs]
        mean = torch.stack(mu_t, dim=1).mean(dim=1)                                           # experiments/utils.py:351
        pvar = torch.stack(mu_t, dim=1).var(dim=1) + torch.stack(var_t, dim=1).mean(dim=1)    # :352
        spread_abs = float(np.abs(mu - mu_a).max())                  # mu: absolute (its values cross zero) ...
        spread_rel = float((np.abs(var - var_a) / var).max())        # ... var = exp(log_var): relative
        o = [net.mlp(x.numpy(), SEED, s) for s in range(S)]
        err = max(np.abs(np.stack([a for a, _ in o]) - mu).max() / np.abs(mu).max(), (np.abs(np.stack([b for _, b in o]) - var) / var).max())
        res.update({"mu": mu, "var": var, "mu_aten": mu_a, "var_aten": var_a, "mean": mean.numpy(), "pred_var": pvar.numpy(),
                    "meta.in_dim": np.int64(in_shape[0])})
    else:
        probs = np.stack([o.numpy() for o in outs]); probs_a = np.stack([o.numpy() for o in outs_aten])
        spread_abs = float(np.abs(probs - probs_a).max())
        spread_rel = float((np.abs(probs - probs_a) / np.maximum(np.minimum(probs, probs_a), 1e-30)).max())
        fwd = net.lenet if "lenet" in model_name else net.resnet
        o = np.stack([fwd(x.numpy(), SEED, s) for s in range(S)])
        err = np.abs(o - probs).max()
        print(f"{model_name}: max prob {probs.max():.3f}, median of row max {np.median(probs.max(-1)):.3f}, sample-to-sample max diff {np.abs(probs[0] - probs[1]).max():.3f}")
        res.update({"probs": probs, "probs_aten": probs_a,
                    "mean_probs": torch.stack([torch.from_numpy(p) for p in probs], dim=1).mean(dim=1).numpy()})
    print(f"{model_name} float MC-Dropout: reference vs reference on another code path: max abs diff {spread_abs:.2e}, max rel {spread_rel:.2e}; "
          f"oracle vs reference max err {err:.2e}")
    assert err < 2e-5
    res["refspread.max_abs"], res["refspread.max_rel"] = np.float64(spread_abs), np.float64(spread_rel)
    res.update({"state/" + k: v for k, v in state.items()})
    path = os.path.join(HERE, out)
    np.savez_compressed(path, **res)
    print("wrote", path, round(os.path.getsize(path) / 1e6, 2), "MB")


if __name__ == "__main__":
    if altref.alt_tag() in (None, "mlp_mc_f32.npz"):
        run("linear_mc", [13], 1, 200, 4, "mlp_mc_f32.npz")
    if altref.alt_tag() is None:
        run("conv_lenet_mc", [1, 28, 28], 10, 4, 3, "lenet_mc_f32.npz", 0.2)
        run("conv_resnet_mc", [1, 3, 32, 32], 10, 2, 3, "resnet_mc_f32.npz", 0.02)
