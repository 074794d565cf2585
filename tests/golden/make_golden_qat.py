#!/usr/bin/env python3
"""Golden vectors for QAT fake-quant EVALUATION with live observers.  RUNS ONLY IN THE BUILD CONTAINER.
Imports the real reference, prepares the model for QAT (quant_utils.prepare_model :109-147), warms the observers with one
train-mode and one eval-mode forward, snapshots the state_dict (parameters, BN statistics, every observer's min/max),
then runs S eval-mode forwards of the same batch with the build's Philox eps injected into Tensor.normal_ and records
the per-sample outputs and the observers' final min/max.
Output: tests/golden/{lenet,mlp,resnet}_bbb_qat.npz (inputs + expected outputs only).  Each fixture also records how far the reference
is from itself when the whole pipeline (calibration forwards included) runs on another CPU code path (altref.py; conv graphs: oneDNN
off as well): `refspread.*`."""
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
ALT = altref.alt_out_path() is not None


def flat(model):
    return {k: v.detach().numpy().copy() for k, v in model.state_dict().items()
            if v.dtype.is_floating_point and not k.endswith(".scale") and "fake_quant_enabled" not in k and "observer_enabled" not in k}


def run(model_name, in_shape, B, S, out, logit_gain, regression=False):
    from src.models import ModelFactory
    import src.quant_utils as qu
    from src.models.stochastic.bbb.conv import Conv2d as Conv2dBBB
    from src.models.stochastic.bbb.linear import Linear as LinearBBB
    args = types.SimpleNamespace(sigma_prior=-2.0, model=model_name, q=True, at=True, samples=S, activation_precision=7, weight_precision=8,
                                 task="regression" if regression else "classification")
    torch.manual_seed(1)
    model = ModelFactory.get_model(model_name, in_shape, 1 if regression else 10, True, args)
    g = torch.Generator().manual_seed(1)
    for m in model.modules():
        if isinstance(m, (Conv2dBBB, LinearBBB)):
            fan_in = m.weight[0].numel()
            m.weight.data = torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5
            m.std.data.fill_(-3.0)
            if m.bias is not None:
                m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.1
        elif isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = torch.rand(m.weight.shape, generator=g) + 0.5
            m.bias.data.zero_()
    last = [m for m in model.modules() if isinstance(m, LinearBBB)][-1]
    if not regression:
        last.weight.data *= logit_gain
    qu.prepare_model(model, args)
    if ALT and not regression:
        torch.backends.mkldnn.enabled = False                # the alternative run of a conv graph: plain ATen convs on the AVX2 kernels
    if regression:
        x = torch.randn(B, in_shape[0], generator=g)
    elif len(in_shape) == 4:
        x = torch.randn(B, *in_shape[1:], generator=g)
    else:
        x = torch.rand(B, *in_shape, generator=g)
    torch.manual_seed(101)
    model.train(); model(x)
    model.eval()
    with torch.no_grad():
        model(x)
    state = flat(model)
    qat_types = tuple(t for t in (type(m) for m in model.modules()) if t.__name__.startswith("QAT") or "qat" in t.__module__)
    shapes = []
    hooks = [m.register_forward_pre_hook(lambda m, i: shapes.append(tuple(m.weight.shape)))
             for m in model.modules() if hasattr(m, "weight_fake_quant")]
    snap = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        model(x)
    for h in hooks:
        h.remove()
    model.load_state_dict(snap)                               # rewind the observers to the snapshot
    queue = []
    orig = torch.Tensor.normal_

    def normal_(t, mean=0, std=1, *, generator=None):
        e = queue.pop(0)
        assert tuple(t.shape) == e.shape
        t.copy_(torch.from_numpy(e))
        return t

    outs = []
    torch.Tensor.normal_ = normal_
    try:
        with torch.no_grad():
            for s in range(S):
                queue[:] = [orc.fill_normal(int(np.prod(sh)), SEED, lid, s).reshape(sh) for lid, sh in enumerate(shapes)]
                o = model(x)
                outs.append([t.numpy().copy() for t in o] if regression else o.numpy().copy())
                assert not queue
    finally:
        torch.Tensor.normal_ = orig
    final = flat(model)
    if ALT:                                    # this process is the reference on the other code path: outputs only
        if altref.alt_tag() == out:
            np.savez(altref.alt_out_path(), **({"mu": np.stack([o[0] for o in outs]), "var": np.stack([o[1] for o in outs])} if regression
                                               else {"probs": np.stack(outs)}))
        return
    alt = altref.run_alt(os.path.abspath(__file__), out)
    if regression:
        sp = altref.spread(np.stack([o[0] for o in outs]), alt["mu"]), altref.spread(np.stack([o[1] for o in outs]), alt["var"])
        spread_abs, spread_rel = sp[0][0], sp[1][1]
        print(f"{model_name} QAT eval: reference vs reference on another code path: mu max abs {spread_abs:.2e} (range {np.abs(alt['mu']).max():.2f}), var max rel {spread_rel:.2e}")
    else:
        spread_abs, spread_rel = altref.spread(np.stack(outs), alt["probs"])
        print(f"{model_name} QAT eval: reference vs reference on another code path: probs max abs {spread_abs:.2e}, max rel {spread_rel:.2e}")
    net = orc.QATOracle(state)
    fwd = net.mlp if regression else (net.lenet if "lenet" in model_name else net.resnet)
    xin = x.numpy()
    worst = 0.0
    for s in range(S):
        o = fwd(xin, SEED, s)
        if regression:
            worst = max(worst, np.abs(o[0] - outs[s][0]).max(), np.abs(o[1] - outs[s][1]).max() / np.abs(outs[s][1]).max())
        else:
            worst = max(worst, np.abs(o - outs[s]).max())
    obs_err = max(abs(float(v.state[0]) - float(final[k + ".activation_post_process.min_val"])) +
                  abs(float(v.state[1]) - float(final[k + ".activation_post_process.max_val"])) for k, v in net.obs.items())
    print(f"{model_name} QAT eval: oracle vs reference max abs err {worst:.2e}; observer state err {obs_err:.2e}; {len(net.obs)} observers")
    res = {"x": xin, "meta.philox_seed": np.int64(SEED), "refspread.max_abs": np.float64(spread_abs), "refspread.max_rel": np.float64(spread_rel)}
    if regression:
        res["mu"] = np.stack([o[0] for o in outs]); res["var"] = np.stack([o[1] for o in outs])
    else:
        res["probs"] = np.stack(outs)
        res["mean_probs"] = torch.stack([torch.from_numpy(p) for p in outs], dim=1).mean(dim=1).numpy()
    res.update({"state/" + k: v for k, v in state.items()})
    res.update({"final/" + k: v for k, v in final.items() if k.endswith("min_val") or k.endswith("max_val")})
    path = os.path.join(HERE, out)
    np.savez_compressed(path, **res)
    print("wrote", path, round(os.path.getsize(path) / 1e6, 2), "MB")


CASES = [("linear_bbb", [13], 64, 4, "mlp_bbb_qat.npz", 1.0, True), ("conv_lenet_bbb", [1, 28, 28], 4, 3, "lenet_bbb_qat.npz", 0.2, False),
         ("conv_resnet_bbb", [1, 3, 32, 32], 2, 3, "resnet_bbb_qat.npz", 0.05, False)]

if __name__ == "__main__":
    for c in CASES:
        if altref.alt_tag() in (None, c[4]):
            run(*c[:6], regression=c[6])
