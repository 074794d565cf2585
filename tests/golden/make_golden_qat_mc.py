#!/usr/bin/env python3
"""Golden vectors for QAT fake-quant EVALUATION with live observers of the NON-BBB graphs (SURVEY 8(f).3 widened: quant_utils.prepare_model's
`prepare_qat` branch, :139-140).  RUNS ONLY IN THE BUILD CONTAINER.
Imports the real reference -- `linear_mc`, `conv_lenet_mc`, `conv_resnet_mc` (mcdropout/models_mc.py, whose BernoulliDropout carries two
FloatFunctionals that receive FakeQuantize observers, dropout.py:9-13) and the SGHMC member template `conv_resnet_sgld`
(sgld/models_sgld.py: `main_net`) --, prepares it for QAT, warms the observers with one train-mode and one eval-mode forward, snapshots the
state_dict, then runs S eval-mode forwards of the same batch with the build's Philox Bernoulli masks injected into Tensor.bernoulli_ and records
the per-sample outputs and the observers' final (min, max).  As in make_golden_qat.py the whole pipeline also runs on another CPU code path
(altref.py) and the distance of the reference from itself is recorded (`refspread.*`).
Each fixture also holds `converted/*`: the flat int8 state the reference's own quant_utils.convert (:62-99) makes of the snapshot (the check of
convert.convert_model_state's non-BBB branch).
Output: tests/golden/{mlp,lenet,resnet}_mc_qat.npz, resnet_sgld_qat.npz (inputs + expected outputs only)."""
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
ALT = altref.alt_out_path() is not None


def flat(model):
    return {k: v.detach().numpy().copy() for k, v in model.state_dict().items()
            if v.dtype.is_floating_point and not k.endswith(".scale") and "fake_quant_enabled" not in k and "observer_enabled" not in k}


def converted_state(model):
    """The reference's quant_utils.convert (:62-99) on a copy of the prepared model: the flat int8 state (what make_golden_resnet_mc.py /
    make_golden_linear_mc.py record: qint8 tensors as int_repr + q_scale + q_zero_point; quantized Linear keeps its tensors in _packed_params)."""
    import copy
    import src.quant_utils as qu
    cm = copy.deepcopy(model).cpu().eval()
    qu.convert(cm)
    out = {}
    for k, v in cm.state_dict().items():
        if v is None or not torch.is_tensor(v) or "_packed_params" in k:
            continue
        if v.is_quantized:
            out[k] = v.int_repr().numpy()
            out[k + ".q_scale"] = np.float64(v.q_scale())
            out[k + ".q_zero_point"] = np.int64(v.q_zero_point())
        else:
            out[k] = v.detach().numpy()
    for n, m in cm.named_modules():
        if isinstance(m, torch.ao.nn.quantized.Linear):
            w = m.weight()
            out[n + ".weight"] = w.int_repr().numpy()
            out[n + ".weight.q_scale"] = np.float64(w.q_scale())
            out[n + ".weight.q_zero_point"] = np.int64(w.q_zero_point())
            if m.bias() is not None:
                out[n + ".bias"] = m.bias().detach().numpy().copy()
    return out


def run(model_name, in_shape, B, S, out, logit_gain):
    from src.models import ModelFactory
    import src.quant_utils as qu
    regression = model_name.startswith("linear")
    args = types.SimpleNamespace(p=P, model=model_name, q=True, at=True, samples=S, activation_precision=7, weight_precision=8,
                                 task="regression" if regression else "classification", sigma_prior=-2.0)
    torch.manual_seed(1)
    model = ModelFactory.get_model(model_name, in_shape, 1 if regression else 10, True, args)
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
    if not regression:
        last = [m for m in model.modules() if isinstance(m, torch.nn.Linear)][-1]
        last.weight.data *= logit_gain
    qu.prepare_model(model, args)
    if ALT and not regression:
        torch.backends.mkldnn.enabled = False                # the alternative run of a conv graph: plain ATen convs
    if regression:
        x = torch.randn(B, in_shape[0], generator=g)
    elif len(in_shape) == 4:
        x = torch.randn(B, *in_shape[1:], generator=g)
    else:
        x = torch.rand(B, *in_shape, generator=g)
    torch.manual_seed(101)
    model.train(); model(x)                                  # BatchNorm statistics + every observer sees one batch
    model.eval()
    with torch.no_grad():
        model(x)
    state = flat(model)
    conv = converted_state(model) if not ALT else {}          # what the reference's convert() makes of exactly this prepared state
    snap = {k: v.clone() for k, v in model.state_dict().items()}
    shapes = []
    orig = torch.Tensor.bernoulli_

    def discover(t, p=0.5, *, generator=None):
        shapes.append(tuple(t.shape))
        return orig(t, p)

    torch.Tensor.bernoulli_ = discover
    try:
        with torch.no_grad():
            model(x)                                         # the mask-draw order and shapes
    finally:
        torch.Tensor.bernoulli_ = orig
    model.load_state_dict(snap)                               # rewind the observers to the snapshot
    keep = np.float32(1.0) - np.float32(P)
    queue = []

    def bernoulli_(t, p=0.5, *, generator=None):
        m = queue.pop(0)
        assert tuple(t.shape) == m.shape
        t.copy_(torch.from_numpy(m))
        return t

    outs = []
    torch.Tensor.bernoulli_ = bernoulli_
    try:
        with torch.no_grad():
            for s in range(S):
                queue[:] = [(orc.fill_uniform(int(np.prod(sh)), SEED, di, s) < keep).astype(np.float32).reshape(sh) for di, sh in enumerate(shapes)]
                o = model(x)
                outs.append([t.numpy().copy() for t in o] if regression else o.numpy().copy())
                assert not queue
    finally:
        torch.Tensor.bernoulli_ = orig
    final = flat(model)
    if ALT:
        if altref.alt_tag() == out:
            np.savez(altref.alt_out_path(), **({"mu": np.stack([o[0] for o in outs]), "var": np.stack([o[1] for o in outs])} if regression
                                               else {"probs": np.stack(outs)}))
        return
    alt = altref.run_alt(os.path.abspath(__file__), out)
    if regression:
        sp = altref.spread(np.stack([o[0] for o in outs]), alt["mu"]), altref.spread(np.stack([o[1] for o in outs]), alt["var"])
        spread_abs, spread_rel = sp[0][0], sp[1][1]
    else:
        spread_abs, spread_rel = altref.spread(np.stack(outs), alt["probs"])
    print(f"{model_name} QAT eval: reference vs reference on another code path: max abs {spread_abs:.2e}, max rel {spread_rel:.2e}; {len(shapes)} mask draws per forward")
    sgld = model_name.endswith("_sgld")
    net = orc.QATMCOracle(state, prefix="main_net." if sgld else "")
    xin = x.numpy()
    worst = 0.0
    for s in range(S):
        if regression:
            o = net.mlp_mc(xin, SEED, s)
            worst = max(worst, np.abs(o[0] - outs[s][0]).max(), np.abs(o[1] - outs[s][1]).max() / np.abs(outs[s][1]).max())
        else:
            o = net.resnet_p(xin) if sgld else (net.lenet_mc(xin, SEED, s) if "lenet" in model_name else net.resnet_mc(xin, SEED, s))
            worst = max(worst, np.abs(o - outs[s]).max())
    obs_err = max(abs(float(v.state[0]) - float(final[k + ".activation_post_process.min_val"])) +
                  abs(float(v.state[1]) - float(final[k + ".activation_post_process.max_val"])) for k, v in net.obs.items())
    print(f"{model_name} QAT eval: oracle vs reference max abs err {worst:.2e}; observer state err {obs_err:.2e}; {len(net.obs)} live observers")
    res = {"x": xin, "meta.philox_seed": np.int64(SEED), "meta.p": np.float32(P), "refspread.max_abs": np.float64(spread_abs),
           "refspread.max_rel": np.float64(spread_rel)}
    if regression:
        res["mu"] = np.stack([o[0] for o in outs]); res["var"] = np.stack([o[1] for o in outs])
    else:
        res["probs"] = np.stack(outs)
        res["mean_probs"] = torch.stack([torch.from_numpy(p) for p in outs], dim=1).mean(dim=1).numpy()
    res.update({"state/" + k: v for k, v in state.items()})
    res.update({"final/" + k: v for k, v in final.items() if k.endswith("min_val") or k.endswith("max_val")})
    res.update({"converted/" + k: v for k, v in conv.items()})
    path = os.path.join(HERE, out)
    np.savez_compressed(path, **res)
    print("wrote", path, round(os.path.getsize(path) / 1e6, 2), "MB")


CASES = [("linear_mc", [13], 64, 4, "mlp_mc_qat.npz", 1.0), ("conv_lenet_mc", [1, 28, 28], 4, 3, "lenet_mc_qat.npz", 0.2),
         ("conv_resnet_mc", [1, 3, 32, 32], 2, 3, "resnet_mc_qat.npz", 0.05), ("conv_resnet_sgld", [1, 3, 32, 32], 2, 3, "resnet_sgld_qat.npz", 0.05)]

if __name__ == "__main__":
    for c in CASES:
        if altref.alt_tag() in (None, c[4]):
            run(*c)
