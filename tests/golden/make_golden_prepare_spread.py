#!/usr/bin/env python3
"""How far apart do two runs of the REFERENCE's own prepare -> calibrate -> convert land?  RUNS ONLY IN THE BUILD CONTAINER.

tests/golden/make_golden_prepare.py records one run of the reference pipeline (quant_utils.prepare_model -> 3 evaluation-mode
forwards with live observers -> quant_utils.convert) as the fixture of the native pipeline's end-to-end test.  The observers start
unseen, every fake-quantised activation feeds the next layer's observer, and the convs sum in fp32 in whatever order the backend
picks -- so the calibrated ranges depend on the summation order.  This script runs the SAME reference pipeline under different conv
backends / thread counts (mkldnn on / off, 1 / 8 threads: different fp32 summation orders of the same arithmetic), with the same
injected noise, and records per observer the largest deviation between any two runs, as a fraction of the observer's range, plus the
deviation of the converted activation qparams.  The native pipeline's test tolerances are these measured spreads.
Output: tests/golden/resnet_bbb_prepare_spread.npz (data only)."""
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

SEED, S = 3, 3


def run(threads, mkldnn):
    from src.models import ModelFactory
    import src.quant_utils as qu
    torch.set_num_threads(threads)
    d = np.load(os.path.join(HERE, "resnet_bbb_f32.npz"))
    fstate = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
    x = torch.from_numpy(d["x"])
    args = types.SimpleNamespace(sigma_prior=-2.0, model="conv_resnet_bbb", q=True, at=True, samples=S, task="classification",
                                 activation_precision=7, weight_precision=8)
    torch.manual_seed(1)
    model = ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, args)
    sd = model.state_dict()
    for k, v in fstate.items():
        sd[k] = torch.from_numpy(np.asarray(v))
    model.load_state_dict(sd)
    qu.prepare_model(model, args)
    model.eval()
    shapes = []
    hooks = [m.register_forward_pre_hook(lambda m, i: shapes.append(tuple(m.weight.shape))) for m in model.modules() if hasattr(m, "weight_fake_quant")]
    snap = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.backends.mkldnn.flags(enabled=mkldnn), torch.no_grad():
        model(x)
    for h in hooks:
        h.remove()
    model.load_state_dict(snap)
    queue, orig = [], torch.Tensor.normal_

    def normal_(t, mean=0, std=1, *, generator=None):
        e = queue.pop(0)
        t.copy_(torch.from_numpy(e))
        return t

    torch.Tensor.normal_ = normal_
    try:
        with torch.backends.mkldnn.flags(enabled=mkldnn), torch.no_grad():
            for s in range(S):
                queue[:] = [orc.fill_normal(int(np.prod(sh)), SEED, lid, s).reshape(sh) for lid, sh in enumerate(shapes)]
                model(x)
    finally:
        torch.Tensor.normal_ = orig
    obs = {k: float(v) for k, v in model.state_dict().items() if k.endswith("min_val") or k.endswith("max_val")}
    qu.convert(model)
    conv = {}
    for k, v in model.state_dict().items():
        if isinstance(v, torch.Tensor) and not v.is_quantized and (k.endswith("scale") or k.endswith("zero_point")):
            conv[k] = float(np.asarray(v.detach().numpy()).reshape(-1)[0])
    return obs, conv


def main():
    configs = [(8, True), (1, True), (8, False), (1, False), (3, True)]
    runs = [run(t, m) for t, m in configs]
    base_obs, base_conv = runs[0]
    out = {"meta.configs": np.array(["threads=%d mkldnn=%d" % c for c in configs])}
    keys = sorted(k[:-len("min_val")] for k in base_obs if k.endswith("min_val"))
    worst = 0.0
    for i, (obs, conv) in enumerate(runs):                      # every run's observers and converted activation qparams
        for k, v in obs.items():
            out["run%d/calibrated/%s" % (i, k)] = np.float32(v)
        for k, v in conv.items():
            out["run%d/converted/%s" % (i, k)] = np.float64(v)
    for k in keys:
        los = np.array([r[0][k + "min_val"] for r in runs]); his = np.array([r[0][k + "max_val"] for r in runs])
        rng = max(his.max(), 0.0) - min(los.min(), 0.0)
        dev = max(los.max() - los.min(), his.max() - his.min()) / rng if rng > 0 else 0.0
        out["spread/" + k + "range_frac"] = np.float64(dev)
        worst = max(worst, dev)
    for k in sorted(base_conv):
        vals = np.array([r[1][k] for r in runs])
        if k.endswith("scale"):
            out["spread_conv/" + k + ".rel"] = np.float64((vals.max() - vals.min()) / max(abs(vals).max(), 1e-30))
        else:
            out["spread_conv/" + k + ".abs"] = np.float64(vals.max() - vals.min())
    path = os.path.join(HERE, "resnet_bbb_prepare_spread.npz")
    np.savez_compressed(path, **out)
    act = [float(out[k]) for k in out if k.startswith("spread/") and not any(t in k for t in ("weight_fake_quant", "std_fake_quant", "mul_noise", "add_weight"))]
    wgt = [float(out[k]) for k in out if k.startswith("spread/") and any(t in k for t in ("weight_fake_quant", "std_fake_quant", "mul_noise", "add_weight"))]
    print("wrote", path)
    print("activation observers: max spread %.3e of range, median %.3e; weight-side observers: max %.3e" % (max(act), float(np.median(act)), max(wgt)))
    print("converted scales: max rel spread %.3e; zero points: max abs spread %g" % (
        max(float(out[k]) for k in out if k.endswith(".rel")), max(float(out[k]) for k in out if k.endswith(".abs"))))


if __name__ == "__main__":
    main()
