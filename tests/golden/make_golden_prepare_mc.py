#!/usr/bin/env python3
"""Fixture for the native prepare -> calibrate -> convert pipeline of the MC-Dropout ResNet (SURVEY 8f row 4 widened to quant_utils.prepare_model's
`prepare_qat` branch, :139-140).  RUNS ONLY IN THE BUILD CONTAINER.

Takes the FLOAT conv_resnet_mc already committed as a fixture (state of tests/golden/resnet_mc_f32.npz), loads it into the real reference's float
model, lets the REFERENCE prepare it (fusion, qconfig, prepare_qat), calibrates it with S eval-mode forwards of the fixture's input (the live
observers update in eval; the build's Philox masks injected into Tensor.bernoulli_: forward i draws sample index i), converts it
(quant_utils.convert) and runs the int8 model on the same input with the masks of sample indices S .. S + 1.  Records the observers after
calibration, the converted model's quantisation parameters + a hash of every int8 tensor, and the int8 probabilities.  Plain ATen convs
(mkldnn off), as make_golden_prepare.py explains.
Output: tests/golden/resnet_mc_prepare_calibrate.npz (data only)."""
import hashlib
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

SEED, S, S_EVAL, LOGIT_GAIN = 3, 3, 2, 1.0


def main():
    from src.models import ModelFactory
    import src.quant_utils as qu
    d = np.load(os.path.join(HERE, "resnet_mc_f32.npz"))
    fstate = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
    fstate["layers.10.weight"] = (np.asarray(fstate["layers.10.weight"]) * np.float32(LOGIT_GAIN)).astype(np.float32)     # (7-bit logits of the fixture's own head are all equal)
    x = torch.from_numpy(d["x"])
    P = float(d["meta.p"])
    args = types.SimpleNamespace(p=P, model="conv_resnet_mc", q=True, at=True, samples=S, task="classification", activation_precision=7, weight_precision=8)
    torch.manual_seed(1)
    model = ModelFactory.get_model("conv_resnet_mc", [1, 3, 32, 32], 10, True, args)
    sd = model.state_dict()
    for k, v in fstate.items():
        assert k in sd, k
        sd[k] = torch.from_numpy(np.asarray(v))
    model.load_state_dict(sd)
    qu.prepare_model(model, args)
    model.eval()
    mk = torch.backends.mkldnn.flags(enabled=False)
    mk.__enter__()
    snap = {k: v.clone() for k, v in model.state_dict().items()}
    shapes, orig = [], torch.Tensor.bernoulli_

    def discover(t, p=0.5, *, generator=None):
        shapes.append(tuple(t.shape))
        return orig(t, p)

    torch.Tensor.bernoulli_ = discover
    try:
        with torch.no_grad():
            model(x)
    finally:
        torch.Tensor.bernoulli_ = orig
    model.load_state_dict(snap)
    keep = np.float32(1.0) - np.float32(P)
    queue = []

    def bernoulli_(t, p=0.5, *, generator=None):
        m = queue.pop(0)
        assert tuple(t.shape) == m.shape
        t.copy_(torch.from_numpy(m))
        return t

    def arm(s):
        queue[:] = [(orc.fill_uniform(int(np.prod(sh)), SEED, di, s) < keep).astype(np.float32).reshape(sh) for di, sh in enumerate(shapes)]

    torch.Tensor.bernoulli_ = bernoulli_
    out = {"meta.samples": np.int64(S), "meta.philox_seed": np.int64(SEED), "meta.p": np.float32(P), "meta.logit_gain": np.float32(LOGIT_GAIN)}
    try:
        with torch.no_grad():
            for s in range(S):
                arm(s)
                model(x)
                assert not queue
        for k, v in model.state_dict().items():
            if k.endswith("min_val") or k.endswith("max_val"):
                out["calibrated/" + k] = v.detach().numpy().copy()
        mk.__exit__(None, None, None)
        qu.convert(model)
        model.eval()
        probs = []
        with torch.no_grad():
            for s in range(S, S + S_EVAL):
                arm(s)
                probs.append(model(x).numpy().copy())
                assert not queue
    finally:
        torch.Tensor.bernoulli_ = orig
    out["int8_probs"] = np.stack(probs)
    n_q = 0
    for k, v in model.state_dict().items():
        if v is None or "_packed_params" in k:
            continue
        if isinstance(v, torch.Tensor) and v.is_quantized:
            out["converted/" + k + ".q_scale"] = np.float64(v.q_scale())
            out["converted/" + k + ".q_zero_point"] = np.int64(v.q_zero_point())
            out["converted/" + k + ".sha1"] = np.array(hashlib.sha1(v.int_repr().numpy().tobytes()).hexdigest())
            n_q += 1
            if k in ("layers.0.weight", "layers.5.0.shortcut.0.weight", "layers.7.1.stem.4.weight"):
                out["converted/" + k] = v.int_repr().numpy()
        elif isinstance(v, torch.Tensor):
            out["converted/" + k] = v.detach().numpy()
        else:
            out["converted/" + k] = np.asarray(v)
    w = model.layers[10].weight()
    out["converted/layers.10.weight"] = w.int_repr().numpy()
    out["converted/layers.10.weight.q_scale"] = np.float64(w.q_scale())
    out["converted/layers.10.weight.q_zero_point"] = np.int64(w.q_zero_point())
    path = os.path.join(HERE, "resnet_mc_prepare_calibrate.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, round(os.path.getsize(path) / 1e6, 3), "MB;", n_q, "qint8 tensors;", len(shapes), "mask draws per forward; int8 probs", out["int8_probs"].shape,
          "max", float(out["int8_probs"].max()))


if __name__ == "__main__":
    main()
