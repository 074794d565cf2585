#!/usr/bin/env python3
"""Golden vectors for the MC-Dropout ResNet (`conv_resnet_mc`), int8 A7/W8.  RUNS ONLY IN THE BUILD
CONTAINER.  Imports the real reference, prepare_model -> calibration -> convert (quant_utils.py:62-147), injects the
build's Philox Bernoulli masks into Tensor.bernoulli_ in draw order and records block outputs of sample 0 and the
per-sample softmax outputs.  Output: tests/golden/resnet_mc_a7w8.npz (inputs + expected outputs only)."""
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

SEED, B, S, P = 3, 4, 3, 0.1


def flat_state(model):
    out = {}
    for k, v in model.state_dict().items():
        if v is None or not torch.is_tensor(v):
            continue
        if v.is_quantized:
            out[k] = v.int_repr().numpy()
            out[k + ".q_scale"] = np.float64(v.q_scale())
            out[k + ".q_zero_point"] = np.int64(v.q_zero_point())
        else:
            out[k] = v.detach().numpy()
    return out


def main():
    from src.models import ModelFactory
    import src.quant_utils as qu
    args = types.SimpleNamespace(p=P, activation_precision=7, weight_precision=8, model="conv_resnet_mc", q=True, at=True,
                                 samples=S, task="classification")
    torch.manual_seed(1)
    model = ModelFactory.get_model("conv_resnet_mc", [1, 3, 32, 32], 10, True, args)
    g = torch.Generator().manual_seed(1)
    for m in model.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)):
            fan_in = m.weight[0].numel()
            m.weight.data = torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5
        elif isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = torch.rand(m.weight.shape, generator=g) + 0.5
            m.bias.data.zero_()
    qu.prepare_model(model, args)
    xcal = torch.randn(32, 3, 32, 32, generator=g)
    model.train()
    for _ in range(2):
        model(xcal)
    model.eval()
    with torch.no_grad():
        for _ in range(2):
            model(xcal)
    qu.convert(model)
    model.eval()
    state = flat_state(model)
    fc = model.layers[10]                                     # quantized Linear keeps its tensors in _packed_params
    w = fc.weight()
    state["layers.10.weight"] = w.int_repr().numpy()
    state["layers.10.weight.q_scale"] = np.float64(w.q_scale())
    state["layers.10.weight.q_zero_point"] = np.int64(w.q_zero_point())
    if fc.bias() is not None:
        state["layers.10.bias"] = fc.bias().detach().numpy()
    state = {k: v for k, v in state.items() if "_packed_params" not in k}
    x = torch.randn(B, 3, 32, 32, generator=g)

    # discover the mask shapes in draw order
    shapes = []
    orig = torch.Tensor.bernoulli_

    def probe(t, p=0.5, *, generator=None):
        shapes.append(tuple(t.shape))
        return orig(t, p)
    torch.Tensor.bernoulli_ = probe
    try:
        with torch.no_grad():
            model(x)
    finally:
        torch.Tensor.bernoulli_ = orig
    keep = np.float32(1.0) - np.float32(P)
    queue = []

    def bernoulli_(t, p=0.5, *, generator=None):
        m = queue.pop(0)
        assert tuple(t.shape) == m.shape, (t.shape, m.shape)
        t.copy_(torch.from_numpy(m))
        return t

    rec, hooks = {}, []

    def mk(name):
        def hook(_m, _i, o):
            rec[name + ".out"] = np.minimum(np.ascontiguousarray(o.int_repr().numpy().transpose(0, 2, 3, 1)), 127).astype(np.uint8)
        return hook
    hooks.append(model.layers[3].register_forward_hook(mk("layers.3")))
    for li in (4, 5, 6, 7):
        for bi, blk in enumerate(model.layers[li]):
            hooks.append(blk.register_forward_hook(mk(f"layers.{li}.{bi}")))
    probs = []
    torch.Tensor.bernoulli_ = bernoulli_
    try:
        with torch.no_grad():
            for s in range(S):
                queue[:] = [(orc.fill_uniform(int(np.prod(sh)), SEED, di, s) < keep).astype(np.float32).reshape(sh) for di, sh in enumerate(shapes)]
                probs.append(model(x).numpy().copy())
                assert not queue
                if s == 0:
                    for h in hooks:
                        h.remove()
    finally:
        torch.Tensor.bernoulli_ = orig
    probs = np.stack(probs)
    net = orc.Int8ResNetMCOracle(state, 7)
    orec = {}
    p0 = net.forward(x.numpy(), SEED, 0, record=orec)
    bad = sum(int((orec[k] != v).sum()) for k, v in rec.items())
    err = max(np.abs(net.forward(x.numpy(), SEED, s) - probs[s]).max() for s in range(S))
    print(f"oracle vs reference (ResNet MC-Dropout): {bad} mismatching integer elements over {len(rec)} tensors; probs max abs err {err:.2e}; "
          f"{len(shapes)} dropouts; row max median {np.median(probs.max(-1)):.3f}")
    assert bad == 0 and err < 1e-6
    out = {"x": x.numpy(), "probs": probs, "mean_probs": torch.stack([torch.from_numpy(p) for p in probs], dim=1).mean(dim=1).numpy(),
           "meta.philox_seed": np.int64(SEED), "meta.a_bits": np.int64(7), "meta.w_bits": np.int64(8), "meta.p": np.float32(P)}
    out.update({"state/" + k: v for k, v in state.items()})
    out.update({"rec/" + k: v for k, v in rec.items() if k in ("layers.3.out", "layers.4.1.out", "layers.5.0.out", "layers.7.1.out")})
    path = os.path.join(HERE, "resnet_mc_a7w8.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, round(os.path.getsize(path) / 1e6, 2), "MB")


if __name__ == "__main__":
    main()
