#!/usr/bin/env python3
"""Golden vectors for the FLOAT Bayes-by-backprop conv graphs.  RUNS ONLY IN THE BUILD CONTAINER.
Imports the real reference (`conv_lenet_bbb` and `conv_resnet_bbb` with q=False, eval mode: bbb/conv.py:33-39,
bbb/linear.py:42-50), injects the build's Philox eps into Tensor.normal_ in draw order and records the per-sample
softmax outputs and their mean (experiments/utils.py:342-355).
Output: tests/golden/lenet_bbb_f32.npz, tests/golden/resnet_bbb_f32.npz (inputs + expected outputs only)."""
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

SEED = 3


def run(model_name, in_shape, B, S, out, logit_gain):
    from src.models import ModelFactory
    from src.models.stochastic.bbb.conv import Conv2d as Conv2dBBB
    from src.models.stochastic.bbb.linear import Linear as LinearBBB
    args = types.SimpleNamespace(sigma_prior=-2.0, model=model_name, q=False, at=False, samples=S, task="classification",
                                 activation_precision=7, weight_precision=8)
    torch.manual_seed(1)
    model = ModelFactory.get_model(model_name, in_shape, 10, False, args)
    g = torch.Generator().manual_seed(1)
    for m in model.modules():                                # SURVEY 8(d) initialisation
        if isinstance(m, (Conv2dBBB, LinearBBB)):
            fan_in = m.weight[0].numel()
            m.weight.data = torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5
            m.std.data.fill_(-3.0)
        elif isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = torch.rand(m.weight.shape, generator=g) + 0.5
            m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.1
            m.running_mean = torch.randn(m.running_mean.shape, generator=g) * 0.1
            m.running_var = torch.rand(m.running_var.shape, generator=g) + 0.5
    last = [m for m in model.modules() if isinstance(m, LinearBBB)][-1]
    last.weight.data *= logit_gain                          # keep the softmax away from saturation (a saturated output tests nothing)
    if logit_gain < 0.01:
        last.std.data.fill_(-12.0)
    model.eval()
    x = torch.randn(B, *in_shape[1:], generator=g) if len(in_shape) == 4 else torch.rand(B, *in_shape, generator=g)
    state = {k: v.detach().numpy().copy() for k, v in model.state_dict().items() if v.dtype.is_floating_point}
    shapes = []
    hooks = []

    def rec(m, i):
        shapes.append(tuple(m.weight.shape))
    for m in model.modules():
        if isinstance(m, (Conv2dBBB, LinearBBB)):
            hooks.append(m.register_forward_pre_hook(rec))
    with torch.no_grad():
        model(x)                                             # discover the noise-draw order
    for h in hooks:
        h.remove()
    queue = []
    orig = torch.Tensor.normal_

    def normal_(t, mean=0, std=1, *, generator=None):
        e = queue.pop(0)
        assert tuple(t.shape) == e.shape
        t.copy_(torch.from_numpy(e))
        return t

    probs, probs_aten = [], []
    torch.Tensor.normal_ = normal_
    try:
        with torch.no_grad():
            for s in range(S):
                queue[:] = [orc.fill_normal(int(np.prod(sh)), SEED, lid, s).reshape(sh) for lid, sh in enumerate(shapes)]
                probs.append(model(x).numpy().copy())
                assert not queue
                # the same forward on the reference's OTHER conv backend (plain ATen instead of oneDNN: another fp32 summation order):
                # how far the reference is from itself is the floor of any fp32 comparison against it
                queue[:] = [orc.fill_normal(int(np.prod(sh)), SEED, lid, s).reshape(sh) for lid, sh in enumerate(shapes)]
                with torch.backends.mkldnn.flags(enabled=False):
                    probs_aten.append(model(x).numpy().copy())
    finally:
        torch.Tensor.normal_ = orig
    probs, probs_aten = np.stack(probs), np.stack(probs_aten)
    spread_abs = float(np.abs(probs - probs_aten).max())
    spread_rel = float((np.abs(probs - probs_aten) / np.maximum(np.minimum(probs, probs_aten), 1e-30)).max())
    print(f"{model_name} float: reference oneDNN vs reference ATen max abs prob diff {spread_abs:.2e}, max rel {spread_rel:.2e}")
    net = orc.F32ConvOracle(state)
    fwd = net.lenet if "lenet" in model_name else net.resnet
    o = np.stack([fwd(x.numpy(), SEED, s) for s in range(S)])
    err = np.abs(o - probs).max()
    print(f"{model_name} float: oracle vs reference max abs prob err {err:.2e} (max prob {probs.max():.3f}, median of row max {np.median(probs.max(-1)):.3f})")
    assert err < 2e-5
    res = {"x": x.numpy(), "probs": probs, "mean_probs": torch.stack([torch.from_numpy(p) for p in probs], dim=1).mean(dim=1).numpy(),
           "meta.philox_seed": np.int64(SEED), "refspread.max_abs": np.float64(spread_abs), "refspread.max_rel": np.float64(spread_rel),
           "probs_aten": probs_aten}
    res.update({"state/" + k: v for k, v in state.items()})
    path = os.path.join(HERE, out)
    np.savez_compressed(path, **res)
    print("wrote", path, round(os.path.getsize(path) / 1e6, 2), "MB")


if __name__ == "__main__":
    run("conv_lenet_bbb", [1, 28, 28], 4, 3, "lenet_bbb_f32.npz", 0.2)
    run("conv_resnet_bbb", [1, 3, 32, 32], 2, 2, "resnet_bbb_f32.npz", 0.002)
