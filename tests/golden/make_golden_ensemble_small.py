#!/usr/bin/env python3
"""Golden vectors for the two small SGHMC ensemble templates (`conv_lenet_sgld`, `linear_sgld`; reference
sgld/models_sgld.py:13-97, wrapper :214-288).  RUNS ONLY IN THE BUILD CONTAINER.
Imports the real reference, builds the evaluation-mode `Network` (args.samples deterministic members), calibrates
(prepare_model -> train + eval forwards), converts, and records every member's converted state, the input, layer outputs of
member 0 and every member's output.  Output: tests/golden/ensemble_{lenet,mlp}_a7w8.npz (data only)."""
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

from make_golden_ensemble import flat_member  # noqa: E402

MEMBERS = 2


def build(model_name, input_size, output_size, task, xcal):
    from src.models import ModelFactory
    import src.quant_utils as qu
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, model=model_name, q=True, at=True, samples=MEMBERS, task=task)
    torch.manual_seed(1)
    net = ModelFactory.get_model(model_name, input_size, output_size, True, args, training_mode=False)
    g = torch.Generator().manual_seed(11)
    for m in net.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)):
            fan_in = m.weight[0].numel()
            m.weight.data = torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5
            if m.bias is not None:
                m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.1
    qu.prepare_model(net, args)
    net.train()
    for _ in range(2 * MEMBERS):
        net(xcal)
    net.eval()
    with torch.no_grad():
        for _ in range(MEMBERS):
            net(xcal)
    qu.convert(net)
    net.eval()
    net.counter = 0
    return net, args


def record(net, x, names):
    rec, hooks = {}, []
    mods = dict(net.ensemble[0].named_modules())

    def mk(name):
        def hook(_m, _i, o):
            a = o.int_repr().numpy()
            rec[name + ".out"] = np.minimum(np.ascontiguousarray(a.transpose(0, 2, 3, 1)) if a.ndim == 4 else a, 127).astype(np.uint8)
        return hook

    for n in names:
        hooks.append(mods[n].register_forward_hook(mk(n)))
    outs = []
    with torch.no_grad():
        for i in range(MEMBERS):
            outs.append(net(x))
            if i == 0:
                for h in hooks:
                    h.remove()
    return outs, rec


def main():
    g = torch.Generator().manual_seed(5)
    net, args = build("conv_lenet_sgld", [1, 1, 28, 28], 10, "classification", torch.rand(32, 1, 28, 28, generator=g))
    x = torch.rand(4, 1, 28, 28, generator=g)
    outs, rec = record(net, x, ["quant", "layers.0", "layers.1", "layers.2", "layers.3", "layers.5", "layers.7"])
    out = {"x": x.numpy(), "probs": np.stack([o.numpy() for o in outs]), "meta.members": np.int64(MEMBERS)}
    for i, mem in enumerate(net.ensemble):
        out.update({f"member{i}/" + k: v for k, v in flat_member(mem).items()})
    out.update({"rec/" + k: v for k, v in rec.items()})
    np.savez_compressed(os.path.join(HERE, "ensemble_lenet_a7w8.npz"), **out)
    print("wrote ensemble_lenet_a7w8.npz", round(os.path.getsize(os.path.join(HERE, "ensemble_lenet_a7w8.npz")) / 1e6, 2), "MB;",
          "members differ:", float(np.abs(out["probs"][0] - out["probs"][1]).max()))

    net, args = build("linear_sgld", [13], 1, "regression", torch.randn(64, 13, generator=g))
    x = torch.randn(200, 13, generator=g)
    outs, rec = record(net, x, ["quant", "layers.0", "layers.2", "layers.4", "mu", "log_var"])
    out = {"x": x.numpy(), "mu": np.stack([o[0].numpy() for o in outs]), "var": np.stack([o[1].numpy() for o in outs]), "meta.members": np.int64(MEMBERS)}
    for i, mem in enumerate(net.ensemble):
        out.update({f"member{i}/" + k: v for k, v in flat_member(mem).items()})
    out.update({"rec/" + k: v for k, v in rec.items()})
    np.savez_compressed(os.path.join(HERE, "ensemble_mlp_a7w8.npz"), **out)
    print("wrote ensemble_mlp_a7w8.npz", round(os.path.getsize(os.path.join(HERE, "ensemble_mlp_a7w8.npz")) / 1e6, 3), "MB")


if __name__ == "__main__":
    main()
