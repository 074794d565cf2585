#!/usr/bin/env python3
"""Golden vectors for BASELINE config 3: SGHMC-style ensemble of deterministic int8 ResNets ("samples" = members).
RUNS ONLY IN THE BUILD CONTAINER.  Imports the real reference `conv_resnet_sgld` (sgld/models_sgld.py: Network :214-288,
_ConvNetwork_ResNet :148-212) with training_mode=False and 2 members, prepare_model (QAT) -> calibrate -> convert, and
records each member's converted state (flat, reference key names) and the wrapper's round-robin outputs.
Output: tests/golden/ensemble_resnet_a7w8.npz (data only)."""
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

MEMBERS, B = 2, 4


def flat_member(member):
    out = {}
    for name, m in member.named_modules():
        cls = type(m).__name__
        if cls in ("Conv2d", "ConvReLU2d", "Linear", "LinearReLU") and hasattr(m, "scale") and callable(getattr(m, "weight", None)):
            w = m.weight()
            out[name + ".weight"] = w.int_repr().numpy()
            out[name + ".weight.q_scale"] = np.float64(w.q_scale())
            out[name + ".weight.q_zero_point"] = np.int64(w.q_zero_point())
            b = m.bias()
            if b is not None:
                out[name + ".bias"] = b.detach().numpy()
            out[name + ".scale"] = np.float64(m.scale)
            out[name + ".zero_point"] = np.int64(m.zero_point)
        elif cls == "QFunctional":
            out[name + ".scale"] = np.float64(m.scale)
            out[name + ".zero_point"] = np.int64(m.zero_point)
        elif cls == "Quantize":
            out[name + ".scale"] = m.scale.numpy()
            out[name + ".zero_point"] = m.zero_point.numpy()
    return out


def main():
    from src.models import ModelFactory
    import src.quant_utils as qu
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, model="conv_resnet_sgld", q=True, at=True,
                                 samples=MEMBERS, task="classification")
    torch.manual_seed(1)
    net = ModelFactory.get_model("conv_resnet_sgld", [1, 3, 32, 32], 10, True, args, training_mode=False)
    g = torch.Generator().manual_seed(7)
    for m in net.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)):
            fan_in = m.weight[0].numel()
            m.weight.data = torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5
        elif isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = torch.rand(m.weight.shape, generator=g) + 0.5
            m.bias.data.zero_()
    qu.prepare_model(net, args)
    xcal = torch.randn(32, 3, 32, 32, generator=g)
    net.train()
    for _ in range(2 * MEMBERS):
        net(xcal)                       # round-robin: every member sees 2 calibration batches
    net.eval()
    with torch.no_grad():
        for _ in range(MEMBERS):
            net(xcal)
    qu.convert(net)
    net.eval()
    net.counter = 0
    x = torch.randn(B, 3, 32, 32, generator=g)
    rec = {}
    hooks = []
    mem0 = net.ensemble[0]
    mods = dict(mem0.named_modules())

    def mk(name):
        def hook(_m, _i, o):
            a = o.int_repr().numpy()
            rec[name + ".out"] = np.minimum(np.ascontiguousarray(a.transpose(0, 2, 3, 1)) if a.ndim == 4 else a, 127).astype(np.uint8)
        return hook

    for n in ["layers.0", "layers.3.1", "layers.4.0", "layers.6.1", "layers.9"]:
        hooks.append(mods[n].register_forward_hook(mk(n)))
    with torch.no_grad():
        probs = [net(x).numpy().copy() for _ in range(MEMBERS)]
        for h in hooks:
            h.remove()
        probs += [net(x).numpy().copy()]          # the counter wrapped: member 0 again
    assert np.array_equal(probs[0], probs[MEMBERS])
    probs = np.stack(probs[:MEMBERS])
    mean = torch.stack([torch.from_numpy(p) for p in probs], dim=1).mean(dim=1).numpy()
    out = {"x": x.numpy(), "probs": probs, "mean_probs": mean, "meta.members": np.int64(MEMBERS), "meta.a_bits": np.int64(7),
           "meta.w_bits": np.int64(8)}
    for i, mem in enumerate(net.ensemble):
        st = flat_member(mem)
        out.update({f"member{i}/" + k: v for k, v in st.items()})
    out.update({"rec/" + k: v for k, v in rec.items()})
    # self-check with the oracle (deterministic members: no eps)
    st0 = {k[len("member0/"):]: v for k, v in out.items() if k.startswith("member0/")}
    o = orc.Int8ResNetDetOracle(st0, 7)
    orec = {}
    p0 = o.forward(x.numpy(), record=orec)
    bad = sum(int((orec[k] != v.reshape(orec[k].shape)).sum()) for k, v in rec.items())
    rel = np.abs(p0 - probs[0]).max() / probs[0].max()
    print(f"oracle vs reference (ensemble member 0): {bad} mismatching integer elements; probs max rel err {rel:.2e}")
    assert bad == 0 and rel < 1e-5
    path = os.path.join(HERE, "ensemble_resnet_a7w8.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")
    # one checkpoint file per member, as an SGHMC run leaves them (src/utils.py:84-93 saves the converted copy of the training
    # wrapper, whose single net is `main_net`; models_sgld.py:245-261 finds `weights_<n>.pt` and src/utils.py:112-123 strips the prefix)
    ck = os.path.join(HERE, "ensemble_ckpt")
    os.makedirs(ck, exist_ok=True)
    for i, mem in enumerate(net.ensemble):
        torch.save({"main_net." + k: v for k, v in mem.state_dict().items()}, os.path.join(ck, f"weights_{i + 1}.pt"))
    print("wrote", ck, [round(os.path.getsize(os.path.join(ck, f)) / 1e6, 2) for f in sorted(os.listdir(ck))], "MB")


if __name__ == "__main__":
    main()
