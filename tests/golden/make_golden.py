#!/usr/bin/env python3
"""Golden-vector generator.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

Imports the real reference (martinferianc/quantised-bayesian-nets) through
ref_shim.py, builds `conv_resnet_bbb`, calibrates it exactly the way the
reference's training script would (prepare_model -> 1 train-mode + 3 eval-mode
forwards -> convert, quant_utils.py:62-147), then runs the reference's own int8
stochastic forward with the weight noise INJECTED from the build's Philox stream
(oracle.fill_eps_i8: the int8 mode draws eps_q directly; the injected fp32 eps = eps_q * s_n quantises back to it), and records:

  * the converted model's state_dict in flat numpy form (reference key names),
  * the input batch,
  * for MC sample 0: every stochastic layer's sampled weight W_q and output,
    every block output, avgpool output,
  * per-sample softmax probabilities for S samples and their mean
    (= experiments/utils.py:342-355).

Output: tests/golden/resnet_bbb_a{A}w{W}.npz   (data only: inputs + expected outputs)

Usage: python tests/golden/make_golden.py [--w-bits 8] [--batch 4] [--samples 3]
"""
import argparse
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

PARAM_SEED, INPUT_SEED, PHILOX_SEED = 1, 2, 3


def build_reference_model(a_bits, w_bits, batch, before_convert=None):
    from src.models import ModelFactory
    import src.quant_utils as qu
    from src.models.stochastic.bbb.conv import Conv2d as Conv2dBBB
    from src.models.stochastic.bbb.linear import Linear as LinearBBB

    args = types.SimpleNamespace(sigma_prior=-2.0, activation_precision=a_bits, weight_precision=w_bits,
                                 model="conv_resnet_bbb", q=True, at=True, samples=4, task="classification")
    torch.manual_seed(PARAM_SEED)
    model = ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, args)
    # SURVEY 8(d) config C3 initialisation (the reference's own init, U(-0.01,0.01)/rho=-10, gives
    # degenerate activations): mu ~ N(0, sqrt(2/fan_in)), rho = -3, BN gamma ~ U(0.5,1.5), beta = 0.
    g = torch.Generator().manual_seed(PARAM_SEED)
    for m in model.modules():
        if isinstance(m, (Conv2dBBB, LinearBBB)):
            fan_in = m.weight[0].numel()
            m.weight.data = torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5
            m.std.data.fill_(-3.0)
        elif isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = torch.rand(m.weight.shape, generator=g) + 0.5
            m.bias.data.zero_()
    qu.prepare_model(model, args)
    gx = torch.Generator().manual_seed(INPUT_SEED)
    x_cal = torch.randn(32, 3, 32, 32, generator=gx)
    torch.manual_seed(PARAM_SEED + 100)
    model.train()
    model(x_cal)
    model.eval()
    with torch.no_grad():
        for _ in range(3):
            model(x_cal)
    if before_convert is not None:
        before_convert(model)          # the prepared (QAT) model, calibrated, right before the reference converts it
    qu.convert(model)
    model.eval()
    return model, args


def flat_state(model):
    out = {}
    for k, v in model.state_dict().items():
        if v is None:
            continue
        if isinstance(v, torch.Tensor) and v.is_quantized:
            out[k] = v.int_repr().numpy()
            out[k + ".q_scale"] = np.float64(v.q_scale())
            out[k + ".q_zero_point"] = np.int64(v.q_zero_point())
        elif isinstance(v, torch.Tensor):
            out[k] = v.detach().numpy()
        else:
            out[k] = np.asarray(v)
    return out


class Injector:
    """Replaces Tensor.normal_ during the reference forward so that the i-th weight-noise draw of
    a forward pass (draw order = execution order) returns the build's eps."""

    def __init__(self, table_shapes):
        # the injected eps = (float)eps_q * s_n must come back as eps_q from the reference's own quantize_per_tensor
        k = np.arange(-128, 128).astype(np.int8)
        back = torch.quantize_per_tensor(torch.from_numpy(orc.eps_from_eps_q(k)), orc.NOISE_SCALE, 0, torch.qint8).int_repr().numpy()
        assert np.array_equal(back, k), "eps_q -> eps -> quantize_per_tensor is not the identity"
        self.shapes = table_shapes
        self.queue = None
        self.orig = torch.Tensor.normal_

    def arm(self, seed, sample):
        self.queue = []
        for lid, shp in enumerate(self.shapes):
            n = int(np.prod(shp))
            e = orc.fill_eps_i8(n, seed, lid, sample)        # int8 noise stream: (float)eps_q * s_n, eps_q from the alias sampler
            if len(shp) == 4:   # our stream is defined on the OHWI flattening
                o, i, kh, kw = shp
                e = e.reshape(o, kh, kw, i).transpose(0, 3, 1, 2)
            self.queue.append(np.ascontiguousarray(e.reshape(shp)))
        self.pos = 0

    def __enter__(self):
        inj = self

        def normal_(t, mean=0, std=1, *, generator=None):
            e = inj.queue[inj.pos]
            assert tuple(t.shape) == e.shape, (tuple(t.shape), e.shape, inj.pos)
            inj.pos += 1
            t.copy_(torch.from_numpy(e))
            return t

        torch.Tensor.normal_ = normal_
        return self

    def __exit__(self, *a):
        torch.Tensor.normal_ = self.orig
        assert self.pos == len(self.queue), (self.pos, len(self.queue))


def nhwc(t):
    a = t.int_repr().numpy() if t.is_quantized else t.numpy()
    return np.ascontiguousarray(a.transpose(0, 2, 3, 1)) if a.ndim == 4 else np.ascontiguousarray(a)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a-bits", type=int, default=7)
    ap.add_argument("--w-bits", type=int, default=8)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--samples", type=int, default=3)
    a = ap.parse_args()

    model, args = build_reference_model(a.a_bits, a.w_bits, a.batch)
    state = flat_state(model)
    table = orc.resnet_layer_table()
    mods = dict(model.named_modules())
    shapes = [tuple(mods[p[:-1]].weight.shape) for p, *_ in table]
    a_hi = orc.UINT_BOUNDS[a.a_bits][1]

    gx = torch.Generator().manual_seed(INPUT_SEED + 1)
    x = torch.randn(a.batch, 3, 32, 32, generator=gx)

    rec = {}
    import src.models.stochastic.bbb.quantized.conv_q as conv_q
    import src.models.stochastic.bbb.quantized.linear_q as linear_q
    wq_log = []
    orig_cw = conv_q.clamp_weight

    def logging_clamp_weight(w, args_):
        out = orig_cw(w, args_)
        wq_log.append(out.int_repr().numpy())
        return out

    conv_q.clamp_weight = logging_clamp_weight
    linear_q.clamp_weight = logging_clamp_weight

    hooks = []
    names = [p[:-1] for p, *_ in table]
    blocks = [f"layers.{li}.{bi}" for li in (3, 4, 5, 6) for bi in (0, 1)]

    def mk(name):
        def hook(_m, _i, o):
            v = nhwc(o)
            rec[name + ".out"] = np.minimum(v, a_hi).astype(np.uint8)   # + clamp_activation (src/utils.py:25-30)
        return hook

    for n in names + blocks + ["quant", "layers.7"]:
        hooks.append(mods[n].register_forward_hook(mk(n)))

    inj = Injector(shapes)
    probs = []
    with torch.no_grad():
        for s in range(a.samples):
            inj.arm(PHILOX_SEED, s)
            wq_log.clear()
            with inj:
                p = model(x)
            probs.append(p.numpy().copy())
            if s == 0:
                for h in hooks:
                    h.remove()
                for lid, n in enumerate(names):
                    w = wq_log[lid]
                    rec[n + ".w_q"] = np.ascontiguousarray(w.transpose(0, 2, 3, 1)) if w.ndim == 4 else w
    probs = np.stack(probs, 0)
    mean = torch.stack([torch.from_numpy(p) for p in probs], dim=1).mean(dim=1).numpy()   # experiments/utils.py:355

    # ---- self-check: the oracle must reproduce the reference bit-for-bit before the fixture is written
    net = orc.Int8ResNetOracle(state, a.a_bits, a.w_bits)
    orec = {}
    op0 = net.forward(x.numpy(), PHILOX_SEED, 0, record=orec)
    bad = 0
    for n in names:
        bad += int((orec[n + ".w_q"] != rec[n + ".w_q"]).sum())
        bad += int((orec[n + ".out"] != rec[n + ".out"].reshape(orec[n + ".out"].shape)).sum())
    for b in blocks:
        bad += int((orec[b + ".out"] != rec[b + ".out"]).sum())
    bad += int((orec["quant.out"] != rec["quant.out"]).sum())
    bad += int((orec["avgpool.out"] != rec["layers.7.out"]).sum())
    rel = np.abs(op0 - probs[0]).max() / probs[0].max()
    print(f"oracle vs reference: {bad} mismatching integer elements; probs max rel err {rel:.2e}")
    assert bad == 0 and rel < 1e-5

    out = {"meta.a_bits": np.int64(a.a_bits), "meta.w_bits": np.int64(a.w_bits), "meta.philox_seed": np.int64(PHILOX_SEED),
           "x": x.numpy(), "probs": probs, "mean_probs": mean}
    out.update({"state/" + k: v for k, v in state.items()})
    out.update({"rec/" + k: v for k, v in rec.items()})
    path = os.path.join(HERE, f"resnet_bbb_a{a.a_bits}w{a.w_bits}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()
