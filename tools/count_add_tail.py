#!/usr/bin/env python3
"""How often does ATen's scalar tail of quantized::add differ from the vector-body formula the oracle / HIP kernels use?

ATen's CPU kernel of quantized::add dequantises with fma(s, q, -z*s) in its vectorised body and with (q - z) * s in the
scalar tail (the last n mod VEC elements of every thread's chunk); the two forms differ on rounding ties.  The build uses
the vector-body form for every element.  This script measures, at the BASELINE size (B = 256), on
THIS host's vector width / thread count:
  (1) per BasicBlock: ATen's own Add output vs the oracle's qadd applied to ATen's own addends (isolated op difference);
  (2) end to end: ATen harness probabilities vs the oracle's at the same injected eps.
Runs on the CPU (torch quantised ops through oracle/fbgemm_baseline.py; no reference import, no GPU).

  python tools/count_add_tail.py [--samples 3] [--batch 256] [--threads N]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from fixtures import load_golden  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle.fbgemm_baseline import FbgemmResNetBBB  # noqa: E402


def weight_chain(a):
    """conv_q.py:113-118 / linear_q.py:86-91 on the 21 stochastic layers' own (mu_q, sigma_q): ATen's quantized::mul and quantized::add
    (tensors of 648 ... 331,776 elements: every one has a scalar tail on AVX-512 / AVX2, per thread chunk) against the oracle's
    vector-body formulas on the same injected eps.  `t` isolates the mul, `w` given an equal `t` the add."""
    from oracle.fbgemm_baseline import NOISE_SCALE, NOISE_ZERO_POINT
    g = load_golden(f"resnet_bbb_a7w{a.w_bits}.npz")
    st = g["state"]
    net = orc.Int8ResNetOracle(st, 7, 8)            # W8: the clamp is the identity, so the add's own output is compared
    fb = FbgemmResNetBBB(st, 7, 8)
    seed = 3
    n_t = n_w = n_el = 0
    per_layer = {}
    for s in range(a.samples):
        for i, (pfx, *_) in enumerate(net.table):
            L, F = net.layers[pfx], fb.layers[pfx]
            eps_ohwi = orc.fill_eps_i8(L.mu_q.size, seed, i, s).reshape(L.mu_q.shape)
            t_or, w_or = orc.sample_weights_i8(L.mu_q, L.sigma_q, eps_ohwi, L.sp, want_t=True)
            e = eps_ohwi.transpose(0, 3, 1, 2) if eps_ohwi.ndim == 4 else eps_ohwi
            e = torch.from_numpy(np.array(e, order="C", copy=True).reshape(-1)).view(tuple(F.std.shape))
            eps_q = torch.quantize_per_tensor(e, NOISE_SCALE, NOISE_ZERO_POINT, torch.qint8)
            t_at = torch.ops.quantized.mul(F.std, eps_q, F.s_m, F.z_m)
            w_at = torch.ops.quantized.add(F.weight, t_at, F.s_a, F.z_a)
            to_ohwi = (lambda v: v.transpose(0, 2, 3, 1)) if eps_ohwi.ndim == 4 else (lambda v: v)
            dt = int((to_ohwi(t_at.int_repr().numpy()) != t_or).sum())
            dw = int((to_ohwi(w_at.int_repr().numpy()) != w_or).sum())
            n_t += dt
            n_w += dw
            n_el += t_or.size
            if dt or dw:
                per_layer[pfx] = tuple(np.add(per_layer.get(pfx, (0, 0)), (dt, dw)))
    print(f"threads={torch.get_num_threads()} cpu_capability={torch.backends.cpu.get_cpu_capability()} S={a.samples} layers=21 (648 ... 331,776 elements)")
    print(f"weight chain: quantized::mul differs in {n_t}, quantized::add in {n_w} of {n_el} elements")
    for k, v in per_layer.items():
        print("  ", k, "mul", v[0], "add", v[1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--w-bits", type=int, default=8)
    ap.add_argument("--weights", action="store_true", help="the WEIGHT chain's quantized::mul / quantized::add (conv_q.py:118) instead of the activations' Add")
    a = ap.parse_args()
    if a.threads:
        torch.set_num_threads(a.threads)
    if a.weights:
        return weight_chain(a)
    g = load_golden(f"resnet_bbb_a7w{a.w_bits}.npz")
    st = g["state"]
    net = orc.Int8ResNetOracle(st, 7, a.w_bits)
    fb = FbgemmResNetBBB(st, 7, a.w_bits)
    x = torch.randn(a.batch, 3, 32, 32, generator=torch.Generator().manual_seed(2)).numpy()
    seed = 3
    tot_add = tot_el = 0
    worst_p = 0.0
    n_prob_diff = 0
    for s in range(a.samples):
        eps = {pfx: orc.fill_eps_i8(net.layers[pfx].mu_q.size, seed, i, s).reshape(net.layers[pfx].mu_q.shape)
               for i, (pfx, *_) in enumerate(net.table)}
        rec = {}
        p_at = fb.forward(x, eps, record=rec)
        p_or = net.forward(x, seed, s)
        worst_p = max(worst_p, float(np.abs(p_at - p_or).max()))
        n_prob_diff += int((p_at != p_or).sum())
        s_prev, z_prev = net.layers["layers.0."].s_y, net.layers["layers.0."].z_y
        for li in (3, 4, 5, 6):
            for bi in (0, 1):
                p = f"layers.{li}.{bi}."
                Lb = net.layers[p + "stem.3."]
                if (p + "shortcut.0.") in net.layers:
                    s_r, z_r = net.layers[p + "shortcut.0."].s_y, net.layers[p + "shortcut.0."].z_y
                else:
                    s_r, z_r = s_prev, z_prev
                sa, za = float(st[p + "add.add.scale"]), int(st[p + "add.add.zero_point"])
                mine = orc.qadd_relu(rec[p + "stem.3.out"], Lb.s_y, Lb.z_y, rec[p + "res"], s_r, z_r, sa, za, True, net.a_hi)
                d = int((mine != rec[p + "out"]).sum())
                tot_add += d
                tot_el += mine.size
                if d:
                    print(f"sample {s} {p}add: {d} / {mine.size} elements differ")
                s_prev, z_prev = sa, za
    print(f"threads={torch.get_num_threads()} cpu_capability={torch.backends.cpu.get_cpu_capability()} B={a.batch} S={a.samples} W{a.w_bits}")
    print(f"isolated Add differences: {tot_add} of {tot_el} elements ({tot_add / max(tot_el, 1):.3e})")
    print(f"end-to-end: {n_prob_diff} of {a.samples * a.batch * 10} probabilities differ, max |dp| = {worst_p:.3e}")


if __name__ == "__main__":
    main()
