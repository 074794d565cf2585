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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--w-bits", type=int, default=8)
    a = ap.parse_args()
    if a.threads:
        torch.set_num_threads(a.threads)
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
