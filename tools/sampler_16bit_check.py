#!/usr/bin/env python3
"""CPU: could the int8 sampler draw TWO weights per 32-bit Philox word (an 8-bit alias column + an 8-bit threshold each)?  Compares the
shipped 24-bit-threshold alias table and an 8-bit-threshold one with the exact distribution of clamp(rne(N(0,1) / s_n)) (tools/make_eps_table.py):
table error, the systematic chi-square excess at 1.5e6 / 1.5e8 draws, mean / variance.  Result: profiles/r04_sampler_16bit_check.txt."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_eps_table as m  # noqa: E402

p = m.exact_probs()
n = 256


def alias_bits(bits):
    q = p * n
    thr = np.zeros(n, np.int64)
    alias = np.arange(n)
    small = [i for i in range(n) if q[i] < 1.0]
    large = [i for i in range(n) if q[i] >= 1.0]
    q = q.copy()
    while small and large:
        s_, l_ = small.pop(), large.pop()
        thr[s_] = int(round(q[s_] * (1 << bits)))
        alias[s_] = l_
        q[l_] -= (1.0 - q[s_])
        (small if q[l_] < 1.0 else large).append(l_)
    for i in small + large:
        thr[i] = (1 << bits)
        alias[i] = i
    pt = np.zeros(n)
    for c in range(n):
        keep = min(thr[c], 1 << bits) / float(1 << bits)
        pt[c] += keep / n
        pt[alias[c]] += (1 - keep) / n
    return pt


print("int8 weight-noise distribution: 256 values, P(0) = %.4f, P(+-127) = %.3e, smallest P = %.3e" % (p[128], p[255], p.min()))
for bits, name in ((24, "shipped: 8-bit column + 24-bit threshold (one weight per Philox word)"), (8, "proposed: 8-bit column + 8-bit threshold (two weights per word)")):
    pt = alias_bits(bits)
    err = np.abs(pt - p)
    print("\n" + name)
    print("   max |P_table - P| = %.3e (2^%.1f)" % (err.max(), math.log2(max(err.max(), 1e-300))))
    for N in (1.5e6, 1.5e8):
        mask = pt > 0
        excess = N * (((pt - p) ** 2)[mask] / pt[mask]).sum()       # expected EXCESS of Pearson's chi-square over its 255 degrees of freedom
        print("   N = %.1e draws: systematic chi-square excess %.1f on top of 255 +- %.0f" % (N, excess, math.sqrt(2 * 255)))
    k = np.arange(256) - 128.0
    print("   mean %.3e (exact %.3e), variance %.5f (exact %.5f, rel. error %.2e)" % ((pt * k).sum(), (p * k).sum(), (pt * k * k).sum(), (p * k * k).sum(),
                                                                                  abs((pt * k * k).sum() - (p * k * k).sum()) / (p * k * k).sum()))
