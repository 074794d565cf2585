"""Host check of a 'collapsed' requantisation: per channel, is the exact FBGEMM form
    q = clamp(rne(fma(bias_c, rcp, (float)acc) * mult))                       6 vector instructions per output
reproduced over the whole reachable accumulator range by the single-fma form
    q' = clamp(rne(fma((float)acc, mult, b'_c))),  b'_c = rn(rn(bias_c * rcp) * mult)   5 instructions ?
The epilogue is uniform over a layer's channels, so a layer qualifies only if ALL its channels do.  CPU only (numpy; the float32
operations are evaluated in extended precision and rounded once, as the hardware fma does).
    python tools/collapsed_requant_check.py            -> table per layer of the committed fixture (resnet_bbb_a7w8.npz)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
from fixtures import load_golden

LD = np.longdouble
f32 = np.float32


def fma32(a, b, c):           # float32 fma: the product of two float32 is exact in 64-bit-mantissa arithmetic, one rounding at the end
    return (LD(a) * LD(b) + LD(c)).astype(f32)


def rne_clamp(v, lo, hi):
    return np.clip(np.rint(v.astype(np.float64)), lo, hi)


def check_layer(bias, s_x, s_w, s_y, z_y, relu, a_hi=127):
    atw = f32(s_x) * f32(s_w)
    rcp = f32(1.0) / atw
    mult = atw / f32(s_y)
    lo, hi = (z_y if relu else 0) - z_y, a_hi - z_y
    bad = 0
    worst = 0
    for b in bias.astype(f32):
        centre = -float(b) * float(rcp)
        a0, a1 = int(np.floor(centre + (lo - 2) / float(mult))), int(np.ceil(centre + (hi + 2) / float(mult)))
        acc = np.arange(a0, a1 + 1, dtype=np.int64).astype(f32)          # |acc| < 2^24: exact
        exact = rne_clamp(fma32(b, rcp, acc) * mult, lo, hi)             # float32 * float32 -> float32 (numpy keeps float32)
        bp = (f32(b) * rcp) * mult
        coll = rne_clamp(fma32(acc, mult, bp), lo, hi)
        n = int((exact != coll).sum())
        bad += n > 0
        worst = max(worst, n)
    return bad, worst


def main():
    g = load_golden("resnet_bbb_a7w8.npz")
    st = g["state"]
    sc = lambda k: float(np.asarray(st[k]).reshape(-1)[0])
    # (layer, input scale, ReLU-fused) in execution order: models_bbb.py:146-256 after convert
    layers = [("layers.0", sc("quant.scale"), True)]
    s_in = sc("layers.0.scale")
    for li in (3, 4, 5, 6):
        for bi in (0, 1):
            p = "layers.%d.%d." % (li, bi)
            layers.append((p + "stem.0", s_in, True))
            layers.append((p + "stem.3", sc(p + "stem.0.scale"), False))
            if (p + "shortcut.0.scale") in st:
                layers.append((p + "shortcut.0", s_in, False))
            s_in = sc(p + "add.add.scale")
    tot_l = ok_l = tot_c = bad_c = 0
    print("%-28s %5s %8s %s" % ("layer", "cout", "bad ch.", "max differing accumulator values in one channel"))
    for n, s_x, relu in layers:
        bias = np.asarray(st[n + ".bias_"], np.float32).reshape(-1)
        bad, worst = check_layer(bias, s_x, sc(n + ".add_weight.scale"), sc(n + ".scale"), int(sc(n + ".zero_point")), relu)
        tot_l += 1; ok_l += bad == 0; tot_c += bias.size; bad_c += bad
        print("%-28s %5d %8d %d" % (n, bias.size, bad, worst))
    print("layers whose every channel collapses: %d of %d;  channels with at least one differing output: %d of %d" % (ok_l, tot_l, bad_c, tot_c))


if __name__ == "__main__":
    main()
