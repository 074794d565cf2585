#!/usr/bin/env python3
"""How many probabilities of the reference's OWN second run (another CPU code path: tests/golden/altref.py) lie outside the north-star tolerance
(1e-5 relative + 1e-6 absolute) of its first, recorded run -- for the two QAT ResNet fixtures whose tests carry the loose `2 x refspread.max_abs` floor.
RUNS ONLY IN THE BUILD CONTAINER (the alternative run imports /root/reference through the fixture generators).  The committed fixtures are read, not
rewritten; the counts go to tests/golden/qat_refspread_counts.json, and the GPU test bounds the build's count of outliers by the reference's own.

Usage: python tests/golden/make_golden_qat_counts.py
"""
import json
import os

import numpy as np

import altref

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [("make_golden_qat_mc.py", "resnet_mc_qat.npz"), ("make_golden_qat_mc.py", "resnet_sgld_qat.npz"), ("make_golden_qat.py", "resnet_bbb_qat.npz")]


def outside(a, b, rtol=1e-5, atol=1e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return int((np.abs(a - b) > atol + rtol * np.abs(b)).sum())


def main():
    out = {}
    for script, name in CASES:
        rec = np.load(os.path.join(HERE, name))["probs"]
        alt = altref.run_alt(os.path.join(HERE, script), name)["probs"]
        assert alt.shape == rec.shape
        d = np.abs(alt.astype(np.float64) - rec)
        out[name] = {"n": int(rec.size), "n_outside_1e-5_1e-6": outside(alt, rec), "max_abs": float(d.max()), "mean_abs": float(d.mean()),
                     "argmax_equal": int((alt.argmax(-1) == rec.argmax(-1)).sum()), "rows": int(np.prod(rec.shape[:-1]))}
        print(name, out[name])
    json.dump(out, open(os.path.join(HERE, "qat_refspread_counts.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
