"""Readers of the committed fixtures (tests/golden/*.npz: inputs and outputs recorded from the reference).  Plain module, no
pytest: shared by tests/conftest.py, bench.py (synthetic workloads start from a recorded, calibrated state) and
__graft_entry__.smoke()."""
import os

import numpy as np

GOLDEN = os.path.dirname(os.path.abspath(__file__))


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name))
    state = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
    rec = {k[len("rec/"):]: d[k] for k in d.files if k.startswith("rec/")}
    meta = {k[len("meta."):]: (float(d[k]) if d[k].dtype.kind == "f" else int(d[k])) for k in d.files if k.startswith("meta.")}
    return dict(state=state, rec=rec, meta=meta, x=d["x"], probs=d["probs"], mean_probs=d["mean_probs"])


def load_ensemble_fixture(name="ensemble_resnet_a7w8.npz"):
    d = np.load(os.path.join(GOLDEN, name))
    n = int(d["meta.members"])
    members = [{k[len(f"member{i}/"):]: d[k] for k in d.files if k.startswith(f"member{i}/")} for i in range(n)]
    rec = {k[len("rec/"):]: d[k] for k in d.files if k.startswith("rec/")}
    return dict(members=members, rec=rec, x=d["x"], probs=d["probs"], mean_probs=d["mean_probs"])


def synth_ensemble_members(g, n):
    """`n` distinct member state dicts for size tests / benches of the SGHMC ensemble (BASELINE config 4: 16 members):
    the members recorded from the reference (fixture: 2) first, then deterministic perturbations of their int8 weights and
    biases (same qparams, so every member stays a valid converted network).  Data only; nothing is read at run time
    besides the committed fixture."""
    out = []
    for i in range(n):
        base = g["members"][i % len(g["members"])]
        if i < len(g["members"]):
            out.append(base)
            continue
        rng = np.random.default_rng(7000 + i)
        st = {}
        for k, v in base.items():
            v = np.asarray(v)
            if k.endswith(".weight") and v.dtype == np.int8:
                st[k] = np.clip(v.astype(np.int32) + rng.integers(-6, 7, v.shape), -128, 127).astype(np.int8)
            elif k.endswith(".bias") and v.size:
                st[k] = (v * (1.0 + 0.1 * rng.standard_normal(v.shape))).astype(np.float32)
            else:
                st[k] = v
        out.append(st)
    return out
