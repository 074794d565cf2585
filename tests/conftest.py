import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name))
    state = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
    rec = {k[len("rec/"):]: d[k] for k in d.files if k.startswith("rec/")}
    meta = {k[len("meta."):]: (float(d[k]) if d[k].dtype.kind == "f" else int(d[k])) for k in d.files if k.startswith("meta.")}
    return dict(state=state, rec=rec, meta=meta, x=d["x"], probs=d["probs"], mean_probs=d["mean_probs"])


@pytest.fixture(scope="session", params=["resnet_bbb_a7w8.npz", "resnet_bbb_a7w4.npz"])
def golden(request):
    return load_golden(request.param)


@pytest.fixture(scope="session")
def golden_w8():
    return load_golden("resnet_bbb_a7w8.npz")


@pytest.fixture(scope="session")
def golden_lenet_mc():
    return load_golden("lenet_mc_a7w8.npz")


@pytest.fixture(scope="session")
def golden_mlp_f32():
    d = np.load(os.path.join(GOLDEN, "mlp_bbb_f32.npz"))
    return dict(state={k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}, x=d["x"], mu=d["mu"], var=d["var"],
                mean=d["mean"], pred_var=d["pred_var"], seed=int(d["meta.philox_seed"]), in_dim=int(d["meta.in_dim"]))


@pytest.fixture(scope="session")
def golden_ensemble():
    d = np.load(os.path.join(GOLDEN, "ensemble_resnet_a7w8.npz"))
    n = int(d["meta.members"])
    members = [{k[len(f"member{i}/"):]: d[k] for k in d.files if k.startswith(f"member{i}/")} for i in range(n)]
    rec = {k[len("rec/"):]: d[k] for k in d.files if k.startswith("rec/")}
    return dict(members=members, rec=rec, x=d["x"], probs=d["probs"], mean_probs=d["mean_probs"])


@pytest.fixture(scope="session")
def golden_lenet_bbb():
    return load_golden("lenet_bbb_a7w8.npz")


@pytest.fixture(scope="session")
def golden_mlp_bbb_q():
    d = np.load(os.path.join(GOLDEN, "mlp_bbb_a7w8.npz"))
    return dict(state={k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")},
                rec={k[len("rec/"):]: d[k] for k in d.files if k.startswith("rec/")}, x=d["x"], mu=d["mu"], var=d["var"],
                seed=int(d["meta.philox_seed"]))


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
