import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)
from fixtures import load_golden, load_ensemble_fixture, synth_ensemble_members      # noqa: E402,F401  (re-exported for the tests)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The reference's bit-width sweep (experiments/run_all_quant.sh:11-37: W in 3..8 at A7, A in 3..6 at W8; src/utils.py:18-20,
# asserts src/quant_utils.py:120-121): one recorded fixture per point (make_golden.py --a-bits / --w-bits; B = 4, S = 3).
SWEEP = [(7, 8), (7, 4), (7, 3), (7, 5), (7, 6), (7, 7), (3, 8), (4, 8), (5, 8), (6, 8)]


@pytest.fixture(scope="session", params=["resnet_bbb_a%dw%d.npz" % aw for aw in SWEEP], ids=["a%dw%d" % aw for aw in SWEEP])
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
    return _mlp_f32(d)


def _mlp_f32(d):
    """Float BBB MLP fixture (make_golden_mlp_f32.py).  mu_atol: the absolute floor that goes with the 1e-5 relative bound on mu -- FOUR times
    the reference's own distance from itself between its AVX-512 and AVX2 code paths (`refspread.mu_abs`, 0.7 - 1.4e-6 on outputs of range
    5 - 7: the reference and the build are each one fp32 summation order away from the exact value; measured on the GPU 1.0 - 2.4e-6)."""
    return dict(state={k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}, x=d["x"], mu=d["mu"], var=d["var"],
                mean=d["mean"], pred_var=d["pred_var"], seed=int(d["meta.philox_seed"]), in_dim=int(d["meta.in_dim"]),
                mu_atol=4.0 * float(d["refspread.mu_abs"]), refspread_var_rel=float(d["refspread.var_rel"]))


@pytest.fixture(scope="session", params=[1, 4, 6, 8, 11], ids=lambda d: "in%d" % d)
def golden_mlp_f32_width(request):
    """BASELINE config 0 at SURVEY 8(d) C1's other input widths (13 is golden_mlp_f32)."""
    return _mlp_f32(np.load(os.path.join(GOLDEN, "mlp_bbb_f32_in%d.npz" % request.param)))


@pytest.fixture(scope="session")
def golden_ensemble():
    return load_ensemble_fixture()


@pytest.fixture(scope="session")
def golden_lenet_bbb():
    return load_golden("lenet_bbb_a7w8.npz")


@pytest.fixture(scope="session")
def golden_mlp_bbb_q():
    d = np.load(os.path.join(GOLDEN, "mlp_bbb_a7w8.npz"))
    return dict(state={k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")},
                rec={k[len("rec/"):]: d[k] for k in d.files if k.startswith("rec/")}, x=d["x"], mu=d["mu"], var=d["var"],
                seed=int(d["meta.philox_seed"]))


def _npz(name):
    d = np.load(os.path.join(GOLDEN, name))
    out = {k: d[k] for k in d.files if "/" not in k and not k.startswith("meta.") and not k.startswith("refspread.")}
    out["state"] = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
    out["rec"] = {k[len("rec/"):]: d[k] for k in d.files if k.startswith("rec/")}
    out["meta"] = {k[len("meta."):]: (float(d[k]) if d[k].dtype.kind == "f" else int(d[k])) for k in d.files if k.startswith("meta.")}
    out["refspread"] = {k[len("refspread."):]: float(d[k]) for k in d.files if k.startswith("refspread.")}
    return out


@pytest.fixture(scope="session")
def golden_mlp_mc_q():
    """`linear_mc` int8 A7/W8 (tests/golden/make_golden_linear_mc.py)."""
    return _npz("mlp_mc_a7w8.npz")


@pytest.fixture(scope="session", params=[("mlp_mc_f32.npz", "linear_mc"), ("lenet_mc_f32.npz", "conv_lenet_mc"), ("resnet_mc_f32.npz", "conv_resnet_mc")],
                ids=["mlp", "lenet", "resnet"])
def golden_mc_f32(request):
    """The float MC-Dropout graphs (tests/golden/make_golden_mc_f32.py): (fixture, factory name)."""
    g = _npz(request.param[0])
    g["model"] = request.param[1]
    return g
