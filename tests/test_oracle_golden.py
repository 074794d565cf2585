"""The oracle (oracle/qbnn_oracle.c) against the golden vectors recorded from the real reference
(tests/golden/make_golden.py).  Integer tensors must match bit-for-bit; fp32 probabilities to 1e-5 rel."""
import os

import numpy as np
import pytest

from oracle import oracle as orc


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert [hex(v) for v in orc.philox([0, 0, 0, 0], [0, 0])] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    assert [hex(v) for v in orc.philox([0xffffffff] * 4, [0xffffffff] * 2)] == ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    assert [hex(v) for v in orc.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])] == \
        ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


def test_normal_stream_is_standard_normal():
    e = orc.fill_normal(1 << 20, 3, 5, 7).astype(np.float64)
    assert abs(e.mean()) < 4e-3 and abs(e.std() - 1) < 4e-3 and abs((e ** 4).mean() - 3) < 0.05
    # ragged tail: n not a multiple of 4 gives a prefix of the same stream
    assert np.array_equal(orc.fill_normal(1003, 3, 5, 7), orc.fill_normal(1 << 20, 3, 5, 7)[:1003])
    # streams are keyed by (seed, layer, sample)
    assert not np.array_equal(orc.fill_normal(64, 3, 5, 7), orc.fill_normal(64, 3, 5, 8))
    assert not np.array_equal(orc.fill_normal(64, 3, 5, 7), orc.fill_normal(64, 3, 6, 7))
    assert not np.array_equal(orc.fill_normal(64, 3, 5, 7), orc.fill_normal(64, 4, 5, 7))


def test_eps_quantisation_ties_round_half_even():
    # reference: quantize_per_tensor(eps, 3/127, 0, qint8) == clamp(rne(eps * (1/s)), -128, 127)  (conv_q.py:115)
    s = np.float32(orc.NOISE_SCALE)
    e = np.array([0.5 * s, 1.5 * s, -0.5 * s, 2.5 * s, 10.0, -10.0, 3.0], np.float32)
    q = orc.quantize_eps(e)
    inv = np.float32(1.0) / s
    ref = np.clip(np.rint(e * inv), -128, 127).astype(np.int8)
    assert np.array_equal(q, ref) and q[4] == 127 and q[5] == -128


def test_layers_and_graph_bit_exact(golden):
    net = orc.Int8ResNetOracle(golden["state"], golden["meta"]["a_bits"], golden["meta"]["w_bits"])
    assert net.n_weights() == 1571592
    rec, orec = golden["rec"], {}
    p0 = net.forward(golden["x"], golden["meta"]["philox_seed"], 0, record=orec)
    lo, hi = orc.INT_BOUNDS[golden["meta"]["w_bits"]]
    for pfx, *_ in net.table:
        n = pfx[:-1]
        assert np.array_equal(orec[pfx + "w_q"], rec[n + ".w_q"]), n
        assert rec[n + ".w_q"].min() >= lo and rec[n + ".w_q"].max() <= hi
        assert np.array_equal(orec[pfx + "out"], rec[n + ".out"].reshape(orec[pfx + "out"].shape)), n
    for li in (3, 4, 5, 6):
        for bi in (0, 1):
            assert np.array_equal(orec[f"layers.{li}.{bi}.out"], rec[f"layers.{li}.{bi}.out"])
    assert np.array_equal(orec["quant.out"], rec["quant.out"])
    assert np.array_equal(orec["avgpool.out"], rec["layers.7.out"])
    np.testing.assert_allclose(p0, golden["probs"][0], rtol=1e-5, atol=1e-8)


def test_mc_reduction_matches_reference(golden):
    net = orc.Int8ResNetOracle(golden["state"], golden["meta"]["a_bits"], golden["meta"]["w_bits"])
    S = golden["probs"].shape[0]
    mean, ps = net.mc_predict(golden["x"], S, golden["meta"]["philox_seed"])
    np.testing.assert_allclose(ps, golden["probs"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(mean, golden["mean_probs"], rtol=1e-5, atol=1e-8)
    # stochastic: samples differ
    assert np.abs(ps[0] - ps[1]).max() > 1e-3


def test_injected_eps_equals_philox_path(golden_w8):
    g = golden_w8
    net = orc.Int8ResNetOracle(g["state"], 7, 8)
    eps = {pfx: orc.fill_eps_i8(net.layers[pfx].mu_q.size, 3, i, 1).reshape(net.layers[pfx].mu_q.shape)
           for i, (pfx, *_) in enumerate(net.table)}
    a = net.forward(g["x"], 3, 1, eps=eps)
    b = net.forward(g["x"], 3, 1)
    assert np.array_equal(a, b)


def test_lenet_mc_dropout_bit_exact(golden_lenet_mc):
    """BASELINE config 2: the quantised BernoulliDropout chain + deterministic int8 LeNet against the reference."""
    g = golden_lenet_mc
    net = orc.Int8LeNetMCOracle(g["state"], 7)
    orec = {}
    p0 = net.forward(g["x"], g["meta"]["philox_seed"], 0, record=orec)
    for k, v in g["rec"].items():
        assert np.array_equal(orec[k].reshape(v.shape), v), k
    np.testing.assert_allclose(p0, g["probs"][0], rtol=1e-5, atol=1e-8)
    mean, ps = net.mc_predict(g["x"], g["probs"].shape[0], g["meta"]["philox_seed"])
    np.testing.assert_allclose(ps, g["probs"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(mean, g["mean_probs"], rtol=1e-5, atol=1e-8)
    # about p of the channels are dropped (mask -> zero point)
    z4 = int(g["state"]["layers.4.mul_mask.zero_point"])
    dropped = (g["rec"]["layers.4.out"].reshape(g["x"].shape[0], -1, 50) == z4).all(axis=1).mean()
    assert 0.02 < dropped < 0.5


def test_float_bbb_mlp_matches_reference(golden_mlp_f32):
    """BASELINE config 0 (fp32 Bayes-by-backprop MLP, 10 samples): oracle vs the reference, 1e-5 relative."""
    g = golden_mlp_f32
    net = orc.F32MLPOracle(g["state"])
    for s in (0, 3, 9):
        mu, var = net.forward(g["x"], g["seed"], s)
        np.testing.assert_allclose(mu, g["mu"][s], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(var, g["var"][s], rtol=1e-5, atol=1e-8)
    mean, pv = net.mc_predict(g["x"], g["mu"].shape[0], g["seed"])
    np.testing.assert_allclose(mean, g["mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pv, g["pred_var"], rtol=1e-4, atol=1e-7)


def test_ensemble_members_match_reference(golden_ensemble):
    """BASELINE config 3: deterministic int8 ResNet members ("samples" = members), wrapper softmax + mean."""
    g = golden_ensemble
    ps = []
    for i, st in enumerate(g["members"]):
        o = orc.Int8ResNetDetOracle(st, 7)
        rec = {}
        ps.append(o.forward(g["x"], record=rec))
        if i == 0:
            for k, v in g["rec"].items():
                assert np.array_equal(rec[k], v.reshape(rec[k].shape)), k
    np.testing.assert_allclose(np.stack(ps), g["probs"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(np.stack(ps).mean(0), g["mean_probs"], rtol=1e-5, atol=1e-8)


def test_small_bbb_int8_graphs_bit_exact(golden_lenet_bbb, golden_mlp_bbb_q):
    """conv_lenet_bbb and linear_bbb (q=True) int8 against the reference."""
    g = golden_lenet_bbb
    net = orc.Int8LeNetBBBOracle(g["state"], 7, 8)
    rec = {}
    p0 = net.forward(g["x"], g["meta"]["philox_seed"], 0, record=rec)
    for k, v in g["rec"].items():
        assert np.array_equal(rec[k].reshape(v.shape), v), k
    np.testing.assert_allclose(p0, g["probs"][0], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(net.forward(g["x"], g["meta"]["philox_seed"], 2), g["probs"][2], rtol=1e-5, atol=1e-8)
    m = golden_mlp_bbb_q
    net = orc.Int8MLPBBBOracle(m["state"], 7, 8)
    rec = {}
    mu, var = net.forward(m["x"], m["seed"], 0, record=rec)
    for k, v in m["rec"].items():
        assert np.array_equal(rec[k].reshape(v.shape), v), k
    np.testing.assert_allclose(mu, m["mu"][0], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(var, m["var"][0], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("name,kind", [("lenet_bbb_f32.npz", "lenet"), ("resnet_bbb_f32.npz", "resnet")])
def test_float_bbb_conv_graphs_match_reference(name, kind):
    """float BBB conv graphs (eval branch, bbb/conv.py:33-39): oracle vs the reference's per-sample softmax
    outputs with the same injected eps; fp32 tolerance 1e-5 relative (+1e-6 absolute on probabilities)."""
    from conftest import load_golden
    g = load_golden(name)
    net = orc.F32ConvOracle(g["state"])
    fwd = net.lenet if kind == "lenet" else net.resnet
    for s in range(g["probs"].shape[0]):
        np.testing.assert_allclose(fwd(g["x"], g["meta"]["philox_seed"], s), g["probs"][s], rtol=1e-5, atol=1e-6)


def test_ema_observer_matches_torch_fake_quantize():
    """Row a2 building block: MovingAverageMinMaxObserver + fake_quantize_per_tensor_affine restated (oracle.EmaObserver)
    against torch's own FakeQuantize module over a sequence of batches (the EMA state matters)."""
    import torch
    from torch.ao.quantization import FakeQuantize, MovingAverageMinMaxObserver
    for (qmin, qmax, dtype) in ((0, 127, torch.quint8), (-128, 127, torch.qint8), (-8, 7, torch.qint8)):
        fq = FakeQuantize(observer=MovingAverageMinMaxObserver, quant_min=qmin, quant_max=qmax, dtype=dtype, qscheme=torch.per_tensor_affine)
        ob = orc.EmaObserver(qmin, qmax)
        rng = np.random.RandomState(qmax)
        for i in range(6):
            x = (rng.randn(4, 50).astype(np.float32) * (1 + i)) + np.float32(0.3 * i)
            want = fq(torch.from_numpy(x)).numpy()
            got = ob.fake_quant(x)
            np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("name,kind", [("mlp_bbb_qat.npz", "mlp"), ("lenet_bbb_qat.npz", "lenet"), ("resnet_bbb_qat.npz", "resnet")])
def test_qat_eval_with_live_observers_matches_reference(name, kind):
    """prepared (QAT) BBB models in eval mode, observers live (conv_qat.py:26-49,139-167; linear_qat.py:18-41):
    the oracle, run sample after sample, against S reference forwards with the same injected eps, and the observers'
    final (min, max) against the reference's."""
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
    st = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
    net = orc.QATOracle(st)
    seed = int(d["meta.philox_seed"])
    if kind == "mlp":
        for s in range(d["mu"].shape[0]):
            mu, var = net.mlp(d["x"], seed, s)
            np.testing.assert_allclose(mu, d["mu"][s], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(var, d["var"][s], rtol=1e-5, atol=1e-8)
    else:
        fwd = net.lenet if kind == "lenet" else net.resnet
        for s in range(d["probs"].shape[0]):
            np.testing.assert_allclose(fwd(d["x"], seed, s), d["probs"][s], rtol=1e-5, atol=2e-6)
    for k, ob in net.obs.items():
        np.testing.assert_allclose(ob.state[0], d["final/" + k + ".activation_post_process.min_val"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(ob.state[1], d["final/" + k + ".activation_post_process.max_val"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name,kind", [("mlp_mc_qat.npz", "mlp_mc"), ("lenet_mc_qat.npz", "lenet_mc"), ("resnet_mc_qat.npz", "resnet_mc"),
                                       ("resnet_sgld_qat.npz", "resnet_p")])
def test_qat_eval_of_the_non_bbb_graphs_matches_reference(name, kind):
    """SURVEY 8(f).3 widened to quant_utils.prepare_model's `prepare_qat` branch (:139-140): the MC-Dropout graphs (FakeQuantize on the dropout's
    mul_mask, mcdropout/dropout.py:9-40) and the SGHMC member template, eval mode, live observers -- the oracle sample after sample against S
    reference forwards with the same injected masks.  Tolerance: 1e-5 relative + twice the reference's own distance from itself on another
    CPU code path (`refspread.max_abs`: 0 / 4.5e-8 for the MLP / LeNet; 3.9e-4 / 5.2e-4 for the ResNets, where a handful of activations sit within
    fp32 summation noise of a quantisation step and round the other way)."""
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
    st = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
    net = orc.QATMCOracle(st, prefix="main_net." if kind == "resnet_p" else "")
    seed = int(d["meta.philox_seed"])
    atol = max(2e-6, 2.0 * float(d["refspread.max_abs"]))
    if kind == "mlp_mc":
        for s in range(d["mu"].shape[0]):
            mu, var = net.mlp_mc(d["x"], seed, s)
            np.testing.assert_allclose(mu, d["mu"][s], rtol=1e-5, atol=atol)
            np.testing.assert_allclose(var, d["var"][s], rtol=1e-5, atol=1e-8)
    else:
        for s in range(d["probs"].shape[0]):
            got = net.resnet_p(d["x"]) if kind == "resnet_p" else getattr(net, kind)(d["x"], seed, s)
            np.testing.assert_allclose(got, d["probs"][s], rtol=1e-5, atol=atol)
    assert len(net.obs) >= 12
    otol = 1e-5 if atol < 1e-5 else 1e-2          # an observer behind a flipped rounding sees another min / max (one quantisation step of ITS input)
    for k, ob in net.obs.items():
        np.testing.assert_allclose(ob.state[0], d["final/" + k + ".activation_post_process.min_val"], rtol=1e-4, atol=otol)
        np.testing.assert_allclose(ob.state[1], d["final/" + k + ".activation_post_process.max_val"], rtol=1e-4, atol=otol)


def test_resnet_mc_dropout_matches_reference():
    """SURVEY row a7 on the ResNet graph (mcdropout/models_mc.py:116-211): oracle vs the reference, bit-exact block outputs
    for sample 0, 1e-5 on every sample's softmax output."""
    from conftest import load_golden
    g = load_golden("resnet_mc_a7w8.npz")
    net = orc.Int8ResNetMCOracle(g["state"], 7)
    rec = {}
    p0 = net.forward(g["x"], g["meta"]["philox_seed"], 0, record=rec)
    for k, v in g["rec"].items():
        assert np.array_equal(rec[k[:-len(".out")] + ".out"], v), k
    np.testing.assert_allclose(p0, g["probs"][0], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(net.forward(g["x"], g["meta"]["philox_seed"], 2), g["probs"][2], rtol=1e-5, atol=1e-8)


def test_fbgemm_harness_matches_golden(golden):
    """bench.py's `cpu_baseline` of kind "torch-fbgemm" (oracle/fbgemm_baseline.py: the reference's op sequence through
    torch's own quantised CPU operators) reproduces the probabilities recorded from the reference, given the same eps."""
    from oracle.fbgemm_baseline import FbgemmResNetBBB
    wb, seed = golden["meta"]["w_bits"], golden["meta"]["philox_seed"]
    net = orc.Int8ResNetOracle(golden["state"], golden["meta"]["a_bits"], wb)
    fb = FbgemmResNetBBB(golden["state"], golden["meta"]["a_bits"], wb)
    for s in range(golden["probs"].shape[0]):
        eps = {pfx: orc.fill_eps_i8(net.layers[pfx].mu_q.size, seed, i, s).reshape(net.layers[pfx].mu_q.shape)
               for i, (pfx, *_) in enumerate(net.table)}
        rec = {}
        p = fb.forward(golden["x"], eps, record=rec)
        np.testing.assert_allclose(p, golden["probs"][s], rtol=1e-6, atol=1e-9)
        if s == 0:
            for k in ("layers.3.0.out", "layers.4.0.out", "layers.6.1.out"):
                if k in golden["rec"]:
                    assert np.array_equal(rec[k], golden["rec"][k]), k


def test_int8_noise_stream_alias_sampler():
    """The int8 mode draws eps_q directly (oracle: qbo_fill_eps_q; GPU: eps_q_from_u32) through the alias table of
    clamp(rne(N(0,1) / s_n), -128, 127) -- the distribution of the reference's quantised noise (conv_q.py:113-115)."""
    import math
    import re
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a = open(os.path.join(root, "oracle", "qbnn_eps_table.h")).read()
    b = open(os.path.join(root, "quantised_bayesian_nets_amd", "csrc", "qbnn_eps_table.h")).read()
    assert a == b                                                   # oracle and kernels read the same committed data
    words = [int(w, 16) for w in re.findall(r"0x([0-9a-f]{8})u", a)]
    assert len(words) == 256
    # implied probabilities of the table vs the analytic ones (independent of the generator's alias construction)
    s = np.float32(orc.NOISE_SCALE)
    inv = float(np.float32(1.0) / s)
    phi = lambda x: 0.5 * math.erfc(-x / math.sqrt(2.0))
    p = np.array([(1.0 if k == 127 else phi((k + 0.5) / inv)) - (0.0 if k == -128 else phi((k - 0.5) / inv)) for k in range(-128, 128)])
    num = np.zeros(256, np.int64)
    for c, w in enumerate(words):
        num[c] += w >> 8
        num[w & 0xff] += (1 << 24) - (w >> 8)
    assert num.sum() == 1 << 32
    assert np.abs(num / float(1 << 32) - p).max() < 2.0 ** -30
    # the stream: empirical moments of 4M draws match the discrete distribution; keyed by (seed, layer, sample); ragged prefix
    e = orc.fill_eps_q(1 << 22, 3, 5, 7)
    k = np.arange(-128, 128)
    mean, var = (p * k).sum(), (p * k * k).sum() - (p * k).sum() ** 2
    assert abs(e.astype(np.float64).mean() - mean) < 5 * math.sqrt(var / e.size)
    assert abs(e.astype(np.float64).var() - var) < 0.01 * var
    hist = np.bincount(e.astype(np.int64) + 128, minlength=256) / e.size
    assert np.abs(hist - p).max() < 5 * math.sqrt(p.max() / e.size)
    assert hist[0] > 0 and hist[255] > 0                           # the folded tails occur (|eps| > 3)
    assert np.array_equal(orc.fill_eps_q(1003, 3, 5, 7), e[:1003])
    assert not np.array_equal(orc.fill_eps_q(64, 3, 5, 8), e[:64]) and not np.array_equal(orc.fill_eps_q(64, 3, 6, 7), e[:64])
    assert not np.array_equal(orc.fill_eps_q(64, 4, 5, 7), e[:64])
    # injection contract: (float)eps_q * s_n comes back as eps_q from the reference's quantisation formula and from torch's op
    kk = np.arange(-128, 128).astype(np.int8)
    assert np.array_equal(orc.quantize_eps(orc.eps_from_eps_q(kk)), kk)
    import torch
    back = torch.quantize_per_tensor(torch.from_numpy(orc.eps_from_eps_q(kk)), orc.NOISE_SCALE, 0, torch.qint8).int_repr().numpy()
    assert np.array_equal(back, kk)


def test_linear_mc_int8_bit_exact(golden_mlp_mc_q):
    """Row a6+ (`linear_mc`, mcdropout/models_mc.py:10-73), int8: every layer of sample 0 bit for bit, every sample's (mu, var) and the
    regression reduction of experiments/utils.py:348-353 against the reference's recorded outputs."""
    g = golden_mlp_mc_q
    net = orc.Int8MLPMCOracle(g["state"], 7)
    seed = g["meta"]["philox_seed"]
    rec = {}
    net.forward(g["x"], seed, 0, record=rec)
    assert set(g["rec"]) <= set(rec) and len(g["rec"]) == 10
    for k, v in g["rec"].items():
        assert np.array_equal(rec[k].reshape(v.shape), v), k
    for s in range(g["mu"].shape[0]):
        mu, var = net.forward(g["x"], seed, s)
        np.testing.assert_allclose(mu, g["mu"][s], rtol=1e-6, atol=0)
        np.testing.assert_allclose(var, g["var"][s], rtol=1e-6, atol=0)
    mean, pv = net.mc_predict(g["x"], g["mu"].shape[0], seed)
    np.testing.assert_allclose(mean, g["mean"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(pv, g["pred_var"], rtol=1e-5, atol=1e-9)
    # injected masks == the Philox masks
    keep = np.float32(1.0) - np.float32(g["meta"]["p"])
    B = g["x"].shape[0]
    masks = [(orc.fill_uniform(B * 100, seed, di, 2) < keep).astype(np.float32).reshape(B, 100) for di in range(4)]
    mu_i, var_i = net.forward(g["x"], 12345, 0, masks=masks)
    assert np.array_equal(mu_i, net.forward(g["x"], seed, 2)[0])


def test_float_mc_dropout_graphs_match_reference(golden_mc_f32):
    """Rows a6+ / a7 with q=False: the float MC-Dropout MLP, LeNet and ResNet against the reference's recorded outputs.  Tolerance: the
    reference's own spread between its two CPU backends (recorded in the fixture) plus the oracle's fp64-accumulation offset (1e-5 rel)."""
    g = golden_mc_f32
    net = orc.F32MCOracle(g["state"])
    seed = g["meta"]["philox_seed"]
    if g["model"] == "linear_mc":
        for s in range(g["mu"].shape[0]):
            mu, var = net.mlp(g["x"], seed, s)
            np.testing.assert_allclose(mu, g["mu"][s], rtol=1e-5, atol=4.0 * g["refspread"]["max_abs"])
            np.testing.assert_allclose(var, g["var"][s], rtol=1e-5, atol=0)
        return
    fwd = net.lenet if "lenet" in g["model"] else net.resnet
    atol = 2 * g["refspread"]["max_abs"] + 1e-7
    for s in range(g["probs"].shape[0]):
        np.testing.assert_allclose(fwd(g["x"], seed, s), g["probs"][s], rtol=1e-5, atol=atol)


def test_float_bbb_mlp_every_input_width(golden_mlp_f32_width):
    """BASELINE config 0 at in_dim 1, 4, 6, 8, 11: the oracle against the reference's recorded outputs."""
    g = golden_mlp_f32_width
    net = orc.F32MLPOracle(g["state"])
    for s in range(g["mu"].shape[0]):
        mu, var = net.forward(g["x"], g["seed"], s)
        np.testing.assert_allclose(mu, g["mu"][s], rtol=1e-5, atol=g["mu_atol"])
        np.testing.assert_allclose(var, g["var"][s], rtol=1e-5, atol=0)
