"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI of libqbnn_hip.so, against
(a) the golden vectors recorded from the real reference and (b) the CPU oracle on the same seeded inputs.
Integer tensors: bit-exact.  fp32 probabilities / moments: 1e-5 relative (BASELINE.json north_star)."""
import ctypes as C
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def _args(g):
    return types.SimpleNamespace(activation_precision=g["meta"]["a_bits"], weight_precision=g["meta"]["w_bits"])


def _model(g):
    import quantised_bayesian_nets_amd as q
    m = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, _args(g))
    return m.load_reference_state(g["state"])


def _pack(layer, w_logical):
    """logical OHWI int8 -> the layer's device layout, via the C ABI host helper."""
    from quantised_bayesian_nets_amd import _lib
    L = _lib.lib()
    cout = w_logical.shape[0]
    k = int(np.prod(w_logical.shape[1:]))
    krow = layer._krow(w_logical)
    n = L.qbnn_packed_weight_bytes(cout, k, krow, layer.layout)
    dst = np.zeros(n, np.int8)
    src = np.ascontiguousarray(w_logical.reshape(cout, k))
    _lib.check(L.qbnn_pack_weights_host(src.ctypes.data_as(C.c_void_p), cout, k, krow, layer.layout, dst.ctypes.data_as(C.c_void_p)))
    return dst



def pred_var_atol(mu_ref, mu_atol, rtol=1e-5):
    """Absolute tolerance per row for the predictive variance mean_s(var_s) + var_s(mu_s) of a regression model whose per-sample means
    are only known to rtol / mu_atol (north_star: 1e-5 relative on the moments): the variance ACROSS samples subtracts nearly equal
    numbers, so a perturbation d_s of mu_s moves it by mean_s(2 (mu_s - m)(d_s - mean d)) <= 2 std_s(mu) max|d| -- first order, derived,
    instead of a looser relative tolerance on the sum."""
    mu_ref = np.asarray(mu_ref, dtype=np.float64)
    dmax = rtol * np.abs(mu_ref).max(axis=0) + mu_atol
    S = mu_ref.shape[0]
    return (2.0 * mu_ref.std(axis=0) * dmax + dmax ** 2) * (S / max(S - 1.0, 1.0))      # (the unbiased estimator divides by S - 1)

def test_library_is_the_hip_one():
    from quantised_bayesian_nets_amd import _lib
    assert _lib.lib().qbnn_version() == _lib.ABI_VERSION
    assert torch.cuda.is_available()


def test_sampler_philox_matches_oracle_and_golden(golden):
    from oracle import oracle as orc
    g = golden
    m = _model(g)
    net = orc.Int8ResNetOracle(g["state"], g["meta"]["a_bits"], g["meta"]["w_bits"])
    seed = g["meta"]["philox_seed"]
    S = 3
    for name, layer, (pfx, *_r) in zip(m.stochastic_layer_names(), m.stochastic_layers(), net.table):
        w = layer.sample_weights("cuda", samples=S, seed=seed, sample_begin=0).cpu().numpy()
        for s in range(S):
            ref = net.layers[pfx].sample(seed, s)
            assert np.array_equal(w[s], _pack(layer, ref)), (name, s)
        # sample 0 is also what the real reference produced with this eps injected
        assert np.array_equal(w[0], _pack(layer, g["rec"][name + ".w_q"])), name
        # sample_begin offsets the global sample index
        w1 = layer.sample_weights("cuda", samples=1, seed=seed, sample_begin=2).cpu().numpy()
        assert np.array_equal(w1[0], w[2]), name


def test_sampler_injected_eps_equals_philox(golden_w8):
    from oracle import oracle as orc
    m = _model(golden_w8)
    for layer in (m.layers[0], m.layers[4][0].shortcut[0], m.layers[6][1].stem[3], m.layers[9]):
        n = int(np.prod(layer.weight.shape))
        eps = np.stack([orc.fill_eps_i8(n, 11, layer.layer_id, s) for s in (5, 6)])
        a = layer.sample_weights("cuda", samples=2, seed=11, sample_begin=5)
        b = layer.sample_weights("cuda", samples=2, seed=0, sample_begin=0, eps=torch.from_numpy(eps))
        assert torch.equal(a, b)


def test_each_conv_layer_matches_golden(golden):
    """Layer-level: golden input activations + golden sampled weights -> golden output (real layer shapes, B=4)."""
    from quantised_bayesian_nets_amd.layers import MCQTensor
    from quantised_bayesian_nets_amd import _lib
    g = golden
    m = _model(g)
    rec, st = g["rec"], g["state"]
    dev = "cuda"

    def act(name, scale_key):
        s, z = float(np.asarray(st[scale_key + "scale"]).reshape(-1)[0]), int(np.asarray(st[scale_key + "zero_point"]).reshape(-1)[0])
        return MCQTensor(torch.from_numpy(rec[name]).to(dev)[None].contiguous(), s, z)

    x0 = act("quant.out", "quant.")
    # layer 0 (im2col path)
    l0 = m.layers[0]
    B, H, W, _ = rec["quant.out"].shape
    col = torch.empty((B, H * W, 32), dtype=torch.int8, device=dev)
    _lib.check(_lib.lib().qbnn_im2col3x3_c3(_lib.ptr(x0.data), B, H, W, x0.zero_point, _lib.ptr(col), _lib.current_stream()))
    w = torch.from_numpy(_pack(l0, rec["layers.0.w_q"])).to(dev)[None]
    y = l0._conv(x0, w, 1, im2col=col)
    assert np.array_equal(y.data[0].cpu().numpy(), rec["layers.0.out"]), "layers.0"

    prev, prev_key = "layers.0.out", "layers.0."
    for li in (3, 4, 5, 6):
        for bi, blk in enumerate(m.layers[li]):
            p = f"layers.{li}.{bi}."
            xin = act(prev, prev_key)
            w0 = torch.from_numpy(_pack(blk.stem[0], rec[p + "stem.0.w_q"])).to(dev)[None]
            o = blk.stem[0]._conv(xin, w0, 1)
            assert np.array_equal(o.data[0].cpu().numpy(), rec[p + "stem.0.out"]), p + "stem.0"
            w3 = torch.from_numpy(_pack(blk.stem[3], rec[p + "stem.3.w_q"])).to(dev)[None]
            o3 = blk.stem[3]._conv(act(p + "stem.0.out", p + "stem.0."), w3, 1)
            assert np.array_equal(o3.data[0].cpu().numpy(), rec[p + "stem.3.out"]), p + "stem.3"
            if len(blk.shortcut):
                ws = torch.from_numpy(_pack(blk.shortcut[0], rec[p + "shortcut.0.w_q"])).to(dev)[None]
                sc = blk.shortcut[0]._conv(xin, ws, 1)
                assert np.array_equal(sc.data[0].cpu().numpy(), rec[p + "shortcut.0.out"]), p + "shortcut.0"
            else:
                sc = xin
            fused = blk.stem[3]._conv(act(p + "stem.0.out", p + "stem.0."), w3, 1, residual=sc,
                                      add_qparams=(blk.add.add.scale, blk.add.add.zero_point))
            assert np.array_equal(fused.data[0].cpu().numpy(), rec[p[:-1] + ".out"]), p + "add/relu"
            prev, prev_key = p[:-1] + ".out", p + "add.add."


def test_resnet_end_to_end_matches_reference(golden):
    import quantised_bayesian_nets_amd as q
    g = golden
    m = _model(g)
    x = torch.from_numpy(g["x"]).cuda()
    S = g["probs"].shape[0]
    rec = {}
    with q.mc_context(S, g["meta"]["philox_seed"], 0):
        probs = m.forward_mc(x, record=rec)
    assert np.array_equal(rec["quant.out"].cpu().numpy(), g["rec"]["quant.out"])
    assert np.array_equal(rec["layers.0.out"][0].cpu().numpy(), g["rec"]["layers.0.out"])
    for li in (3, 4, 5, 6):
        for bi in (0, 1):
            k = f"layers.{li}.{bi}.out"
            assert np.array_equal(rec[k][0].cpu().numpy(), g["rec"][k]), k
    np.testing.assert_allclose(probs.cpu().numpy(), g["probs"], rtol=RTOL, atol=1e-8)
    mean, var = q.mc_predict(m, x, S, g["meta"]["philox_seed"], return_var=True)
    np.testing.assert_allclose(mean.cpu().numpy(), g["mean_probs"], rtol=RTOL, atol=1e-8)
    ref_var = torch.from_numpy(g["probs"]).double().var(dim=0).numpy()
    np.testing.assert_allclose(var.cpu().numpy(), ref_var, rtol=1e-3, atol=1e-7)           # vs the reference's fp32 probabilities
    np.testing.assert_allclose(var.cpu().numpy(), probs.double().var(dim=0).cpu().numpy(), rtol=1e-5, atol=1e-12)   # fp64 sums: no cancellation
    # reference single-forward call contract
    with q.mc_context(1, g["meta"]["philox_seed"], 1):
        p1 = m(x)
    np.testing.assert_allclose(p1.cpu().numpy(), g["probs"][1], rtol=RTOL, atol=1e-8)


def test_fused_block_chain_equals_layerwise_and_golden(golden):
    """qbnn_block_chain_i8_mc (persistent fused BasicBlocks) against the layer-by-layer C ABI path and the golden."""
    import quantised_bayesian_nets_amd as q
    from quantised_bayesian_nets_amd.layers import MCQTensor
    from quantised_bayesian_nets_amd.models import run_identity_chain, run_down_block
    g = golden
    m = _model(g)
    rec, st = g["rec"], g["state"]
    S, seed = 3, g["meta"]["philox_seed"]

    def act(name, key):
        s, z = float(np.asarray(st[key + "scale"]).reshape(-1)[0]), int(np.asarray(st[key + "zero_point"]).reshape(-1)[0])
        return MCQTensor(torch.from_numpy(rec[name]).cuda()[None].contiguous(), s, z, shared=True)

    with q.mc_context(S, seed, 0):
        # layer 1: two identity blocks in one launch, input = golden layers.0 output (shared by the 3 samples)
        x = act("layers.0.out", "layers.0.")
        y2 = run_identity_chain(list(m.layers[3]), x)
        y1 = run_identity_chain([m.layers[3][0]], x)
        ref1 = m.layers[3][0](x)
        ref2 = m.layers[3][1](ref1)
        assert torch.equal(y1.data, ref1.data) and torch.equal(y2.data, ref2.data)
        assert np.array_equal(y1.data[0].cpu().numpy(), rec["layers.3.0.out"])
        assert np.array_equal(y2.data[0].cpu().numpy(), rec["layers.3.1.out"])
        # down-sampling blocks (shortcut conv + stem + add fused)
        prev, key = "layers.3.1.out", "layers.3.1.add.add."
        for li in (4, 5, 6):
            x = act(prev, key)
            y = run_down_block(m.layers[li][0], x)
            ref = m.layers[li][0](x)
            assert torch.equal(y.data, ref.data), ("down", li)
            assert np.array_equal(y.data[0].cpu().numpy(), rec[f"layers.{li}.0.out"]), ("down", li)
            prev, key = f"layers.{li}.1.out", f"layers.{li}.1.add.add."
        # second block of the other stages
        for li in (4, 5, 6):
            x = act(f"layers.{li}.0.out", f"layers.{li}.0.add.add.")
            y = run_identity_chain([m.layers[li][1]], x)
            ref = m.layers[li][1](x)
            assert torch.equal(y.data, ref.data), li
            assert np.array_equal(y.data[0].cpu().numpy(), rec[f"layers.{li}.1.out"]), li
    # whole model: fused == layer-wise, all samples
    xin = torch.from_numpy(g["x"]).cuda()
    with q.mc_context(S, seed, 0):
        m.fuse_blocks = True
        pf = m.forward_mc(xin)
        m.fuse_blocks = False
        pu = m.forward_mc(xin)
        m.fuse_blocks = True
    assert torch.equal(pf, pu)
    np.testing.assert_allclose(pf.cpu().numpy(), g["probs"], rtol=RTOL, atol=1e-8)


def test_fused_kernels_first_item_race_regression(golden_w8):
    """Few work items per workgroup (S = 1, 2 at B = 256 / 37): the first conv of a workgroup's first item reads the bias
    table and halos written by the kernel prologue.  Repeated fused runs must equal the layer-wise path every time
    (a missing prologue barrier made this fail about one run in three)."""
    import quantised_bayesian_nets_amd as q
    m = _model(golden_w8)
    gen = torch.Generator().manual_seed(11)
    for B in (256, 37):
        x = torch.randn(B, 3, 32, 32, generator=gen).cuda()
        for S in (1, 2):
            with q.mc_context(S, 5, 3):
                m.fuse_blocks = False
                ref = m.forward_mc(x)
                m.fuse_blocks = True
                for _ in range(6):
                    assert torch.equal(m.forward_mc(x), ref), (B, S)


def test_full_size_against_oracle_and_properties(golden_w8):
    """BASELINE config 3 shape (B=256): one sample against the CPU oracle bit-for-bit on the logits path, and
    size-independent properties: chunking / sharding invariance, batch-permutation equivariance, determinism."""
    import quantised_bayesian_nets_amd as q
    from oracle import oracle as orc
    g = golden_w8
    m = _model(g)
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(256, 3, 32, 32, generator=gen)
    xc = x.cuda()
    S, seed = 6, 3
    with q.mc_context(S, seed, 0):
        rec = {}
        probs = m.forward_mc(xc, record=rec)
    net = orc.Int8ResNetOracle(g["state"], 7, 8)
    orec = {}
    p_or = net.forward(x.numpy(), seed, 4, record=orec)
    for k in ["layers.0.out", "layers.3.1.out", "layers.4.0.out", "layers.5.1.out", "layers.6.1.out"]:
        kk = k if k.startswith("layers.0") else k
        ok = orec["layers.0.out"] if k == "layers.0.out" else orec[k]
        assert np.array_equal(rec[kk][4].cpu().numpy(), ok), k
    np.testing.assert_allclose(probs[4].cpu().numpy(), p_or, rtol=RTOL, atol=1e-8)
    # probabilities are normalised
    np.testing.assert_allclose(probs.sum(-1).cpu().numpy(), 1.0, rtol=1e-5)
    # chunking / sharding invariance: samples [0,6) in one launch == [0,2) + [2,6)
    with q.mc_context(2, seed, 0):
        pa = m.forward_mc(xc)
    with q.mc_context(4, seed, 2):
        pb = m.forward_mc(xc)
    assert torch.equal(torch.cat([pa, pb]), probs)
    # determinism
    with q.mc_context(S, seed, 0):
        assert torch.equal(m.forward_mc(xc), probs)
    # batch permutation equivariance (each image is independent given the sample's weights)
    perm = torch.randperm(256, generator=gen)
    with q.mc_context(S, seed, 0):
        pp = m.forward_mc(xc[perm.cuda()])
    assert torch.equal(pp, probs[:, perm.cuda()])
    # ragged batch (not a multiple of the per-workgroup image group)
    with q.mc_context(2, seed, 0):
        pr = m.forward_mc(xc[:37])
    assert torch.equal(pr, probs[:2, :37])
    # mc_predict == mean over samples, chunked or not
    mean = q.mc_predict(m, xc, S, seed)
    mean_c = q.mc_predict(m, xc, S, seed, chunk=4)
    np.testing.assert_allclose(mean.cpu().numpy(), probs.mean(0).cpu().numpy(), rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(mean_c.cpu().numpy(), mean.cpu().numpy(), rtol=RTOL, atol=1e-8)


def test_lenet_mc_dropout_matches_reference(golden_lenet_mc):
    """BASELINE config 2 (MNIST-shaped LeNet, MC-Dropout, A7/W8): in-kernel Philox masks, quantised dropout, generic int8
    conv / linear, max-pool, head -- every layer of sample 0 and all per-sample probabilities against the reference."""
    import quantised_bayesian_nets_amd as q
    from oracle import oracle as orc
    g = golden_lenet_mc
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, p=float(g["meta"]["p"]) if "p" in g["meta"] else 0.2)
    m = q.ModelFactory.get_model("conv_lenet_mc", [1, 1, 28, 28], 10, True, args).load_reference_state(g["state"])
    x = torch.from_numpy(g["x"]).cuda()
    S, seed = g["probs"].shape[0], g["meta"]["philox_seed"]
    rec = {}
    with q.mc_context(S, seed, 0):
        probs = m.forward_mc(x, record=rec)
    for k, v in g["rec"].items():
        got = rec[k][0].cpu().numpy()
        assert np.array_equal(got.reshape(v.shape), v), k
    np.testing.assert_allclose(probs.cpu().numpy(), g["probs"], rtol=RTOL, atol=1e-8)
    mean = q.mc_predict(m, x, S, seed)
    np.testing.assert_allclose(mean.cpu().numpy(), g["mean_probs"], rtol=RTOL, atol=1e-8)
    # without `record` the model pools before it drops (the two commute exactly): same bits as the layer-by-layer order above
    with q.mc_context(S, seed, 0):
        assert torch.equal(m.forward_mc(x), probs)
    # injected masks == Philox masks; sample_begin offsets the stream; bigger batch against the oracle
    net = orc.Int8LeNetMCOracle(g["state"], 7)
    keep = np.float32(1.0) - np.float32(0.2)
    B = x.shape[0]
    masks = {di: torch.from_numpy(np.stack([(orc.fill_uniform(B * c, seed, di, s) < keep).astype(np.float32).reshape(B, c) for s in (1, 2)]))
             for di, c in enumerate((20, 50, 500))}
    with q.mc_context(2, 999, 0):
        pm = m.forward_mc(x, masks=masks)
    assert torch.equal(pm, probs[1:3])
    gen = torch.Generator().manual_seed(5)
    xb = torch.rand(128, 1, 28, 28, generator=gen)
    with q.mc_context(2, seed, 7):
        pb = m.forward_mc(xb.cuda())
    np.testing.assert_allclose(pb[1].cpu().numpy(), net.forward(xb.numpy(), seed, 8), rtol=RTOL, atol=1e-8)


def test_float_bbb_mlp_matches_reference(golden_mlp_f32):
    """BASELINE config 0: fp32 BBB MLP, in-kernel Philox eps, per-sample (mu, var) and the regression MC reduction
    (experiments/utils.py:348-353) against the reference; tolerance 1e-5 relative (BASELINE north_star)."""
    import quantised_bayesian_nets_amd as q
    g = golden_mlp_f32
    args = types.SimpleNamespace(sigma_prior=-2.0)
    m = q.ModelFactory.get_model("linear_bbb", [g["in_dim"]], 1, False, args).load_reference_state(g["state"])
    x = torch.from_numpy(g["x"]).cuda()
    S = g["mu"].shape[0]
    with q.mc_context(S, g["seed"], 0):
        mu, var = m.forward_mc(x)
    np.testing.assert_allclose(mu.cpu().numpy(), g["mu"], rtol=1e-5, atol=g["mu_atol"])      # measured floor: conftest._mlp_f32
    np.testing.assert_allclose(var.cpu().numpy(), g["var"], rtol=1e-5, atol=0)
    mean, pv = q.mc_predict_regression(m, x, S, g["seed"])
    np.testing.assert_allclose(mean.cpu().numpy(), g["mean"], rtol=1e-5, atol=g["mu_atol"])
    np.testing.assert_allclose(pv.cpu().numpy(), g["pred_var"], rtol=1e-4, atol=1e-7)
    own = mu.double().var(dim=0) + var.double().mean(dim=0)                # experiments/utils.py:352-353 on the device's own samples
    np.testing.assert_allclose(pv.cpu().numpy(), own.cpu().numpy(), rtol=1e-5, atol=1e-12)
    with q.mc_context(1, g["seed"], 4):
        mu4, var4 = m(x)
    np.testing.assert_allclose(mu4.cpu().numpy(), g["mu"][4], rtol=1e-5, atol=g["mu_atol"])


def test_ensemble_matches_reference(golden_ensemble):
    """BASELINE config 3: SGHMC-style ensemble of deterministic int8 ResNets; members are the MC samples."""
    import quantised_bayesian_nets_amd as q
    g = golden_ensemble
    n = len(g["members"])
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, model="conv_resnet_sgld", samples=n)
    net = q.ModelFactory.get_model("conv_resnet_sgld", [1, 3, 32, 32], 10, True, args, training_mode=False).load_reference_state(g["members"])
    x = torch.from_numpy(g["x"]).cuda()
    rec = {}
    with q.mc_context(n, 0, 0):
        probs = net.forward_mc(x, record=rec)
    for k in ("layers.0.out", "layers.3.1.out", "layers.4.0.out", "layers.6.1.out"):
        got = rec[k][0].cpu().numpy() if rec[k].dim() == 5 else rec[k].cpu().numpy()
        assert np.array_equal(got, g["rec"][k]), k
    np.testing.assert_allclose(probs.cpu().numpy(), g["probs"], rtol=RTOL, atol=1e-8)
    mean = q.mc_predict(net, x, n, 0)
    np.testing.assert_allclose(mean.cpu().numpy(), g["mean_probs"], rtol=RTOL, atol=1e-8)
    # the members' launch chains are now captured as HIP graphs: later passes replay them on side streams, on new inputs too
    for _ in range(3):
        with q.mc_context(n, 0, 0):
            assert torch.equal(net.forward_mc(x), probs)
    x2 = torch.flip(x, dims=[0])
    with q.mc_context(n, 0, 0):
        assert torch.equal(net.forward_mc(x2), torch.flip(probs, dims=[1]))
    # the reference's round-robin call contract (models_sgld.py:277-288)
    outs = [net(x).cpu().numpy() for _ in range(n + 1)]
    np.testing.assert_allclose(np.stack(outs[:n]), g["probs"], rtol=RTOL, atol=1e-8)
    assert np.array_equal(outs[0], outs[n])
    with pytest.raises(NotImplementedError):
        q.ModelFactory.get_model("conv_resnet_sgld", [1, 3, 32, 32], 10, True, args, training_mode=True)


def test_classification_metrics_match_reference_formulas():
    """src/metrics.py formulas evaluated with torch on the CPU (the reference's own expressions) vs the device kernel."""
    import torch.nn.functional as F
    import quantised_bayesian_nets_amd as q
    gen = torch.Generator().manual_seed(0)
    B, Cc = 1000, 10
    probs = torch.softmax(torch.randn(B, Cc, generator=gen) * 2, -1)
    target = torch.randint(0, Cc, (B,), generator=gen)
    m = q.ClassificationMetric(Cc)
    m.update(probs[:600].cuda(), target[:600].cuda())
    m.update(probs[600:].cuda(), target[600:].cuda())
    oh = F.one_hot(target, Cc).float()
    assert abs(m.error - float((probs.argmax(1) != target).sum()) / B) < 1e-12
    assert abs(m.nll - float(torch.sum(-oh * torch.log(probs + 1e-8))) / B) < 1e-5
    assert abs(m.brier - float(torch.sum((probs - oh) ** 2)) / B) < 1e-5
    assert abs(m.entropy - float(torch.sum(-probs * torch.log(probs + 1e-8))) / B) < 1e-5
    conf, pred = probs.max(1)
    acc = (pred == target).float()
    bins = torch.bucketize(conf, torch.linspace(0, 1, 11), right=True) - 1
    ece = sum(abs(acc[bins == b].mean() - conf[bins == b].mean()) * (bins == b).float().mean() for b in range(10) if (bins == b).any())
    assert abs(m.ece - float(ece)) < 1e-5


def test_regression_metrics_match_reference_formulas(golden_mlp_f32):
    """src/metrics.py:119-230 (Gaussian NLL with its 1e-8 guards, MSE / RMSE, MAE) evaluated with torch on the CPU -- the
    reference's literal expressions -- vs the device kernel, on the MC-reduced (mean, variance) of the fp32 BBB MLP."""
    import math
    import torch.nn.functional as F
    import quantised_bayesian_nets_amd as q
    g = golden_mlp_f32
    m = q.ModelFactory.get_model("linear_bbb", [g["in_dim"]], 1, False, types.SimpleNamespace(sigma_prior=-2.0)).load_reference_state(g["state"])
    x = torch.from_numpy(g["x"]).cuda()
    mean, var = q.mc_predict_regression(m, x, g["mu"].shape[0], g["seed"])
    target = torch.randn(x.shape[0], generator=torch.Generator().manual_seed(4))
    met = q.RegressionMetric(1)
    half = x.shape[0] // 2
    met.update((mean[:half], var[:half]), target[:half])             # accumulates over batches like the reference's metric objects
    met.update((mean[half:], var[half:]), target[half:])
    mu_c, var_c, B = mean.cpu().squeeze(), var.cpu().squeeze(), x.shape[0]
    nll = torch.sum(0.5 * torch.log(2 * math.pi * var_c + 1e-8) + (target - mu_c) ** 2 / (2 * var_c + 1e-8)) / B
    mse = F.mse_loss(mu_c, target, reduction="sum") / B
    mae = F.l1_loss(mu_c, target, reduction="sum") / B
    assert abs(met.nll - float(nll)) < 1e-5 * max(1.0, abs(float(nll)))
    assert abs(met.mse - float(mse)) < 1e-5 * float(mse) and abs(met.rmse - float(torch.sqrt(mse))) < 1e-5 * float(torch.sqrt(mse))
    assert abs(met.mae - float(mae)) < 1e-5 * float(mae)
    assert met.get_key_metric() == met.rmse and sorted(met.compute()) == ["mae", "mse", "nll", "rmse"]
    only_mean = q.RegressionMetric(1)
    only_mean.update((mean, None), target)                            # metrics.py:154: a mean-only model is scored with unit variance
    nll1 = torch.sum(0.5 * torch.log(torch.tensor(2 * math.pi) + 1e-8) + (target - mu_c) ** 2 / (2 + 1e-8)) / B
    assert abs(only_mean.nll - float(nll1)) < 1e-5 * abs(float(nll1))


def test_small_bbb_int8_graphs_match_reference(golden_lenet_bbb, golden_mlp_bbb_q):
    """int8 BBB LeNet and MLP (linear_q.Linear / LinearReLU forward for real): every layer of sample 0
    bit-exact, all samples' outputs to 1e-5 relative."""
    import quantised_bayesian_nets_amd as q
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8)
    g = golden_lenet_bbb
    m = q.ModelFactory.get_model("conv_lenet_bbb", [1, 1, 28, 28], 10, True, args).load_reference_state(g["state"])
    S, seed = g["probs"].shape[0], g["meta"]["philox_seed"]
    rec = {}
    with q.mc_context(S, seed, 0):
        probs = m.forward_mc(torch.from_numpy(g["x"]).cuda(), record=rec)
    for k, v in g["rec"].items():
        assert np.array_equal(rec[k][0].cpu().numpy().reshape(v.shape), v), k
    np.testing.assert_allclose(probs.cpu().numpy(), g["probs"], rtol=RTOL, atol=1e-8)
    mm = golden_mlp_bbb_q
    net = q.ModelFactory.get_model("linear_bbb", [13], 1, True, args).load_reference_state(mm["state"])
    rec = {}
    with q.mc_context(mm["mu"].shape[0], mm["seed"], 0):
        mu, var = net.forward_mc(torch.from_numpy(mm["x"]).cuda(), record=rec)
    for k, v in mm["rec"].items():
        assert np.array_equal(rec[k][0].cpu().numpy().reshape(v.shape), v), k
    np.testing.assert_allclose(mu.cpu().numpy(), mm["mu"], rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(var.cpu().numpy(), mm["var"], rtol=RTOL, atol=1e-9)
    mean, pv = q.mc_predict_regression(net, torch.from_numpy(mm["x"]).cuda(), mm["mu"].shape[0], mm["seed"])
    np.testing.assert_allclose(mean.cpu().numpy(), mm["mu"].mean(0), rtol=1e-5, atol=1e-6)


def test_lenet_bbb_fast_path_equals_generic_kernels_and_reference(golden_lenet_bbb):
    """int8 BBB LeNet with sampled weights on the small networks' own kernels (fused conv + pool + Flatten, pitched NHWC -> NCHW flatten,
    int8 GEMMs; the sampler writes the fragment layouts) against the any-geometry kernels: bit-identical probabilities on the fixture's
    batch, on a ragged batch and at a sample offset; and the fixture's recorded probabilities from the real reference through the fast path."""
    import quantised_bayesian_nets_amd as q
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8)
    g = golden_lenet_bbb
    m = q.ModelFactory.get_model("conv_lenet_bbb", [1, 1, 28, 28], 10, True, args).load_reference_state(g["state"])
    S, seed = g["probs"].shape[0], g["meta"]["philox_seed"]
    xg = torch.from_numpy(g["x"]).cuda()
    assert m._can_run_fast(xg, None)
    with q.mc_context(S, seed, 0):
        fast = m.forward_mc(xg)
    np.testing.assert_allclose(fast.cpu().numpy(), g["probs"], rtol=RTOL, atol=1e-8)
    x = torch.rand(37, 1, 28, 28, generator=torch.Generator().manual_seed(3)).cuda()
    for sb in (0, 250):
        with q.mc_context(5, 11, sb):
            a = m.forward_mc(x)
            m.fast_path = False
            try:
                b = m.forward_mc(x)
            finally:
                m.fast_path = True
            c = m.forward_mc(x)                         # and back: the packed layouts switch with the path
        assert torch.equal(a, b) and torch.equal(a, c)


def test_errors_are_loud():
    from quantised_bayesian_nets_amd import _lib
    d = _lib.ConvDesc()
    d.B, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.pad, d.a_hi = 1, 7, 7, 5, 9, 3, 1, 1, 127
    d.s_x = d.s_w = d.s_y = 1.0
    t = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    rc = _lib.lib().qbnn_conv2d_i8_mc(_lib.ptr(t), 0, _lib.ptr(t), 0, None, None, 0, _lib.ptr(t), 0, 1, C.byref(d), _lib.current_stream())
    assert rc < 0 and b"unsupported geometry" in _lib.lib().qbnn_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc)


@pytest.mark.parametrize("name,model", [("lenet_bbb_f32.npz", "conv_lenet_bbb"), ("resnet_bbb_f32.npz", "conv_resnet_bbb")])
def test_float_bbb_conv_graphs_match_reference(name, model):
    """float BBB conv graphs on the GPU (MFMA fp32 implicit-GEMM conv, per-sample weights, in-kernel Philox
    eps) against the reference's per-sample softmax outputs and their MC mean.  Tolerance: 1e-5 relative (north_star) plus an
    absolute term that is MEASURED, not chosen: the reference evaluated on its two CPU conv backends (oneDNN / plain ATen, another fp32
    summation order of the same arithmetic; tests/golden/make_golden_conv_f32.py records both) differs from itself by 3.0e-7 (LeNet)
    / 1.8e-7 (ResNet) absolute and 1.13e-5 / 1.3e-6 relative -- no fp32 implementation can be closer to "the reference" than the
    reference is to itself, so atol = 2 x that spread (6e-7 / 3.6e-7; round 2 used a flat 2e-6).  Measured on the MI355X: max |dp|
    8.3e-7 / 1.5e-7, max relative error on a probability (= on its logit's exp) 1.04e-5 / 9.4e-7, against the closer backend."""
    import quantised_bayesian_nets_amd as q
    from conftest import load_golden, GOLDEN
    g = load_golden(name)
    raw = np.load(os.path.join(GOLDEN, name))
    spread_abs, spread_rel = float(raw["refspread.max_abs"]), float(raw["refspread.max_rel"])
    atol = 2.0 * spread_abs
    assert 1e-8 < spread_abs < 5e-7 and atol <= 6.5e-7
    args = types.SimpleNamespace(sigma_prior=-2.0)
    shape = [1, 28, 28] if "lenet" in model else [1, 3, 32, 32]
    m = q.ModelFactory.get_model(model, shape, 10, False, args).load_reference_state(g["state"])
    x = torch.from_numpy(g["x"]).cuda()
    S = g["probs"].shape[0]
    with q.mc_context(S, g["meta"]["philox_seed"], 0):
        p = m.forward_mc(x)
    pn = p.cpu().numpy()
    np.testing.assert_allclose(pn, g["probs"], rtol=1e-5, atol=atol)
    # relative error on the probabilities themselves (= on the logits up to the softmax's per-row shift: d log p), against the closer
    # of the two reference backends per element: within 1e-5 + what the backends differ by between themselves
    both = np.stack([g["probs"], raw["probs_aten"]]).astype(np.float64)
    rel = (np.abs(pn.astype(np.float64)[None] - both) / both).min(0)
    print("\n%s: max |dp| %.2e, max relative error on p (d log p) %.2e; reference oneDNN vs ATen: %.2e abs, %.2e rel" % (
        model, np.abs(pn - g["probs"]).max(), rel.max(), spread_abs, spread_rel))
    assert rel.max() <= 1e-5 + spread_rel
    mean = q.mc_predict(m, x, S, g["meta"]["philox_seed"])
    np.testing.assert_allclose(mean.cpu().numpy(), g["mean_probs"], rtol=1e-5, atol=atol)
    # the oracle on a sample the fixture does not hold
    from oracle import oracle as orc
    net = orc.F32ConvOracle(g["state"])
    fwd = net.lenet if "lenet" in model else net.resnet
    with q.mc_context(1, 11, 5):
        p5 = m(x)
    np.testing.assert_allclose(p5.cpu().numpy(), fwd(g["x"], 11, 5), rtol=1e-5, atol=atol)


def test_float_resnet_full_batch_against_oracle():
    """Row a1 at the headline's batch: the float BBB ResNet-18 at B = 256 (every workgroup tiling of the fp32 MFMA conv in play, not
    the B = 2 of the reference fixture) on one MC sample against the CPU oracle of reference bbb/conv.py:33-39 (F32ConvOracle, itself
    pinned to the reference by the fixture generator), 1e-5 relative + the measured 3.6e-7."""
    import quantised_bayesian_nets_amd as q
    from conftest import load_golden
    from oracle import oracle as orc
    g = load_golden("resnet_bbb_f32.npz")
    m = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, False, types.SimpleNamespace(sigma_prior=-2.0)).load_reference_state(g["state"])
    x = torch.randn(256, 3, 32, 32, generator=torch.Generator().manual_seed(12))
    with q.mc_context(1, 21, 3):
        p = m(x.cuda())
    want = orc.F32ConvOracle(g["state"]).resnet(x.numpy(), 21, 3)
    np.testing.assert_allclose(p.cpu().numpy(), want, rtol=1e-5, atol=3.6e-7)


@pytest.mark.parametrize("name,model", [("mlp_bbb_qat.npz", "linear_bbb"), ("lenet_bbb_qat.npz", "conv_lenet_bbb"),
                                        ("resnet_bbb_qat.npz", "conv_resnet_bbb")])
def test_qat_eval_with_live_observers_matches_reference(name, model):
    """the prepared (QAT) model in eval mode on the GPU -- all S samples in one batched pass with the
    observer recurrence resolved on the device -- against S sequential reference forwards (same injected eps): per-sample
    outputs, and every observer's final (min, max).  fp32 tolerance 1e-5 relative + a measured absolute floor (below)."""
    import os
    import quantised_bayesian_nets_amd as q
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
    st = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
    args = types.SimpleNamespace(sigma_prior=-2.0, activation_precision=7, weight_precision=8, qat_eval=True)
    shape = {"linear_bbb": [13], "conv_lenet_bbb": [1, 28, 28], "conv_resnet_bbb": [1, 3, 32, 32]}[model]
    m = q.ModelFactory.get_model(model, shape, 1 if model == "linear_bbb" else 10, True, args).load_reference_state(st)
    x = torch.from_numpy(d["x"]).cuda()
    seed = int(d["meta.philox_seed"])
    # Absolute floor beside the 1e-5 relative bound: four times the reference's own distance from itself on another CPU code path
    # (`refspread.max_abs`: MLP 9.5e-7, LeNet 1.8e-7; measured here: 0 and 6e-8).  The ResNet's own spread is 2.0e-3 -- on another code
    # path some of the reference's fake-quantisers round the other way, which is NOT what the build is allowed: it reproduces the recorded
    # run's roundings, and its floor is 1e-6 (measured 4.9e-7 absolute, 1.1e-5 relative on probabilities of 0.009 - 0.25).
    atol = min(4.0 * float(d["refspread.max_abs"]), 1e-6 if model == "conv_resnet_bbb" else 1.0)
    if model == "linear_bbb":
        S = d["mu"].shape[0]
        with q.mc_context(S, seed, 0):
            mu, var = m.forward_mc(x)
        np.testing.assert_allclose(mu.cpu().numpy(), d["mu"], rtol=1e-5, atol=atol)
        np.testing.assert_allclose(var.cpu().numpy(), d["var"], rtol=1e-5, atol=0)
    else:
        S = d["probs"].shape[0]
        with q.mc_context(S, seed, 0):
            p = m.forward_mc(x)
        np.testing.assert_allclose(p.cpu().numpy(), d["probs"], rtol=1e-5, atol=atol)
    checked = 0
    for k in d.files:
        if k.startswith("final/") and k.endswith("min_val"):
            prefix = k[len("final/"):-len(".activation_post_process.min_val")]
            mod = m
            for part in prefix.replace(".add.add.activation_post_process", ".add").replace(".mul_noise.activation_post_process", ".mul_noise") \
                             .replace(".add_weight.activation_post_process", ".add_weight").split("."):
                mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
            mn, mx = mod.min_max()
            np.testing.assert_allclose(mn, float(d[k]), rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(mx, float(d[k.replace("min_val", "max_val")]), rtol=1e-4, atol=1e-5)
            checked += 1
    assert checked >= 20


@pytest.mark.parametrize("name,model", [("mlp_mc_qat.npz", "linear_mc"), ("lenet_mc_qat.npz", "conv_lenet_mc"), ("resnet_mc_qat.npz", "conv_resnet_mc"),
                                        ("resnet_sgld_qat.npz", "conv_resnet_sgld")])
def test_qat_eval_of_the_non_bbb_graphs_matches_reference(name, model):
    """SURVEY 8(f).3 widened to quant_utils.prepare_model's `prepare_qat` branch (:139-140): the prepared MC-Dropout graphs (FakeQuantize on the
    dropout's mul_mask, mcdropout/dropout.py:9-40) and the SGHMC member template in eval mode on the GPU, all S samples in one batched pass,
    against S sequential reference forwards with the same injected masks: per-sample outputs and every live observer's final (min, max).
    Tolerance: 1e-5 relative + twice the reference's own distance from itself on another CPU code path (`refspread.max_abs`: 0 / 4.5e-8 for
    the MLP / LeNet, i.e. the 2e-6 floor; 3.9e-4 / 5.2e-4 for the ResNets, where a few activations sit within fp32 summation noise of a
    quantisation step -- the CPU oracle, which accumulates in fp64, is 2.4e-4 from the recorded run for the same reason)."""
    import os
    import quantised_bayesian_nets_amd as q
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
    st = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
    args = types.SimpleNamespace(p=float(d["meta.p"]), activation_precision=7, weight_precision=8, qat_eval=True, model=model)
    shape = {"linear_mc": [13], "conv_lenet_mc": [1, 28, 28]}.get(model, [1, 3, 32, 32])
    m = q.ModelFactory.get_model(model, shape, 1 if model == "linear_mc" else 10, True, args).load_reference_state(st)
    assert type(m).__module__.endswith("models_qat_mc")
    x = torch.from_numpy(d["x"]).cuda()
    seed = int(d["meta.philox_seed"])
    atol = max(2e-6, 2.0 * float(d["refspread.max_abs"]))
    if model == "linear_mc":
        S = d["mu"].shape[0]
        with q.mc_context(S, seed, 0):
            mu, var = m.forward_mc(x)
        np.testing.assert_allclose(mu.cpu().numpy(), d["mu"], rtol=1e-5, atol=atol)
        np.testing.assert_allclose(var.cpu().numpy(), d["var"], rtol=1e-5, atol=0)
    else:
        S = d["probs"].shape[0]
        with q.mc_context(S, seed, 0):
            p = m.forward_mc(x)
        np.testing.assert_allclose(p.cpu().numpy(), d["probs"], rtol=1e-5, atol=atol)
        if atol > 1e-5:
            # Round 6: the loose floor above cannot catch an error of 1e-4, so the build is also held to the reference's OWN statistics on this fixture
            # (tests/golden/qat_refspread_counts.json, from make_golden_qat_counts.py: the reference's second run on another CPU code path against its
            # recorded first): no more probabilities outside 1e-5 + 1e-6 than the reference shows against itself, a mean deviation no larger than its
            # own, and the same arg-max class in every row.
            import json
            ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "qat_refspread_counts.json")))[name]
            dev = np.abs(p.cpu().numpy().astype(np.float64) - d["probs"])
            n_out = int((dev > 1e-6 + 1e-5 * np.abs(d["probs"])).sum())
            print("%s: %d of %d probabilities outside 1e-5 + 1e-6 (reference vs itself: %d), mean |d| %.3g (reference: %.3g), max %.3g (%.3g)"
                  % (name, n_out, dev.size, ref["n_outside_1e-5_1e-6"], dev.mean(), ref["mean_abs"], dev.max(), ref["max_abs"]))
            assert n_out <= ref["n_outside_1e-5_1e-6"] and dev.mean() <= ref["mean_abs"]
            assert int((p.cpu().numpy().argmax(-1) == d["probs"].argmax(-1)).sum()) == ref["rows"]
    otol = 1e-5 if atol < 1e-5 else 1e-2
    st2 = m.prepared_state()
    checked = 0
    for k in d.files:
        if k.startswith("final/") and k.endswith("min_val") and np.isfinite(float(d[k])):
            key = k[len("final/"):]
            np.testing.assert_allclose(float(st2[key]), float(d[k]), rtol=1e-4, atol=otol)
            np.testing.assert_allclose(float(st2[key.replace("min_val", "max_val")]), float(d[k.replace("min_val", "max_val")]), rtol=1e-4, atol=otol)
            checked += 1
    assert checked >= 12


def test_native_prepare_calibrate_convert_pipeline_of_the_mc_dropout_resnet():
    """SURVEY 8f row 4 widened to quant_utils.prepare_model's `prepare_qat` branch (:139-140), end to end without the reference: the float
    conv_resnet_mc state -> `prepare_model_state` -> calibration by live-observer evaluation forwards on the GPU (models_qat_mc; forward i draws
    the masks of sample index i) -> `convert_model_state` -> the int8 model on the HIP path; against what the REFERENCE produced from the same
    float model with prepare_model -> 3 eval forwards (same injected masks, plain ATen convs) -> convert (tests/golden/make_golden_prepare_mc.py).
    Observers: weight side 1e-5 relative; activation side within 1e-3 of the observer's range (measured: 1.4e-4 -- this fixture's BatchNorm
    statistics let the activations grow to ~270 by the last stage, and a few of them round the other way under the upstream fake-quantisers,
    as the reference itself does on another conv backend: resnet_mc_qat.npz `refspread`).  Converted state: zero points within one step, scales
    1e-3, 19 of the 20 qint8 conv tensors bit-identical and the recorded ones equal up to one element on a rounding tie; the converted model's
    int8 probabilities on the HIP path equal the reference's int8 model's."""
    import hashlib
    import os
    import quantised_bayesian_nets_amd as q
    from conftest import GOLDEN
    from quantised_bayesian_nets_amd.convert import prepare_model_state, calibrate, convert_model_state, convert_model
    d = np.load(os.path.join(GOLDEN, "resnet_mc_f32.npz"))
    fstate = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
    ref = np.load(os.path.join(GOLDEN, "resnet_mc_prepare_calibrate.npz"))
    S, seed, p = int(ref["meta.samples"]), int(ref["meta.philox_seed"]), float(ref["meta.p"])
    fstate["layers.10.weight"] = (np.asarray(fstate["layers.10.weight"]) * np.float32(ref["meta.logit_gain"])).astype(np.float32)
    aq = types.SimpleNamespace(p=p, activation_precision=7, weight_precision=8, qat_eval=True)
    m = q.ModelFactory.get_model("conv_resnet_mc", [1, 3, 32, 32], 10, True, aq).load_reference_state(prepare_model_state(fstate))
    x = torch.from_numpy(d["x"]).cuda()
    calibrate(m, [x] * S, seed)
    st = m.prepared_state()
    n_w = n_a = 0
    worst = 0.0
    for k in ref.files:
        if not k.startswith("calibrated/") or not k.endswith("min_val") or not np.isfinite(float(ref[k])):
            continue
        kk = k[len("calibrated/"):]
        lo, hi = float(ref[k]), float(ref[k.replace("min_val", "max_val")])
        glo, ghi = float(st[kk]), float(st[kk.replace("min_val", "max_val")])
        if "weight_fake_quant" in kk:
            assert abs(glo - lo) <= 1e-5 * max(1e-3, abs(lo)) + 1e-9 and abs(ghi - hi) <= 1e-5 * max(1e-3, abs(hi)) + 1e-9, (kk, glo, lo, ghi, hi)
            n_w += 1
        else:
            rng = max(hi, 0.0) - min(lo, 0.0)
            dev = max(abs(glo - lo), abs(ghi - hi)) / rng
            worst = max(worst, dev)
            assert dev <= 1e-3, (kk, glo, lo, ghi, hi)
            n_a += 1
    print("worst activation-observer deviation (fraction of its range):", worst)
    assert n_w == 21 and n_a == 21 + 20 + 8 + 1            # 21 layers; their outputs + 20 dropouts' mul_mask + 8 Adds + the stub
    conv = convert_model_state(st, types.SimpleNamespace(activation_precision=7, weight_precision=8))
    n_int8 = n_same = 0
    for k in ref.files:
        if not k.startswith("converted/"):
            continue
        key = k[len("converted/"):]
        if key.endswith(".sha1"):
            base = key[:-len(".sha1")]
            n_int8 += 1
            n_same += int(hashlib.sha1(np.ascontiguousarray(conv[base]).tobytes()).hexdigest() == str(ref[k]))
        elif key.endswith("scale"):
            np.testing.assert_allclose(float(np.asarray(conv[key]).reshape(-1)[0]), float(np.asarray(ref[k]).reshape(-1)[0]), rtol=1e-3, err_msg=key)
        elif key.endswith("zero_point"):
            assert abs(int(np.asarray(conv[key]).reshape(-1)[0]) - int(np.asarray(ref[k]).reshape(-1)[0])) <= 1, key
    print("int8 tensors equal:", n_same, "of", n_int8)
    assert n_int8 == 20 and n_same >= 19, (n_same, n_int8)
    for key in ("layers.0.weight", "layers.5.0.shortcut.0.weight", "layers.7.1.stem.4.weight", "layers.10.weight"):
        dd = np.asarray(conv[key]).astype(np.int32) - ref["converted/" + key].astype(np.int32)
        assert int((dd != 0).sum()) <= 1 and int(np.abs(dd).max()) <= 1, key
    # ... and the converted model runs on the HIP path: the int8 MC-Dropout ResNet with the masks of sample indices S, S + 1
    a8 = types.SimpleNamespace(p=p, activation_precision=7, weight_precision=8)
    mi = convert_model(m, "conv_resnet_mc", [1, 3, 32, 32], 10, a8)
    with q.mc_context(ref["int8_probs"].shape[0], seed, S):
        pi = mi.forward_mc(x)
    np.testing.assert_allclose(pi.cpu().numpy(), ref["int8_probs"], rtol=1e-5, atol=1e-7)


def test_many_samples_fused_equals_layerwise(golden_w8):
    """BASELINE config 3 size in the sample dimension too (B = 256, S = 48: every persistent workgroup walks dozens of work
    items across several MC samples, reloading its LDS-resident weights on the way): the fully fused path (stem + chains +
    down blocks) must equal the one-launch-per-conv path bit for bit, twice."""
    import quantised_bayesian_nets_amd as q
    m = _model(golden_w8)
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(256, 3, 32, 32, generator=gen).cuda()
    with q.mc_context(48, 9, 100):
        m.fuse_blocks = False
        ref = m.forward_mc(x)
        m.fuse_blocks = True
        for _ in range(2):
            assert torch.equal(m.forward_mc(x), ref)


def test_resnet_mc_dropout_matches_reference():
    """SURVEY row a7 on the ResNet graph (`conv_resnet_mc`): deterministic int8 convs with an in-kernel Philox channel
    dropout after every conv; block outputs of sample 0 bit-exact, per-sample probabilities 1e-5."""
    import quantised_bayesian_nets_amd as q
    from conftest import load_golden
    g = load_golden("resnet_mc_a7w8.npz")
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, p=g["meta"]["p"])
    m = q.ModelFactory.get_model("conv_resnet_mc", [1, 3, 32, 32], 10, True, args).load_reference_state(g["state"])
    x = torch.from_numpy(g["x"]).cuda()
    S = g["probs"].shape[0]
    rec = {}
    with q.mc_context(S, g["meta"]["philox_seed"], 0):
        p = m.forward_mc(x, record=rec)
    for k, v in g["rec"].items():
        assert np.array_equal(rec[k][0].cpu().numpy(), v), k
    np.testing.assert_allclose(p.cpu().numpy(), g["probs"], rtol=1e-5, atol=1e-8)
    mean = q.mc_predict(m, x, S, g["meta"]["philox_seed"])
    np.testing.assert_allclose(mean.cpu().numpy(), g["mean_probs"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("acc64", [False, True])
def test_fp32_conv_kernels_every_path_against_torch(acc64):
    """qbnn_conv2d_f32_fused_mc through the C ABI against torch's fp32 conv2d + the same pointwise tail, on geometries that
    reach every kernel variant: float4 path with 64x64 and 128x32 tiles, K not a multiple of the 16-wide chunk (Cin = 20, 24),
    ragged pixel / channel tiles, stride 2, 1x1, shared input (sample stride 0), the gather kernel (Cin = 3, reference
    weight order) and the fused tail (/ div, + bias, * alpha, + beta, + res, ReLU).  fp32 tolerance: 1e-5 of the output scale
    (summation order differs); the fp64-accumulating variant must agree with a float64 torch conv to one fp32 rounding."""
    from quantised_bayesian_nets_amd.models_f32 import conv2d_f32
    gen = torch.Generator().manual_seed(5)
    cases = [  # S, B, H, Cin, Cout, k, stride, pad, ohwi, shared_x, tail
        (3, 5, 12, 24, 24, 3, 1, 1, True, False, True),     # 128x32 tiles, K = 216 (13.5 chunks), ragged pixel tile
        (2, 3, 9, 20, 40, 3, 2, 1, True, False, False),     # Cin % 4 == 0 only, 64x64 tiles, ragged Cout, stride 2
        (2, 4, 8, 48, 96, 1, 2, 0, True, True, True),       # 1x1 stride 2, Cout = 96 -> narrow tiles, input shared by the samples
        (2, 2, 6, 192, 192, 3, 1, 1, True, False, True),    # 64x64 tiles, K = 1728
        (2, 3, 10, 3, 24, 3, 1, 1, False, True, True),      # gather kernel, reference weight order
        (1, 2, 7, 6, 10, 5, 1, 2, True, False, False),      # Cin % 4 != 0 -> gather kernel with K-contiguous weights
    ]
    for (S, B, H, ci, co, k, st, pad, ohwi, shared, tail) in cases:
        x = torch.randn(1 if shared else S, B, ci, H, H, generator=gen)
        w = torch.randn(S, co, ci, k, k, generator=gen) * 0.1
        bias, div, alpha, beta = (torch.randn(co, generator=gen) for _ in range(4))
        div = div.abs() + 0.5
        Ho = (H + 2 * pad - k) // st + 1
        res = torch.randn(S, B, co, Ho, Ho, generator=gen)
        dt = torch.float64 if acc64 else torch.float32
        ref = torch.stack([torch.nn.functional.conv2d(x[0 if shared else s].to(dt), w[s].to(dt), None, st, pad) for s in range(S)]).float()
        if tail:
            ref = ref / div.view(1, 1, -1, 1, 1)
            ref = ref + bias.view(1, 1, -1, 1, 1)
            ref = ref * alpha.view(1, 1, -1, 1, 1)
            ref = ref + beta.view(1, 1, -1, 1, 1)
            ref = torch.relu(ref + res)
        xg = x.permute(0, 1, 3, 4, 2).contiguous().cuda()                       # NHWC
        wg = (w.permute(0, 1, 3, 4, 2) if ohwi else w).reshape(S, -1).contiguous().cuda()
        kw = dict(acc64=acc64, ohwi=ohwi)
        if tail:
            kw.update(div=div.cuda(), bn=(alpha.cuda(), beta.cuda()), res=res.permute(0, 1, 3, 4, 2).contiguous().cuda())
        y = conv2d_f32(xg, wg, bias.cuda() if tail else None, ci, co, k, st, pad, tail, **kw)
        got = y.permute(0, 1, 4, 2, 3).cpu()
        scale = float(ref.abs().max())
        tol = (2e-7 if acc64 else 1e-5) * scale * (8 if tail else 1)      # the tail's roundings amplify a 1-ulp conv difference
        assert float((got - ref).abs().max()) <= tol, ((S, B, H, ci, co, k, st), float((got - ref).abs().max()), tol)


def test_generic_int8_conv_mfma_equals_scalar_form_and_integer_reference():
    """qbnn_conv2d_i8_generic_mc (byte / dword gathered MFMA implicit GEMM with zero-point corrections) against the scalar
    one-thread-per-output kernel bit for bit, and both against an int64 numpy restatement of sum (x - z_x)(w - z_w), on
    adversarial geometries and zero points: odd Cin, K not a multiple of 32, ragged pixel / channel tiles, padding + stride,
    1x1 linear shapes, extreme zero points, shared input and shared weights.  The bias centres the outputs so that no case
    saturates (a saturated output would hide the arithmetic)."""
    import ctypes as C
    from quantised_bayesian_nets_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(7)
    cases = [  # S, B, H, Cin, Cout, k, stride, pad, z_x, z_w, shared_x, shared_w, relu
        (3, 5, 9, 20, 50, 5, 1, 2, 60, 3, True, False, True),
        (2, 3, 12, 1, 20, 5, 1, 2, 0, -128, False, False, False),
        (2, 130, 1, 2450, 70, 1, 1, 0, 255, 127, False, True, True),
        (4, 7, 10, 7, 9, 3, 2, 1, 128, -5, False, False, False),
        (2, 2, 8, 24, 130, 3, 1, 1, 64, 11, True, True, True),
        (1, 66, 1, 13, 100, 1, 1, 0, 17, -77, False, False, True),
    ]
    for (S, B, H, ci, co, k, st, pad, zx, zw, sx, sw, relu) in cases:
        Ho = (H + 2 * pad - k) // st + 1
        xh = rng.integers(0, 256, size=(1 if sx else S, B, H, H, ci), dtype=np.uint8)
        wh = rng.integers(-128, 128, size=(1 if sw else S, co, k, k, ci), dtype=np.int8)
        # int64 restatement; padded taps contribute (x - z_x) = 0
        xp = np.pad(xh.astype(np.int64) - zx, ((0, 0), (0, 0), (pad, pad), (pad, pad), (0, 0)))
        wn = wh.astype(np.int64) - zw
        acc = np.zeros((S, B, Ho, Ho, co), np.int64)
        for s in range(S):
            for kh in range(k):
                for kw in range(k):
                    patch = xp[0 if sx else s, :, kh:kh + st * Ho:st, kw:kw + st * Ho:st, :]
                    acc[s] += np.einsum("bhwc,oc->bhwo", patch, wn[0 if sw else s, :, kh, kw, :])
        s_x, s_w = np.float32(0.02), np.float32(0.003)
        atw = s_x * s_w
        s_y = np.float32(atw * acc.std() / 25.0)
        bias_h = (-acc.mean(axis=(0, 1, 2, 3)) * np.float64(atw)).astype(np.float32)      # centres every channel on z_y
        d = _lib.ConvDesc()
        d.B, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.pad = B, H, H, ci, co, k, st, pad
        d.s_x, d.z_x, d.s_w, d.z_w, d.s_y, d.z_y = float(s_x), zx, float(s_w), zw, float(s_y), 60
        d.relu, d.a_hi, d.has_bias = int(relu), 127, 1
        x, w, bias = torch.from_numpy(xh).cuda(), torch.from_numpy(wh).cuda(), torch.from_numpy(bias_h).cuda()
        outs = []
        for fn in (L.qbnn_conv2d_i8_generic_mc, L.qbnn_conv2d_i8_generic_scalar_mc):
            y = torch.zeros((S, B, Ho, Ho, co), dtype=torch.uint8, device="cuda")
            _lib.check(fn(_lib.ptr(x), 0 if sx else x[0].numel(), _lib.ptr(w), 0 if sw else w[0].numel(), _lib.ptr(bias),
                          _lib.ptr(y), y[0].numel(), S, C.byref(d), _lib.current_stream()))
            outs.append(y.cpu().numpy())
        case = (S, B, H, ci, co, k, st, pad, zx, zw)
        assert np.array_equal(outs[0], outs[1]), case
        assert outs[0].std() > 5.0, case
        # requantisation as the kernels do it: fma(bias, 1 / (s_x s_w), acc) * (s_x s_w / s_y), rne, + z_y, clamp
        rcp, mult = np.float32(1.0) / atw, atw / s_y
        # (float)acc rounds first (|acc| may exceed 2^24); the fma's product is exact in float64, one rounding to fp32
        xf = (bias_h.astype(np.float64) * np.float64(rcp) + acc.astype(np.float32).astype(np.float64)).astype(np.float32)
        q = np.clip(60 + np.rint(xf * mult).astype(np.int64), 60 if relu else 0, 127).astype(np.uint8)
        assert np.array_equal(q, outs[0]), case


# ------------------------------------------------------------------------------------------ full-size parity (round 2)
def test_w4_full_size_fused_against_oracle_sample_by_sample():
    """BASELINE config 5 arithmetic (A7/W4: sampled weights clamped to [-8, 7]) at the full batch (B = 256), S = 6 samples
    of the fused path against the CPU oracle sample by sample: integer block outputs bit-exact, probabilities 1e-5."""
    import quantised_bayesian_nets_amd as q
    from conftest import load_golden
    from oracle import oracle as orc
    g = load_golden("resnet_bbb_a7w4.npz")
    m = _model(g)
    x = torch.randn(256, 3, 32, 32, generator=torch.Generator().manual_seed(2))
    xc = x.cuda()
    S, seed, begin = 6, 3, 0
    with q.mc_context(S, seed, begin):
        probs = m.forward_mc(xc)                      # fully fused path (stem + chains + down blocks)
        rec = {}
        m.forward_mc(xc, record=rec)                  # per-block launches, recording the block outputs
    net = orc.Int8ResNetOracle(g["state"], 7, 4)
    for s in range(S):
        orec = {}
        p_or = net.forward(x.numpy(), seed, begin + s, record=orec)
        for k in ("layers.0.out", "layers.3.1.out", "layers.4.1.out", "layers.5.0.out", "layers.6.1.out"):
            assert np.array_equal(rec[k][s].cpu().numpy(), orec["layers.0.out" if k == "layers.0.out" else k]), (k, s)
        np.testing.assert_allclose(probs[s].cpu().numpy(), p_or, rtol=RTOL, atol=1e-8)
        w = orec["layers.5.1.stem.0.w_q"]
        assert w.min() >= -8 and w.max() <= 7


@pytest.mark.parametrize("w_bits", [8, 4])
def test_high_sample_indices_against_oracle(w_bits):
    """Config 5 draws S = 1024 samples: Philox subsequences >= 256 (a second byte of the sample counter) must match the
    oracle too.  Sampler for every layer at sample_begin in {255, 256, 1023}, and the fused path end to end at B = 256."""
    import quantised_bayesian_nets_amd as q
    from conftest import load_golden
    from oracle import oracle as orc
    g = load_golden(f"resnet_bbb_a7w{w_bits}.npz")
    m = _model(g)
    net = orc.Int8ResNetOracle(g["state"], 7, w_bits)
    seed = 3
    for begin in (255, 256, 1023):
        for name, layer, (pfx, *_r) in zip(m.stochastic_layer_names(), m.stochastic_layers(), net.table):
            w = layer.sample_weights("cuda", samples=2, seed=seed, sample_begin=begin).cpu().numpy()
            for i in range(2):
                assert np.array_equal(w[i], _pack(layer, net.layers[pfx].sample(seed, begin + i))), (name, begin + i)
    x = torch.randn(256, 3, 32, 32, generator=torch.Generator().manual_seed(5))
    xc = x.cuda()
    with q.mc_context(2, seed, 255):
        p = m.forward_mc(xc)                          # samples 255, 256
    with q.mc_context(1, seed, 1023):
        p1023 = m.forward_mc(xc)
    for s, got in ((255, p[0]), (256, p[1]), (1023, p1023[0])):
        np.testing.assert_allclose(got.cpu().numpy(), net.forward(x.numpy(), seed, s), rtol=RTOL, atol=1e-8)
    # a 1024-sample evaluation sharded as 8 ranks would shard it: rank 7 owns [896, 1024)
    from quantised_bayesian_nets_amd.mc import shard_samples
    assert shard_samples(1024, 7, 8) == (896, 128)
    with q.mc_context(128, seed, 896):
        tail = m.forward_mc(xc)
    assert torch.equal(tail[127], p1023[0])


@pytest.mark.parametrize("a_bits,w_bits", [(3, 8), (7, 3), (5, 8), (7, 6)], ids=["a3w8", "a7w3", "a5w8", "a7w6"])
def test_bit_width_sweep_full_batch_against_oracle(a_bits, w_bits):
    """The reference's sweep (experiments/run_all_quant.sh:11-37) away from the two BASELINE points, at the full batch: A3 and A5
    move every activation clamp (src/utils.py:25-30: [0, 7] / [0, 31]) and the accumulator bound of the 1.5 * 2^23 start, W3 / W6 the
    sampled-weight clamp ([-4, 3] / [-32, 31], src/utils.py:32-37).  Fused path and per-block launches, B = 256, two samples, against
    the CPU oracle: integer block outputs bit-exact, probabilities 1e-5."""
    import quantised_bayesian_nets_amd as q
    from conftest import load_golden
    from oracle import oracle as orc
    g = load_golden("resnet_bbb_a%dw%d.npz" % (a_bits, w_bits))
    m = _model(g)
    x = torch.randn(256, 3, 32, 32, generator=torch.Generator().manual_seed(11 + a_bits + w_bits))
    xc = x.cuda()
    S, seed, begin = 2, 3, 40
    with q.mc_context(S, seed, begin):
        probs = m.forward_mc(xc)
        rec = {}
        m.forward_mc(xc, record=rec)
    net = orc.Int8ResNetOracle(g["state"], a_bits, w_bits)
    a_hi, (w_lo, w_hi) = orc.UINT_BOUNDS[a_bits][1], orc.INT_BOUNDS[w_bits]
    for s in range(S):
        orec = {}
        p_or = net.forward(x.numpy(), seed, begin + s, record=orec)
        for k in ("layers.0.out", "layers.3.1.out", "layers.4.0.out", "layers.4.1.out", "layers.5.0.out", "layers.5.1.out", "layers.6.0.out",
                  "layers.6.1.out"):
            got = rec[k][s].cpu().numpy()
            assert np.array_equal(got, orec[k]), (k, s)
            assert got.max() <= a_hi
        np.testing.assert_allclose(probs[s].cpu().numpy(), p_or, rtol=RTOL, atol=1e-8)
        w = orec["layers.5.1.stem.0.w_q"]
        assert w.min() >= w_lo and w.max() <= w_hi and (w_bits == 8 or w.min() == w_lo or w.max() == w_hi)      # the clamp is hit


def test_ensemble_16_members_full_batch_against_oracle(golden_ensemble):
    """BASELINE config 4 size: 16 members at B = 256.  Every member's probabilities against the deterministic-member oracle
    (integer logits path bit-exact -> 1e-5 on probabilities), through forward_mc (all members of the mc_context) -- first
    call and replays -- and through the reference's round-robin forward()."""
    import quantised_bayesian_nets_amd as q
    from conftest import synth_ensemble_members
    from oracle import oracle as orc
    n = 16
    members = synth_ensemble_members(golden_ensemble, n)
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, model="conv_resnet_sgld", samples=n)
    net = q.ModelFactory.get_model("conv_resnet_sgld", [1, 3, 32, 32], 10, True, args, training_mode=False).load_reference_state(members)
    x = torch.randn(256, 3, 32, 32, generator=torch.Generator().manual_seed(2))
    xc = x.cuda()
    ref = np.stack([orc.Int8ResNetDetOracle(st, 7).forward(x.numpy()) for st in members])
    assert np.abs(ref[0] - ref[5]).max() > 1e-3                    # the synthetic members really differ
    for _ in range(3):                                              # eager / captured / replayed
        with q.mc_context(n, 0, 0):
            probs = net.forward_mc(xc)
        np.testing.assert_allclose(probs.cpu().numpy(), ref, rtol=RTOL, atol=1e-8)
    with q.mc_context(5, 0, 9):                                     # a rank's shard: members 9..13
        part = net.forward_mc(xc)
    assert torch.equal(part, probs[9:14])
    mean = q.mc_predict(net, xc, n, 0)
    np.testing.assert_allclose(mean.cpu().numpy(), ref.mean(0), rtol=RTOL, atol=1e-7)
    # the per-member launch chains (captured graphs on side streams) give the same bits as the fused multi-call launches
    net.fused_members = False
    for _ in range(2):
        with q.mc_context(n, 0, 0):
            assert torch.equal(net.forward_mc(xc), probs)
    with q.mc_context(n + 3, 0, 0):                                 # more samples than members: the round-robin wraps
        wrap = net.forward_mc(xc)
    assert torch.equal(wrap[:n], probs) and torch.equal(wrap[n:], probs[:3])
    net.fused_members = True
    with q.mc_context(3, 0, 14):                                    # wraps inside a fused call: members 14, 15, 0
        assert torch.equal(net.forward_mc(xc), torch.cat([probs[14:], probs[:1]]))
    xr = torch.randn(37, 3, 32, 32, generator=torch.Generator().manual_seed(9)).cuda()      # ragged batch through the fused launches
    with q.mc_context(n, 0, 0):
        pr = net.forward_mc(xr)
    net.fused_members = False
    with q.mc_context(n, 0, 0):
        assert torch.equal(net.forward_mc(xr), pr)
    net.fused_members = True
    net.counter = 0
    for i in range(n):
        np.testing.assert_allclose(net(xc).cpu().numpy(), ref[i], rtol=RTOL, atol=1e-8)


def test_ensemble_members_with_their_own_qparams(golden_ensemble):
    """Members whose quantisation parameters ALL differ -- input QuantStub scale and zero point (three distinct values over 8 members: the
    layer-0 patches are shared per distinct value), every conv's output scale, the Add scales -- through the one-launch-per-stage form
    (argument blocks in device memory), the by-value multi-call form and the per-member chains: bit-identical to each other and
    1e-5 to the deterministic-member oracle."""
    import quantised_bayesian_nets_amd as q
    from conftest import synth_ensemble_members
    from oracle import oracle as orc
    n = 8
    members = [dict(m) for m in synth_ensemble_members(golden_ensemble, n)]
    for i, st in enumerate(members):
        f = 1.0 + 0.03 * (i % 3)
        for k in list(st):
            v = np.asarray(st[k])
            if k == "quant.scale" or (k.endswith(".scale") and v.dtype.kind == "f" and v.size == 1):
                st[k] = (v * np.float32(f if k == "quant.scale" else 1.0 + 0.01 * ((i + len(k)) % 5))).astype(v.dtype)
            elif k == "quant.zero_point":
                st[k] = (v + (i % 3)).astype(v.dtype)
    assert len({(float(np.asarray(st["quant.scale"]).reshape(-1)[0]), int(np.asarray(st["quant.zero_point"]).reshape(-1)[0])) for st in members}) == 3
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, model="conv_resnet_sgld", samples=n)
    x = torch.randn(21, 3, 32, 32, generator=torch.Generator().manual_seed(4))
    xc = x.cuda()
    ref = np.stack([orc.Int8ResNetDetOracle(st, 7).forward(x.numpy()) for st in members])
    outs = []
    for mode in ("prepared", "by_value", "chains"):
        net = q.ModelFactory.get_model("conv_resnet_sgld", [1, 3, 32, 32], 10, True, args, training_mode=False).load_reference_state(members)
        net.prepared_launches = mode == "prepared"
        net.fused_members = mode != "chains"
        with q.mc_context(n, 0, 0):
            outs.append(net.forward_mc(xc))
        if mode == "prepared":
            plan = next(iter(net._plans.values()))
            assert plan["col"].shape[0] == 3 and len(plan["dev_steps"]) == 7
    np.testing.assert_allclose(outs[0].cpu().numpy(), ref, rtol=RTOL, atol=1e-8)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


_DIST_WORKER = r"""
import os, sys, types, numpy as np, torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from conftest import load_golden
import quantised_bayesian_nets_amd as q
torch.cuda.set_device(0)
g = load_golden("resnet_bbb_a7w8.npz")
args = types.SimpleNamespace(activation_precision=7, weight_precision=8)
m = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, args).load_reference_state(g["state"])
x = torch.randn(64, 3, 32, 32, generator=torch.Generator().manual_seed(2)).cuda()
mean0, var0 = q.mc_predict(m, x, 7, 3, return_var=True)            # no process group
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
os.environ["QBNN_BENCH_FORCE_DIST"] = "1"                          # take the all-reduce with one rank too
mean1, var1 = q.mc_predict(m, x, 7, 3, return_var=True)
torch.cuda.synchronize()
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
assert torch.equal(mean0, mean1) and torch.equal(var0, var1)
print("DIST-OK", flush=True)
os._exit(0)        # no process-group teardown: RCCL's can hang at exit on a one-GPU box, and nothing is left to check
"""


def test_rccl_path_one_rank_equals_no_dist(tmp_path):
    """The RCCL leg of mc_predict (init_process_group('nccl') + the sum all-reduce of the moments) with one rank, in a child
    process: bit-identical to the evaluation without a process group."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "dist_worker.py"
    script.write_text(_DIST_WORKER)
    import socket
    out = err = ""
    for attempt in range(2):                      # RCCL's one-rank bootstrap has been seen to hang once on a fresh box: bounded, retried once
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        p = subprocess.Popen([sys.executable, str(script), root], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            out, err = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            p.kill()
            out, err = p.communicate()
            err += "\n[attempt %d: no result after 150 s, child killed]" % attempt
            continue
        break
    assert "DIST-OK" in out, out[-2000:] + err[-4000:]


def test_layers_take_and_return_torch_quantized_tensors(golden):
    """The layer-level drop-in seam (reference conv_q.py:107-125, linear_q.py:80-94): `layer(x)` with x a torch quint8 NCHW
    tensor -- what the reference's graph passes between modules, followed by its own `clamp_activation` (src/utils.py:25-30,
    which tests `x.dtype == torch.quint8`) -- runs ONE stochastic forward on the GPU and hands a torch quint8 tensor back.
    Checked on the recorded layer inputs / outputs of the reference (sample 0 of the golden noise stream)."""
    import quantised_bayesian_nets_amd as q
    g = golden
    m = _model(g)
    st, rec, seed = g["state"], g["rec"], g["meta"]["philox_seed"]
    a_hi = 2 ** g["meta"]["a_bits"] - 1

    def qt(nhwc, scale, zp):
        t = torch.from_numpy(np.ascontiguousarray(nhwc.transpose(0, 3, 1, 2)) if nhwc.ndim == 4 else nhwc.copy())
        return torch._make_per_tensor_quantized_tensor(t, float(scale), int(zp))

    def clamp_activation(x):                      # the reference's helper, applied by its graph after every module
        assert x.dtype == torch.quint8
        return torch.clamp(x, (0 - x.q_zero_point()) * x.q_scale(), (a_hi - x.q_zero_point()) * x.q_scale())

    cases = [("layers.0", m.layers[0], rec["quant.out"], st["quant.scale"].reshape(-1)[0], st["quant.zero_point"].reshape(-1)[0]),
             ("layers.3.0.stem.0", m.layers[3][0].stem[0], rec["layers.0.out"], st["layers.0.scale"], st["layers.0.zero_point"]),
             ("layers.4.0.shortcut.0", m.layers[4][0].shortcut[0], rec["layers.3.1.out"], st["layers.3.1.add.add.scale"], st["layers.3.1.add.add.zero_point"]),
             ("layers.9", m.layers[9], rec["layers.7.out"].reshape(rec["layers.7.out"].shape[0], -1), st["layers.6.1.add.add.scale"], st["layers.6.1.add.add.zero_point"])]
    for name, layer, x_in, s_in, z_in in cases:
        x = qt(x_in, s_in, z_in)                                      # a CPU quint8 tensor, as in the reference's int8 graph
        with q.mc_context(1, seed, 0):
            y = clamp_activation(layer(x))
        assert y.dtype == torch.quint8 and y.device == x.device and y.shape[0] == x.shape[0]
        assert y.q_scale() == pytest.approx(float(st[name + ".scale"]), rel=0, abs=0) and y.q_zero_point() == int(st[name + ".zero_point"])
        got = y.int_repr().numpy()
        got = got.transpose(0, 2, 3, 1) if got.ndim == 4 else got
        assert np.array_equal(got, rec[name + ".out"]), name
    with pytest.raises(ValueError):
        m.layers[3][0].stem[0](qt(rec["layers.7.out"].reshape(4, -1), 0.1, 3))        # conv_q.py:190-191: "Input shape must be `(N, C, H, W)`!"


_SWITCH_WORKER = r"""
import os, sys, types, numpy as np, torch
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from conftest import load_golden
import quantised_bayesian_nets_amd as q
g = load_golden("resnet_bbb_a7w8.npz")
args = types.SimpleNamespace(activation_precision=7, weight_precision=8)
m = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, args).load_reference_state(g["state"])
x = torch.from_numpy(g["x"]).cuda()
S, seed = g["probs"].shape[0], g["meta"]["philox_seed"]
with q.mc_context(S, seed, 0):
    p = m.forward_mc(x)
np.testing.assert_allclose(p.cpu().numpy(), g["probs"], rtol=1e-5, atol=1e-8)
xb = torch.randn(74, 3, 32, 32, generator=torch.Generator().manual_seed(4)).cuda()
with q.mc_context(5, seed, 40):
    m.fuse_blocks = False
    ref = m.forward_mc(xb)
    m.fuse_blocks = True
    assert torch.equal(m.forward_mc(xb), ref)
gl = load_golden("lenet_mc_a7w8.npz")
la = types.SimpleNamespace(activation_precision=7, weight_precision=8, p=0.2)
lm = q.ModelFactory.get_model("conv_lenet_mc", [1, 1, 28, 28], 10, True, la).load_reference_state(gl["state"])
with q.mc_context(gl["probs"].shape[0], gl["meta"]["philox_seed"], 0):
    pl = lm.forward_mc(torch.from_numpy(gl["x"]).cuda())
np.testing.assert_allclose(pl.cpu().numpy(), gl["probs"], rtol=1e-5, atol=1e-8)
# the ensemble's prepared multi-call launches (argument blocks in device memory) honour the switches too (advisor, round 3)
from conftest import load_ensemble_fixture
ge = load_ensemble_fixture()
ea = types.SimpleNamespace(activation_precision=7, weight_precision=8, model="conv_resnet_sgld", samples=len(ge["members"]))
net = q.ModelFactory.get_model("conv_resnet_sgld", [1, 3, 32, 32], 10, True, ea, training_mode=False).load_reference_state(ge["members"])
with q.mc_context(len(ge["members"]), 0, 0):
    pe = net.forward_mc(torch.from_numpy(ge["x"]).cuda())
np.testing.assert_allclose(pe.cpu().numpy(), ge["probs"], rtol=1e-5, atol=1e-8)
print("SWITCH-OK")
"""


@pytest.mark.parametrize("switch", ["QBNN_NO_PINGPONG", "QBNN_NO_STEM_FUSION", "QBNN_GENERIC_NAIVE", "QBNN_W16=0",
                                    "QBNN_C48=0", "QBNN_W16_MAGIC=0", "QBNN_D24=0", "QBNN_CHAIN_2WG=0", "QBNN_DOWN_R16=0", "QBNN_HEAD_POOL=0"])
def test_environment_switches_give_the_same_results(switch, tmp_path):
    """The A/B switches of README.md select other kernels for the same arithmetic (weights-stationary instead of ping-pong
    48-channel block; layers.0 as its own launch; the scalar any-geometry conv; the 8-wave layer-1 kernel instead of the 16-wave one; the round-3 forms of the wide down-sampling and identity blocks):
    each, in a child process (the switches are read once), reproduces the golden probabilities and the fused == layer-wise identity."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "switch_worker.py"
    script.write_text(_SWITCH_WORKER)
    name, _, value = switch.partition("=")
    r = subprocess.run([sys.executable, str(script), root], env=dict(os.environ, **{name: value or "1"}), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SWITCH-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_F32_SWITCH_WORKER = r"""
import os, sys, types, numpy as np, torch
root, out = sys.argv[1], sys.argv[2]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from conftest import load_golden
import quantised_bayesian_nets_amd as q
res = {}
g = load_golden("resnet_bbb_f32.npz")
m = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, False, types.SimpleNamespace(sigma_prior=-2.0)).load_reference_state(g["state"])
x = torch.randn(70, 3, 32, 32, generator=torch.Generator().manual_seed(9)).cuda()
with q.mc_context(3, 11, 2):
    res["f32"] = m.forward_mc(x).cpu().numpy()
gq = load_golden("resnet_bbb_qat.npz")
qa = types.SimpleNamespace(sigma_prior=-2.0, activation_precision=7, weight_precision=8, qat_eval=True)
mq = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, qa).load_reference_state(gq["state"])
with q.mc_context(3, 11, 2):
    res["qat"] = mq.forward_mc(x).cpu().numpy()
with q.mc_context(2, 12, 0):                                  # a second pass: the observers' states after the first one matter
    res["qat2"] = mq.forward_mc(x).cpu().numpy()
res["qat_observers"] = np.array([v for k, v in sorted(mq.prepared_state().items()) if k.endswith("min_val") or k.endswith("max_val")], np.float32)
import numpy as _np, os as _os
for name, model, shape, xin in (("lenet_bbb_qat.npz", "conv_lenet_bbb", [1, 28, 28], torch.rand(5, 1, 28, 28, generator=torch.Generator().manual_seed(3))),
                                ("mlp_bbb_qat.npz", "linear_bbb", [13], torch.randn(9, 13, generator=torch.Generator().manual_seed(4)))):
    d = _np.load(_os.path.join(root, "tests", "golden", name))
    st = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
    ms = q.ModelFactory.get_model(model, shape, 1 if model == "linear_bbb" else 10, True, qa).load_reference_state(st)
    for rep in range(2):
        with q.mc_context(3, 21 + rep, 0):
            o = ms.forward_mc(xin.cuda())
        res["%s_%d" % (model, rep)] = (torch.cat(list(o), -1) if isinstance(o, tuple) else o).cpu().numpy()
np.savez(out, **res)
print("F32-SWITCH-OK")
"""


@pytest.mark.parametrize("switch", ["QBNN_F32_TAPMASK=0", "QBNN_QAT_PRESAMPLE=0", "QBNN_Q8_TILED=0", "QBNN_QAT_WBATCH=0"])
def test_float_path_switches_give_the_same_bits(switch, tmp_path):
    """The fp32 / fp64 conv's two gather forms (per-row tap masks against per-element bounds compares), the QAT weight pipelines on side
    streams against in line, the QAT 3 x 3 convs LDS-tiled (round 6, csrc/qbnn_q8t.hip) against the gather forms of round 5 (same integer sums, same tail),
    and all layers' weight pipelines in four launches (qbnn_qat_weights_mc) against ~15 launches per layer: the float BBB ResNet and the QAT evaluation, ragged batch of 70, give bit-identical probabilities either way."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "f32_switch_worker.py"
    script.write_text(_F32_SWITCH_WORKER)
    outs = []
    name, _, value = switch.partition("=")
    base = {"QBNN_QAT_WBATCH": "0"} if name == "QBNN_QAT_PRESAMPLE" else {}      # (the side-stream pipelines are what runs when the batched form is off)
    for env in (base, dict(base, **{name: value})):
        out = tmp_path / ("probs_%d.npz" % len(outs))
        r = subprocess.run([sys.executable, str(script), root, str(out)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "F32-SWITCH-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        outs.append(np.load(out))
    for k in outs[0].files:                  # float ResNet, QAT ResNet (two passes + every observer's state), QAT LeNet and MLP (two passes each)
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_qat_convs_on_the_int8_pipe_agree_with_the_fp64_sums(tmp_path):
    """Round 5: the QAT convs as exact integer sums on the int8 matrix pipe (both operands are fake-quantised tensors: integers on a per-sample
    grid; csrc/qbnn_f32.hip conv2d_q8_kernel) against the fp64 sums of the fp32-rounded operands (QBNN_QAT_I8=0) on the QAT ResNet at a ragged batch of
    70, three samples with live observers.  Per conv the two differ by the operands' own rounding (<= 1.2e-7 relative) -- but a prepared network is not
    continuous in that: a conv output within 1e-7 of a rounding boundary falls on the other side, the activation moves by a whole quantisation step, and
    through the observers' EMA every later scale moves with it.  The reference shows exactly this sensitivity between its own two conv backends
    (`refspread.max_abs` = 2e-3 on resnet_bbb_qat.npz at B = 2; 3.9e-4 / 5.2e-4 on the MC-Dropout / SGHMC fixtures), so the yardstick here is that
    spread, not fp32 epsilon: mean |difference| of the probabilities below 2e-3, the largest below 5e-2, the arg-max class equal for >= 97 % of the
    (sample, image) pairs.  (Either form against the REFERENCE's recorded run, to 1e-5 + 1e-6: test_qat_eval_with_live_observers_matches_reference.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "f32_switch_worker.py"
    script.write_text(_F32_SWITCH_WORKER)
    outs = []
    for env in ({"QBNN_QAT_I8": "1"}, {"QBNN_QAT_I8": "0"}):
        out = tmp_path / ("probs_%d.npz" % len(outs))
        r = subprocess.run([sys.executable, str(script), root, str(out)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "F32-SWITCH-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        outs.append(np.load(out))
    assert np.array_equal(outs[0]["f32"], outs[1]["f32"])
    d = np.abs(outs[0]["qat"] - outs[1]["qat"])
    same = float((outs[0]["qat"].argmax(-1) == outs[1]["qat"].argmax(-1)).mean())
    print("int8 pipe vs fp64 sums: max abs diff %.3g, mean %.3g, same arg-max %.4f" % (float(d.max()), float(d.mean()), same))
    assert float(d.mean()) < 2e-3 and float(d.max()) < 5e-2 and same >= 0.97


def test_graphed_predictor_equals_eager(golden_w8, golden_lenet_mc, golden_mlp_f32, golden_ensemble, golden_mlp_bbb_q, golden_lenet_bbb):
    """One captured HIP graph per (model, input shape), replayed with new inputs and new seeds (read from device memory):
    bit-identical to the eager `mc_predict` for the int8 BBB ResNet, the MC-Dropout LeNet (dropout masks), the fp32 BBB MLP
    (regression reduction), the small int8 BBB graphs and the 16-member ensemble."""
    import quantised_bayesian_nets_amd as q
    from conftest import synth_ensemble_members
    gen = torch.Generator().manual_seed(31)
    # int8 BBB ResNet
    m = _model(golden_w8)
    gp = q.GraphedPredictor(m, 7, return_var=True)
    for seed, sb in ((3, 0), (2 ** 40 + 17, 5), (3, 0)):
        x = torch.randn(64, 3, 32, 32, generator=gen).cuda()
        mean, var = gp(x, seed, sample_begin=sb)
        with q.mc_context(7, seed, sb):
            probs = m.forward_mc(x)
        assert torch.equal(mean, q.mc_predict(m, x, 7, seed, return_var=True)[0]) if sb == 0 else True
        np.testing.assert_allclose(mean.cpu().numpy(), probs.double().mean(0).cpu().numpy(), rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(var.cpu().numpy(), probs.double().var(0).cpu().numpy(), rtol=1e-5, atol=1e-12)
    # MC-Dropout LeNet: the masks follow the device seed too
    g = golden_lenet_mc
    la = types.SimpleNamespace(activation_precision=7, weight_precision=8, p=0.2)
    lm = q.ModelFactory.get_model("conv_lenet_mc", [1, 1, 28, 28], 10, True, la).load_reference_state(g["state"])
    gl = q.GraphedPredictor(lm, 9)
    for seed in (g["meta"]["philox_seed"], 12345):
        x = torch.rand(128, 1, 28, 28, generator=gen).cuda()
        assert torch.equal(gl(x, seed), q.mc_predict(lm, x, 9, seed))
    # fp32 BBB MLP, regression reduction
    gm = golden_mlp_f32
    mm = q.ModelFactory.get_model("linear_bbb", [gm["in_dim"]], 1, False, types.SimpleNamespace(sigma_prior=-2.0)).load_reference_state(gm["state"])
    gr = q.GraphedPredictor(mm, 10, regression=True)
    for seed in (gm["seed"], 99):
        x = torch.randn(1000, gm["in_dim"], generator=gen).cuda()
        a, b = gr(x, seed)
        c, d = q.mc_predict_regression(mm, x, 10, seed)
        assert torch.equal(a, c) and torch.equal(b, d)
    # the small int8 BBB graphs (sampled weights in the fragment / row-major layouts of their own kernels): MLP (regression) and LeNet
    qa = types.SimpleNamespace(activation_precision=7, weight_precision=8)
    gq = golden_mlp_bbb_q
    mq = q.ModelFactory.get_model("linear_bbb", [13], 1, True, qa).load_reference_state(gq["state"])
    gq_ = q.GraphedPredictor(mq, 10, regression=True)
    for seed in (5, 2 ** 33 + 1):
        x = torch.randn(1000, 13, generator=gen).cuda()
        a, b = gq_(x, seed)
        c, d = q.mc_predict_regression(mq, x, 10, seed)
        assert torch.equal(a, c) and torch.equal(b, d)
    gb = golden_lenet_bbb
    lb = q.ModelFactory.get_model("conv_lenet_bbb", [1, 1, 28, 28], 10, True, qa).load_reference_state(gb["state"])
    glb = q.GraphedPredictor(lb, 6)
    for seed in (7, 70):
        x = torch.rand(40, 1, 28, 28, generator=gen).cuda()
        assert torch.equal(glb(x, seed), q.mc_predict(lb, x, 6, seed))
    # ensemble (no noise at all: the graph only saves the launches)
    n = 16
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, model="conv_resnet_sgld", samples=n)
    net = q.ModelFactory.get_model("conv_resnet_sgld", [1, 3, 32, 32], 10, True, args, training_mode=False).load_reference_state(
        synth_ensemble_members(golden_ensemble, n))
    ge = q.GraphedPredictor(net, n)
    for _ in range(2):
        x = torch.randn(64, 3, 32, 32, generator=gen).cuda()
        assert torch.equal(ge(x, 0), q.mc_predict(net, x, n, 0))


def test_small_sghmc_templates_match_reference():
    """reference sgld.Network's other two templates (models_sgld.py:13-97, :219-226): `conv_lenet_sgld` (deterministic int8 LeNet
    members, softmax in the wrapper) and `linear_sgld` (MLP members returning (mu, exp(log_var))), members = the MC samples."""
    import os
    import quantised_bayesian_nets_amd as q
    from conftest import GOLDEN

    def members(d):
        n = int(d["meta.members"])
        return [{k[len(f"member{i}/"):]: d[k] for k in d.files if k.startswith(f"member{i}/")} for i in range(n)]

    d = np.load(os.path.join(GOLDEN, "ensemble_lenet_a7w8.npz"))
    n = int(d["meta.members"])
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, model="conv_lenet_sgld", samples=n, task="classification")
    net = q.ModelFactory.get_model("conv_lenet_sgld", [1, 1, 28, 28], 10, True, args, training_mode=False).load_reference_state(members(d))
    x = torch.from_numpy(d["x"]).cuda()
    rec = {}
    with q.mc_context(n, 0, 0):
        probs = net.forward_mc(x, record=rec)
    for k in ("quant.out", "layers.0.out", "layers.1.out", "layers.2.out", "layers.3.out", "layers.5.out", "layers.7.out"):
        got = rec[k][0].cpu().numpy()
        assert np.array_equal(got.reshape(d["rec/" + k].shape), d["rec/" + k]), k
    np.testing.assert_allclose(probs.cpu().numpy(), d["probs"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(q.mc_predict(net, x, n, 0).cpu().numpy(), d["probs"].mean(0), rtol=RTOL, atol=1e-8)
    outs = [net(x).cpu().numpy() for _ in range(n + 1)]                     # the wrapper's round-robin call contract
    np.testing.assert_allclose(np.stack(outs[:n]), d["probs"], rtol=RTOL, atol=1e-8)
    assert np.array_equal(outs[0], outs[n])

    d = np.load(os.path.join(GOLDEN, "ensemble_mlp_a7w8.npz"))
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, model="linear_sgld", samples=n, task="regression")
    net = q.ModelFactory.get_model("linear_sgld", [13], 1, True, args, training_mode=False).load_reference_state(members(d))
    x = torch.from_numpy(d["x"]).cuda()
    rec = {}
    with q.mc_context(n, 0, 0):
        mu, var = net.forward_mc(x, record=rec)
    for k in ("quant.out", "layers.0.out", "layers.2.out", "layers.4.out", "mu.out", "log_var.out"):
        assert np.array_equal(rec[k][0].cpu().numpy().reshape(d["rec/" + k].shape), d["rec/" + k]), k
    np.testing.assert_allclose(mu.cpu().numpy(), d["mu"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(var.cpu().numpy(), d["var"], rtol=1e-5, atol=1e-8)
    mean, pv = q.mc_predict_regression(net, x, n, 0)
    np.testing.assert_allclose(mean.cpu().numpy(), d["mu"].mean(0), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pv.cpu().numpy(), d["mu"].astype(np.float64).var(0, ddof=1) + d["var"].mean(0), rtol=1e-4, atol=1e-8)
    m0, v0 = net(x)
    np.testing.assert_allclose(m0.cpu().numpy(), d["mu"][0], rtol=1e-5, atol=1e-6)


def test_native_prepare_calibrate_convert_pipeline():
    """SURVEY 8f row 4 end to end without the reference: float state -> `prepare_model_state` -> calibration by live-observer
    evaluation forwards on the GPU -> `convert_model_state`; against what the REFERENCE produced from the same float model with
    prepare_model -> 3 eval forwards (same injected eps) -> convert (tests/golden/make_golden_prepare.py).

    Tolerances are measured, not assumed (tests/golden/make_golden_prepare_spread.py, resnet_bbb_prepare_spread.npz): the reference's
    own calibration depends on its conv backend -- its observers start unseen, every fake-quantised activation feeds the next
    observer, and oneDNN sums the fp32 products in another order than ATen's own convolution: mkldnn on / off x 1 / 3 / 8 threads fall
    into exactly two groups, up to 3.0 % of an observer's range (4.1 % in a converted scale, 1 in a zero point) apart.  The recorded
    fixture is the plain-ATen run (mkldnn off, thread-count independent); the build (fp64 conv sums under the fake-quantisers) lands
    on it: 2.7e-7 of the range on the worst activation observer, scales to 2.7e-7, every zero point equal, 41 of the 42 int8 tensors equal and
    the 42nd (named below) in all but one element."""
    import hashlib
    import os
    import quantised_bayesian_nets_amd as q
    from conftest import GOLDEN
    from quantised_bayesian_nets_amd.convert import prepare_model_state, calibrate, convert_model_state, convert_model
    d = np.load(os.path.join(GOLDEN, "resnet_bbb_f32.npz"))
    fstate = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
    ref = np.load(os.path.join(GOLDEN, "resnet_bbb_prepare_calibrate.npz"))
    S, seed = int(ref["meta.samples"]), int(ref["meta.philox_seed"])
    aq = types.SimpleNamespace(sigma_prior=-2.0, activation_precision=7, weight_precision=8, qat_eval=True)
    m = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, aq).load_reference_state(prepare_model_state(fstate))
    x = torch.from_numpy(d["x"]).cuda()
    calibrate(m, [x] * S, seed)
    st = m.prepared_state()
    weight_like = lambda k: any(t in k for t in ("weight_fake_quant", "std_fake_quant", "mul_noise", "add_weight"))
    n_w = n_a = 0
    worst_act = 0.0
    for k in ref.files:                                       # observers after calibration
        if not k.startswith("calibrated/") or not k.endswith("min_val"):
            continue
        kk = k[len("calibrated/"):]
        lo, hi = float(ref[k]), float(ref[k.replace("min_val", "max_val")])
        glo, ghi = float(st[kk]), float(st[kk.replace("min_val", "max_val")])
        if weight_like(kk):
            assert abs(glo - lo) <= 1e-5 * max(1e-3, abs(lo)) + 1e-9 and abs(ghi - hi) <= 1e-5 * max(1e-3, abs(hi)) + 1e-9, (kk, glo, lo, ghi, hi)
            n_w += 1
        else:
            rng = max(hi, 0.0) - min(lo, 0.0)
            dev = max(abs(glo - lo), abs(ghi - hi)) / rng
            worst_act = max(worst_act, dev)
            assert dev <= 1e-5, (kk, glo, lo, ghi, hi)       # measured: 2.7e-7 (the reference's two backends: up to 3.0e-2 apart)
            n_a += 1
    assert n_w == 4 * 21 and n_a == 21 + 8 + 1                 # 21 layers x (weight, std, mul, add) ; 21 outputs + 8 Adds + the stub
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8)
    conv = convert_model_state(st, args)
    n_same = n_int8 = 0
    differing = []
    for k in ref.files:
        if not k.startswith("converted/"):
            continue
        key = k[len("converted/"):]
        if key.endswith(".sha1"):
            base = key[:-len(".sha1")]
            n_int8 += 1
            same = hashlib.sha1(np.ascontiguousarray(conv[base]).tobytes()).hexdigest() == str(ref[k])
            n_same += int(same)
            if not same:
                differing.append(base)
        elif key.endswith("scale"):                            # weight-side and activation-side alike
            np.testing.assert_allclose(float(np.asarray(conv[key]).reshape(-1)[0]), float(np.asarray(ref[k]).reshape(-1)[0]), rtol=1e-5, err_msg=key)
        elif key.endswith("zero_point"):
            assert int(np.asarray(conv[key]).reshape(-1)[0]) == int(np.asarray(ref[k]).reshape(-1)[0]), key
    # int8 weight / std tensors bit-identical to the reference's, with ONE named exception: layers.6.1.stem.0.weight (192 x 192 x 3 x 3), whose
    # BN-folded weight observer lands one fp32 ulp from the reference's (scale 0.0018490724 against 0.0018490722, inside the 1e-5 bound
    # above) -- ONE of its 331,776 elements sits on a rounding tie and comes out one step higher.  The fixture holds that tensor in full.
    assert n_int8 == 42 and n_same >= 41 and set(differing) <= {"layers.6.1.stem.0.weight"}, (n_same, n_int8, differing)
    for key in differing:
        dd = np.asarray(conv[key]).astype(np.int32) - ref["converted/" + key].astype(np.int32)
        assert int((dd != 0).sum()) <= 1 and int(np.abs(dd).max()) <= 1, (key, int((dd != 0).sum()), int(np.abs(dd).max()))
    for key in ("layers.0.weight", "layers.4.0.shortcut.0.weight", "layers.9.weight", "layers.0.std"):
        assert np.array_equal(np.asarray(conv[key]).astype(np.int32), ref["converted/" + key].astype(np.int32)), key
    # the committed reference-vs-reference measurement: the oneDNN runs sit up to ~3 % of a range away from the ATen runs (and from the build)
    spr = np.load(os.path.join(GOLDEN, "resnet_bbb_prepare_spread.npz"))
    cfg = [str(c) for c in spr["meta.configs"]]
    far = {}
    for i, c in enumerate(cfg):
        w = 0.0
        for k in ref.files:
            if k.startswith("calibrated/") and k.endswith("min_val") and not weight_like(k):
                kk = k[len("calibrated/"):]
                lo, hi = float(spr["run%d/calibrated/%s" % (i, kk)]), float(spr["run%d/calibrated/%s" % (i, kk.replace("min_val", "max_val"))])
                rng = max(hi, 0.0) - min(lo, 0.0)
                w = max(w, abs(float(st[kk]) - lo) / rng, abs(float(st[kk.replace("min_val", "max_val")]) - hi) / rng)
        far[c] = w
    assert all(v <= 1e-5 for c, v in far.items() if "mkldnn=0" in c) and all(0.01 < v < 0.05 for c, v in far.items() if "mkldnn=1" in c), far
    # and the converted model runs
    model = convert_model(m, "conv_resnet_bbb", [1, 3, 32, 32], 10, args)
    p = q.mc_predict(model, x, 4, 1)
    np.testing.assert_allclose(p.sum(-1).cpu().numpy(), 1.0, rtol=1e-5)


def test_edge_shapes_single_image_single_sample_odd_counts(golden_w8):
    """Edge cases of the fused path: one image, one sample; three images with 101 samples (an odd count above 100, items that do not
    fill the persistent grid evenly); 255 images (one short of the tile-group multiples).  Fused == layer-wise bit for bit, and the
    single-image case against the CPU oracle."""
    import quantised_bayesian_nets_amd as q
    from oracle import oracle as orc
    m = _model(golden_w8)
    gen = torch.Generator().manual_seed(77)
    net = orc.Int8ResNetOracle(golden_w8["state"], 7, 8)
    x1 = torch.randn(1, 3, 32, 32, generator=gen)
    with q.mc_context(1, 5, 0):
        p = m.forward_mc(x1.cuda())
    np.testing.assert_allclose(p[0].cpu().numpy(), net.forward(x1.numpy(), 5, 0), rtol=RTOL, atol=1e-8)
    assert torch.equal(q.mc_predict(m, x1.cuda(), 1, 5), p[0])
    for B, S in ((3, 101), (255, 2), (1, 7)):
        x = torch.randn(B, 3, 32, 32, generator=gen).cuda()
        with q.mc_context(S, 9, 1000):
            m.fuse_blocks = False
            ref = m.forward_mc(x)
            m.fuse_blocks = True
            assert torch.equal(m.forward_mc(x), ref), (B, S)
        mean, var = q.mc_predict(m, x, S, 9, return_var=True)
        assert mean.shape == (B, 10) and bool(torch.isfinite(var).all())


_TWO_RANK_WORKER = r"""
import os, sys, types, numpy as np, torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from conftest import load_golden
import quantised_bayesian_nets_amd as q
torch.cuda.set_device(0)                                   # both ranks share the one GPU of the test box
dist.init_process_group("gloo")                            # RCCL refuses two ranks on one device; the sharding logic is the same
rank = dist.get_rank()
g = load_golden("resnet_bbb_a7w8.npz")
args = types.SimpleNamespace(activation_precision=7, weight_precision=8)
m = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, args).load_reference_state(g["state"])
x = torch.randn(64, 3, 32, 32, generator=torch.Generator().manual_seed(2)).cuda()
S, seed = 7, 3                                             # odd: the ranks evaluate 4 and 3 samples
mean, var = q.mc_predict(m, x, S, seed, return_var=True)
gp_mean = q.GraphedPredictor(m, S)(x, seed)                # the captured-graph form shards the same way
if rank == 0:
    with q.mc_context(S, seed, 0):
        probs = m.forward_mc(x)                            # all 7 samples on this rank
    np.testing.assert_allclose(mean.cpu().numpy(), probs.double().mean(0).cpu().numpy(), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(var.cpu().numpy(), probs.double().var(0).cpu().numpy(), rtol=1e-5, atol=1e-12)
    assert torch.equal(gp_mean, mean)
    print("TWO-RANK-OK")
dist.barrier()
dist.destroy_process_group()
"""


def test_two_ranks_sharing_the_gpu_equal_one_rank(tmp_path):
    """The N > 1 path on the real kernels: two ranks (gloo, both on cuda:0) shard 7 samples 4 + 3 by GLOBAL sample index, sum their fp64
    moments and finalise -- equal to one rank evaluating all 7 (1e-6: the fp64 sums are added in another order)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "two_rank_worker.py"
    script.write_text(_TWO_RANK_WORKER)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29653", str(script), root]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "TWO-RANK-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_two_ranks_measure_both_multi_gpu_configs():
    """`bench.py --gpus 2` as the driver launches it (self-launch -> torch.distributed.run), rehearsed on the one GPU of this box (both ranks
    on cuda:0, gloo in place of RCCL): the ONE line of the N > 1 run carries BASELINE configs[4] (A7/W4, 1024 global samples, 512 per rank)
    and configs[3] (16 members, 8 per rank) in `secondary`, beside the weak-scaling headline."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(QBNN_BENCH_SHARE_GPU="1", QBNN_BENCH_BACKEND="gloo", OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--prime", "2", "--samples", "20"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["config"]["global_samples"] == 40 and out["rccl_ranks"] == 0          # gloo: no RCCL claim
    sec = out["secondary"]
    w4, ens = sec["resnet_bbb_w4"], sec["ensemble16"]
    assert "error" not in w4 and "error" not in ens, sec
    assert w4["units_per_step_global"] == 1024 and w4["shards"] == [[0, 512], [512, 512]] and w4["value"] > 0 and len(w4["ms_per_step_by_rank"]) == 2
    assert ens["units_per_step_global"] == 16 and ens["shards"] == [[0, 8], [8, 8]] and ens["value"] > 0
    assert w4["scaling"] == "strong" and ens["scaling"] == "strong" and out["scaling"] == "weak"


def test_resnet_mc_fused_post_ops_equal_separate_launches():
    """`conv_resnet_mc`: dropout (+ Add + ReLU) in the convs' store passes (qbnn_conv2d_i8_post_mc) against one launch per op --
    bit-identical with Philox masks and with injected masks, on a batch that leaves ragged image groups in every layer."""
    import quantised_bayesian_nets_amd as q
    from quantised_bayesian_nets_amd import models_mc
    from conftest import load_golden
    g = load_golden("resnet_mc_a7w8.npz")
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, p=g["meta"]["p"])
    m = q.ModelFactory.get_model("conv_resnet_mc", [1, 3, 32, 32], 10, True, args).load_reference_state(g["state"])
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(7, 3, 32, 32, generator=gen).cuda()
    S = 3
    widths = [24] + [24] * 4 + [48] * 5 + [96] * 5 + [192] * 5          # dropouts in draw order (a down block has three)
    assert len(widths) == len(m.dropouts())
    masks = [(torch.rand(S, 7, c, generator=gen) < 0.8).float() for c in widths]
    out = {}
    for fused in (False, True):
        models_mc.BasicBlock.fuse_post = fused
        try:
            rec_p, rec_m = {}, {}
            with q.mc_context(S, 99, 5):
                p_philox = m.forward_mc(x, record=rec_p)
                p_masks = m.forward_mc(x, record=rec_m, masks=masks)
            out[fused] = (p_philox, p_masks, rec_p, rec_m)
        finally:
            models_mc.BasicBlock.fuse_post = True
    for k in out[False][2]:
        assert torch.equal(out[False][2][k], out[True][2][k]), k
        assert torch.equal(out[False][3][k], out[True][3][k]), k
    assert torch.equal(out[False][0], out[True][0]) and torch.equal(out[False][1], out[True][1])
    assert not torch.equal(out[True][0], out[True][1])


def _perturb_activation_qparams(state, seed):
    """Random output scales / zero points for every conv, dropout, Add and the input QuantStub of a converted MC-Dropout state."""
    rng = np.random.default_rng(seed)
    st = dict(state)
    for k in list(st):
        v = np.asarray(st[k])
        if k.endswith("weight.q_scale") or k.endswith("weight.q_zero_point") or "multiplier" in k or k.endswith(".p"):
            continue
        if k.endswith("scale") and v.size == 1:
            st[k] = (v.astype(np.float64) * np.exp(rng.uniform(-0.6, 0.6))).astype(v.dtype)
        elif k.endswith("zero_point") and v.size == 1:
            st[k] = np.asarray(rng.integers(0, 31 if "mul_mask" in k else 90)).astype(v.dtype).reshape(v.shape)
    return st


@pytest.mark.parametrize("B,qseed", [(7, None), (70, None), (9, 1), (21, 2)])
def test_resnet_mc_fused_blocks_equal_per_conv_launches(B, qseed):
    """`conv_resnet_mc` on the fused block kernels with dropout (qbnn_stem_chain_drop_i8_mc / qbnn_block_chain_drop_i8_mc /
    qbnn_block_down_drop_i8_mc: both convs, the dropouts, the Add and the ReLU of a BasicBlock in one launch) against one launch per conv:
    every block's output and the probabilities bit-identical, with Philox masks and with injected masks, at a sample offset, on batches
    that leave ragged image groups in every kernel (and, B = 70, several work items per workgroup range); qseed: the same with random
    output scales and zero points of every conv, dropout mask and Add (the per-conv path is what the reference fixtures pin)."""
    import quantised_bayesian_nets_amd as q
    from quantised_bayesian_nets_amd import models_mc, _lib
    from conftest import load_golden
    g = load_golden("resnet_mc_a7w8.npz")
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, p=g["meta"]["p"])
    state = g["state"] if qseed is None else _perturb_activation_qparams(g["state"], qseed)      # qseed: random activation qparams everywhere
    m = q.ModelFactory.get_model("conv_resnet_mc", [1, 3, 32, 32], 10, True, args).load_reference_state(state)
    gen = torch.Generator().manual_seed(12)
    x = torch.randn(B, 3, 32, 32, generator=gen).cuda()
    S = 3
    widths = [24] + [24] * 4 + [48] * 5 + [96] * 5 + [192] * 5
    masks = [(torch.rand(S, B, c, generator=gen) < 0.8).float() for c in widths]
    for inj in (None, masks):
        rec = {}
        with q.mc_context(S, 99, 5):
            p_ref = m.forward_mc(x, record=rec, masks=inj)              # one launch per conv (recording path)
            assert m._can_fuse_blocks(x, None)
            p_fused = m.forward_mc(x, masks=inj)
            # block by block on the recorded inputs
            mk = list(inj) if inj is not None else None
            xq = torch.empty((1, B, 32, 32, 3), dtype=torch.uint8, device="cuda")
            _lib.check(_lib.lib().qbnn_quantize_input_nchw(_lib.ptr(x), B, 3, 32, 32, m.quant.scale, m.quant.zero_point, 127, _lib.ptr(xq), _lib.current_stream()))
            col = torch.empty((B, 1024, 32), dtype=torch.int8, device="cuda")
            _lib.check(_lib.lib().qbnn_im2col3x3_c3(_lib.ptr(xq), B, 32, 32, m.quant.zero_point, _lib.ptr(col), _lib.current_stream()))
            h = models_mc.run_identity_chain_drop(list(m.layers[4]), None, mk, stem=(m.layers[0], m.layers[3], col, m.quant.scale))
            assert torch.equal(h.data, rec["layers.4.1.out"]), "stem + layer 1"
            for li in (5, 6, 7):
                prev = h
                h = models_mc.run_down_block_drop(m.layers[li][0], prev, mk)
                assert torch.equal(h.data, rec[f"layers.{li}.0.out"]), f"down block {li}"
                h = models_mc.run_identity_chain_drop([m.layers[li][1]], h, mk)
                assert torch.equal(h.data, rec[f"layers.{li}.1.out"]), f"identity block {li}"
            assert mk is None or len(mk) == 0
        assert torch.equal(p_ref, p_fused)


def _small_layer_setup(gen, S, B, shape, shared):
    x = torch.randint(0, 128, ((1 if shared else S), B) + shape, generator=gen, dtype=torch.int32).to(torch.uint8)
    return x


def test_linear_i8_gemm_against_generic_kernels():
    """qbnn_linear_i8_mc (LDS-tiled int8 GEMM, optional per-element dropout in the epilogue) through the C ABI against
    qbnn_conv2d_i8_generic_mc (1x1) -> qbnn_dropout_q_mc on ragged shapes: K and N that are no tile multiples, fewer rows than a tile,
    more than one row tile, sample-shared input, no bias, pitch-padded and dense output rows.  Bit-exact."""
    import ctypes as C
    from quantised_bayesian_nets_amd import _lib
    L = _lib.lib()
    gen = torch.Generator().manual_seed(21)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (S, B, K, N, shared, bias, relu, drop, dense, wps) in [(3, 5, 100, 10, False, True, False, False, True, False), (2, 130, 2450, 500, False, True, True, True, False, False),
                                                                 (2, 64, 37, 64, True, False, True, False, False, False), (1, 257, 500, 100, False, True, False, True, False, False),
                                                                 (2, 16, 64, 33, False, True, True, True, True, False), (3, 40, 100, 100, True, True, True, False, False, True)]:
        ldx = (K + 15) // 16 * 16
        xr = torch.randint(0, 128, ((1 if shared else S), B, ldx), generator=gen, dtype=torch.int32).to(torch.uint8)
        w = torch.randint(-128, 128, ((S if wps else 1), N, K), generator=gen, dtype=torch.int32).to(torch.int8)
        b = (torch.randn(N, generator=gen) * 3).float().cuda() if bias else None
        d = _lib.ConvDesc()
        d.B, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.pad = B, 1, 1, K, N, 1, 1, 0
        d.s_x, d.z_x, d.s_w, d.z_w, d.s_y, d.z_y = 0.05, 17, 0.01, -3, 0.2 * (K / 100.0) ** 0.5, 60
        d.relu, d.a_hi, d.has_bias = int(relu), 127, int(bias)
        nb = L.qbnn_packed_weight_bytes(N, K, K, 0)
        wp = np.zeros((w.shape[0], nb), np.int8)            # per-sample weights (Bayes-by-backprop: sample stride nb) or one fixed weight
        for si in range(w.shape[0]):
            wn = np.ascontiguousarray(w[si].numpy())
            _lib.check(L.qbnn_pack_weights_host(wn.ctypes.data_as(C.c_void_p), N, K, K, 0, wp[si].ctypes.data_as(C.c_void_p)))
        wp = torch.from_numpy(wp).cuda()
        xg = xr.cuda()
        ldy = N if dense else (N + 15) // 16 * 16
        y = torch.full((S, B, ldy), 0xAB, dtype=torch.uint8, device="cuda")
        dd = _lib.DropoutDesc(0.75, 1.0 / 255.0 * 2, 3, 5) if drop else None
        _lib.check(L.qbnn_linear_i8_mc(_lib.ptr(xg), 0 if shared else B * ldx, ldx, _lib.ptr(wp), nb if wps else 0, _lib.ptr(b), _lib.ptr(y), B * ldy, ldy, S, C.byref(d),
                                       None if dd is None else C.byref(dd), None, 77, 4, st))
        # reference: the any-geometry kernel on dense rows, then the stand-alone dropout
        xd = xg[:, :, :K].contiguous()
        wd = w.cuda().contiguous()
        yr = torch.empty((S, B, N), dtype=torch.uint8, device="cuda")
        _lib.check(L.qbnn_conv2d_i8_generic_mc(_lib.ptr(xd), 0 if shared else B * K, _lib.ptr(wd), N * K if wps else 0, _lib.ptr(b), _lib.ptr(yr), B * N, S, C.byref(d), st))
        if drop:
            yd = torch.empty_like(yr)
            _lib.check(L.qbnn_dropout_q_mc(_lib.ptr(yr), B * N, B, 1, N, 0.75, d.s_y, d.z_y, dd.s_m, dd.z_m, 127, 77, 5, 4, None, _lib.ptr(yd), B * N, S, st))
            yr = yd
        torch.cuda.synchronize()
        assert torch.equal(y[:, :, :N], yr), (S, B, K, N)
        assert bool((y[:, :, N:] == 0).all())


def test_conv_pool_drop_small_map_against_generic_kernels():
    """qbnn_conv_pool_drop_i8_mc (LeNet's 20 -> 50 5x5 conv on the 14 x 14 map with the dropout in front, the max-pool, the dropout
    behind and the flatten fused) against the chain of any-geometry kernels, with a batch that leaves a ragged image group; every
    combination of the optional stages.  Bit-exact."""
    import ctypes as C
    from quantised_bayesian_nets_amd import _lib
    L = _lib.lib()
    gen = torch.Generator().manual_seed(22)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    S, B = 3, 6
    w = torch.randint(-128, 128, (50, 5, 5, 20), generator=gen, dtype=torch.int32).to(torch.int8)
    bias = (torch.randn(50, generator=gen) * 2).float().cuda()
    nb = L.qbnn_packed_weight_bytes(50, 500, 100, 0)
    wp = np.zeros(nb, np.int8)
    wn = np.ascontiguousarray(w.numpy().reshape(50, 500))
    _lib.check(L.qbnn_pack_weights_host(wn.ctypes.data_as(C.c_void_p), 50, 500, 100, 0, wp.ctypes.data_as(C.c_void_p)))
    wp, wd = torch.from_numpy(wp).cuda(), w.cuda().contiguous()
    # the last case: per-sample weights (a Bayes-by-backprop conv: sample stride = the packed size), 3 different weights
    w3 = torch.randint(-128, 128, (S, 50, 5, 5, 20), generator=gen, dtype=torch.int32).to(torch.int8)
    wp3 = np.zeros((S, nb), np.int8)
    for si in range(S):
        wn3 = np.ascontiguousarray(w3[si].numpy().reshape(50, 500))
        _lib.check(L.qbnn_pack_weights_host(wn3.ctypes.data_as(C.c_void_p), 50, 500, 100, 0, wp3[si].ctypes.data_as(C.c_void_p)))
    wp3, wd3 = torch.from_numpy(wp3).cuda(), w3.cuda().contiguous()
    for (pool, drop, din, shared) in [(1, 1, 1, True), (0, 0, 0, False), (1, 0, 0, False), (0, 1, 0, False), (1, 1, 0, False), (0, 0, 1, True), (1, 0, 0, "wps")]:
        wps = shared == "wps"
        shared = False if wps else shared
        x = torch.randint(0, 128, ((1 if shared else S), B, 14, 14, 20), generator=gen, dtype=torch.int32).to(torch.uint8).cuda()
        s_in, z_in = 0.04, 23
        d_in = _lib.DropoutDesc(0.8, 0.0039, 2, 0)
        d_out = _lib.DropoutDesc(0.7, 0.0041, 1, 1)
        # reference chain
        xin, sx, zx, xss = x, s_in, z_in, (0 if shared else B * 3920)
        if din:
            xd = torch.empty((S, B, 14, 14, 20), dtype=torch.uint8, device="cuda")
            _lib.check(L.qbnn_dropout_q_mc(_lib.ptr(x), xss, B, 196, 20, 0.8, s_in, z_in, d_in.s_m, d_in.z_m, 127, 9, 0, 2, None, _lib.ptr(xd), B * 3920, S, st))
            xin, sx, zx, xss = xd, d_in.s_m * 1.25, d_in.z_m, B * 3920
        d = _lib.ConvDesc()
        d.B, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.pad = B, 14, 14, 20, 50, 5, 1, 2
        d.s_x, d.z_x, d.s_w, d.z_w, d.s_y, d.z_y = sx, zx, 0.01, 4, sx * 4.0, 55
        d.relu, d.a_hi, d.has_bias = 0, 127, 1
        Sx = S if xss else 1
        yc = torch.empty((Sx, B, 14, 14, 50), dtype=torch.uint8, device="cuda")
        _lib.check(L.qbnn_conv2d_i8_generic_mc(_lib.ptr(xin), xss, _lib.ptr(wd3 if wps else wd), 25000 if wps else 0, _lib.ptr(bias), _lib.ptr(yc), B * 9800, Sx,
                                               C.byref(d), st))
        ref, ho = yc, 14
        if pool:
            yp = torch.empty((Sx, B, 7, 7, 50), dtype=torch.uint8, device="cuda")
            _lib.check(L.qbnn_maxpool2_q_mc(_lib.ptr(ref), B * 9800, B, 14, 14, 50, 127, _lib.ptr(yp), B * 2450, Sx, st))
            ref, ho = yp, 7
        if drop:
            yd = torch.empty((S, B, ho, ho, 50), dtype=torch.uint8, device="cuda")
            _lib.check(L.qbnn_dropout_q_mc(_lib.ptr(ref), (B * ho * ho * 50) if Sx > 1 else 0, B, ho * ho, 50, 0.7, d.s_y, d.z_y, d_out.s_m, d_out.z_m, 127, 9, 1, 2,
                                           None, _lib.ptr(yd), B * ho * ho * 50, S, st))
            ref = yd
        width = ho * ho * 50
        ld = (width + 15) // 16 * 16
        Sy = S if (drop or din or not shared) else 1
        y = torch.full((Sy, B, ld), 0xCD, dtype=torch.uint8, device="cuda")
        _lib.check(L.qbnn_conv_pool_drop_i8_mc(_lib.ptr(x), 0 if shared else B * 3920, _lib.ptr(wp3 if wps else wp), nb if wps else 0, _lib.ptr(bias), _lib.ptr(y), B * ld, ld, Sy,
                                               C.byref(d), pool,
                                               C.byref(d_out) if drop else None, None, C.byref(d_in) if din else None, None, s_in, z_in, 9, 2, st))
        torch.cuda.synchronize()
        assert torch.equal(y[:, :, :width], ref.reshape(ref.shape[0], B, width)), (pool, drop, din)
        assert bool((y[:, :, width:] == 0).all())


def test_conv_c1_pool_against_generic_kernels():
    """qbnn_im2col5x5_c1 + qbnn_conv_c1_pool_i8_mc (LeNet's first conv: one input channel, per-sample weights, max-pool in the wave) through
    the C ABI against qbnn_conv2d_i8_generic_mc -> qbnn_maxpool2_q_mc: random quantisation parameters (negative weight zero point included),
    with and without bias, a batch of 5, the input shared by the samples.  Bit-exact."""
    import ctypes as C
    from quantised_bayesian_nets_amd import _lib
    L = _lib.lib()
    gen = torch.Generator().manual_seed(31)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    S, B = 4, 5
    nb = L.qbnn_packed_weight_bytes(20, 25, 25, 0)
    assert nb == 1024
    for case, (z_x, z_w, z_y, s_y, has_bias) in enumerate([(0, 0, 64, 0.9, 0), (33, -7, 5, 2.5, 1), (127, 11, 120, 0.4, 1)]):
        w = torch.randint(-128, 128, (S, 20, 5, 5, 1), generator=gen, dtype=torch.int32).to(torch.int8)
        wp = np.zeros((S, nb), np.int8)
        for si in range(S):
            wn = np.ascontiguousarray(w[si].numpy().reshape(20, 25))
            _lib.check(L.qbnn_pack_weights_host(wn.ctypes.data_as(C.c_void_p), 20, 25, 25, 0, wp[si].ctypes.data_as(C.c_void_p)))
        wp, wd = torch.from_numpy(wp).cuda(), w.cuda().contiguous()
        bias = (torch.randn(20, generator=gen) * 3).float().cuda()
        x = torch.randint(0, 128, (1, B, 28, 28, 1), generator=gen, dtype=torch.int32).to(torch.uint8).cuda()
        d = _lib.ConvDesc()
        d.B, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.pad = B, 28, 28, 1, 20, 5, 1, 2
        d.s_x, d.z_x, d.s_w, d.z_w, d.s_y, d.z_y = 0.03, z_x, 0.011, z_w, s_y, z_y
        d.relu, d.a_hi, d.has_bias = 0, 127, has_bias
        yc = torch.empty((S, B, 28, 28, 20), dtype=torch.uint8, device="cuda")
        _lib.check(L.qbnn_conv2d_i8_generic_mc(_lib.ptr(x), 0, _lib.ptr(wd), 500, _lib.ptr(bias) if has_bias else None, _lib.ptr(yc), B * 15680, S, C.byref(d), st))
        ref = torch.empty((S, B, 14, 14, 20), dtype=torch.uint8, device="cuda")
        _lib.check(L.qbnn_maxpool2_q_mc(_lib.ptr(yc), B * 15680, B, 28, 28, 20, 127, _lib.ptr(ref), B * 3920, S, st))
        col = torch.empty((B, 784, 32), dtype=torch.int8, device="cuda")
        _lib.check(L.qbnn_im2col5x5_c1(_lib.ptr(x), B, 28, 28, z_x, _lib.ptr(col), st))
        y = torch.full((S, B, 14, 14, 20), 0xCD, dtype=torch.uint8, device="cuda")
        _lib.check(L.qbnn_conv_c1_pool_i8_mc(_lib.ptr(col), 0, _lib.ptr(wp), nb, _lib.ptr(bias) if has_bias else None, _lib.ptr(y), B * 3920, S, C.byref(d), st))
        torch.cuda.synchronize()
        assert torch.equal(y, ref), case
        assert len(torch.unique(ref)) > 8, "the case must not saturate"


def test_lenet_mc_full_sample_count_against_oracle(golden_lenet_mc):
    """BASELINE config 1 at its full MC size: 100 samples (global sample indices 0..99) of the MC-Dropout LeNet on the fused kernels
    (qbnn_conv_pool_drop_i8_mc / qbnn_linear_i8_mc), every sample's probabilities against the oracle at batch 16; the captured-graph
    predictor's mean against the oracle's mean."""
    import quantised_bayesian_nets_amd as q
    from oracle import oracle as orc
    g = golden_lenet_mc
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, p=0.2)
    m = q.ModelFactory.get_model("conv_lenet_mc", [1, 1, 28, 28], 10, True, args).load_reference_state(g["state"])
    net = orc.Int8LeNetMCOracle(g["state"], 7)
    S, seed = 100, 1234
    xb = torch.rand(16, 1, 28, 28, generator=torch.Generator().manual_seed(8))
    with q.mc_context(S, seed, 0):
        probs = m.forward_mc(xb.cuda()).cpu().numpy()
    ref = np.stack([net.forward(xb.numpy(), seed, s) for s in range(S)])
    np.testing.assert_allclose(probs, ref, rtol=RTOL, atol=1e-8)
    mean = q.GraphedPredictor(m, S)(xb.cuda(), seed).cpu().numpy()
    np.testing.assert_allclose(mean, ref.astype(np.float64).mean(0), rtol=RTOL, atol=1e-8)


def _pack_per_sample(L, w, layout=0):
    """int8 [S, Cout, KH, KW, Cin] -> QBNN_LAYOUT_MFMA32 (0) / _MFMA32_N24 (2) fragments [S, nbytes] on the device (krow as layers.Conv2d chooses it)."""
    import ctypes as C
    from quantised_bayesian_nets_amd import _lib
    S, cout, kh, kw, cin = w.shape
    k = kh * kw * cin
    krow = kw * cin if cin % 8 == 0 else k
    nb = L.qbnn_packed_weight_bytes(cout, k, krow, layout)
    out = np.zeros((S, nb), np.int8)
    for s in range(S):
        src = np.ascontiguousarray(w[s].reshape(cout, k))
        _lib.check(L.qbnn_pack_weights_host(src.ctypes.data_as(C.c_void_p), cout, k, krow, layout, out[s].ctypes.data_as(C.c_void_p)))
    return torch.from_numpy(out).cuda(), nb


@pytest.mark.parametrize("seed,extreme", [(s, False) for s in range(int(os.environ.get("QBNN_TEST_SEEDS", "6")))] + [(0, True), (1, True), (2, True)])
def test_fused_blocks_random_qparams_against_oracle(seed, extreme):
    """The fused BasicBlock kernels through the C ABI with RANDOM quantisation parameters (the fixtures only carry the calibrated ones):
    zero points over their whole range incl. negative weight zero points, scales over two decades, 7- / 6- / 5-bit activations, with and
    without bias, per-sample weights, ragged batches -- identity blocks at 24 / 48 / 96 / 192 channels and the three down-sampling
    blocks, each against the oracle's conv -> conv -> quantized::add chain.  Bit-exact.
    `extreme` (round 5): the 24-channel cases at the ACCUMULATOR BOUND of the magic start (accumulators begin at 1.5 * 2^23, exact while
    |sum| < 2^22): 7-bit activations at 127 with zero point 0 against weights of +127 / -128 in per-channel proportions 0 .. 1 -- sums from
    -3.5 M to +3.5 M of the 4.19 M the trick allows (K = 216).  (QBNN_TEST_SEEDS=n runs n seeds instead of six.)"""
    import ctypes as C
    from oracle import oracle as orc
    from quantised_bayesian_nets_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(100 + seed)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    S, B = 2, 3

    def qp(use_bias, a_hi):
        """(s_w, z_w, output zero point): without a bias the weight zero point stays small (nothing would take the mean out)"""
        s_w = float(np.float32(10 ** rng.uniform(-3, -1.5)))
        z_w = int(rng.integers(-25, 26)) if use_bias else int(rng.integers(-2, 3))
        return s_w, z_w, int(rng.integers(a_hi // 4, 3 * a_hi // 4 + 1))

    def conv_ref(x, w, b, stride, pad, s_x, z_x, s_w, z_w, s_y, z_y, relu, a_hi):
        return np.stack([orc.conv2d_i8(x[s], w[s], b, stride, pad, s_x, z_x, s_w, z_w, s_y, z_y, relu, a_hi) for s in range(x.shape[0])])

    def out_scale(x, w, stride, pad, s_x, z_x, s_w, z_w, a_hi, use_bias):
        """(output scale, bias): the scale spreads this conv's real-valued outputs over about half of the activation range; the bias (when
        the case has one) takes out the mean that far-off zero points put on the accumulator, plus noise per channel"""
        xf = torch.from_numpy(x[0].astype(np.float64) - z_x).permute(0, 3, 1, 2)
        wf = torch.from_numpy(w[0].astype(np.float64) - z_w).permute(0, 3, 1, 2)
        acc = torch.nn.functional.conv2d(xf, wf, stride=stride, padding=pad)
        sd = float((acc - acc.mean(dim=(0, 2, 3), keepdim=True)).std()) if use_bias else float(acc.std())
        s_y = float(np.float32(s_x * s_w * sd * 4.0 / a_hi * rng.uniform(0.7, 1.5)))
        bias = None
        if use_bias:
            bias = (-acc.mean(dim=(0, 2, 3)).numpy() * s_x * s_w + rng.normal(size=w.shape[1]) * s_y * a_hi / 8).astype(np.float32)
        return s_y, bias

    for Cc, H in ((24, 32), (48, 16), (96, 8), (192, 4)):
        for down in (False, True):
            if down and Cc == 192:
                continue
            a_hi = int(rng.choice([127, 127, 63, 31, 15, 7]))          # src/utils.py:18: UINT_BOUNDS of A7 ... A3
            ext = extreme and Cc == 24
            if extreme and not ext:
                continue
            if ext:
                a_hi = 127
            Ci, Co, Hi = Cc, (2 * Cc if down else Cc), H
            Ho = Hi // 2 if down else Hi
            use_bias = bool(rng.integers(0, 2))
            s_x = float(np.float32(10 ** rng.uniform(-2, -1)))
            z_x = int(rng.integers(0, a_hi + 1)) if use_bias else int(np.clip(a_hi // 2 + rng.integers(-4, 5), 0, a_hi))
            x = rng.integers(0, a_hi + 1, (S, B, Hi, Hi, Ci), dtype=np.uint8)
            wa = rng.integers(-128, 128, (S, Co, 3, 3, Ci), dtype=np.int8)
            wb = rng.integers(-128, 128, (S, Co, 3, 3, Co), dtype=np.int8)
            if ext:
                z_x = 0
                x = np.where(rng.random(x.shape) < 0.9, a_hi, x).astype(np.uint8)
                frac = rng.choice([0.0, 0.25, 0.5, 0.75, 1.0], (S, Co, 1, 1, 1))
                wa = np.where(rng.random(wa.shape) < frac, 127, -128).astype(np.int8)
            s_wa, z_wa, z_a = qp(use_bias, a_hi)
            s_wb, z_wb, z_b = qp(use_bias, a_hi)
            if not use_bias:
                z_a = int(a_hi // 8)                   # ReLU output: zero point low, and conv b's input mean near it
            # ---- oracle
            stride = 2 if down else 1
            s_a, ba = out_scale(x, wa, stride, 1, s_x, z_x, s_wa, z_wa, a_hi, use_bias)
            t = conv_ref(x, wa, ba, stride, 1, s_x, z_x, s_wa, z_wa, s_a, z_a, True, a_hi)
            s_b, bb = out_scale(t, wb, 1, 1, s_a, z_a, s_wb, z_wb, a_hi, use_bias)
            u = conv_ref(t, wb, bb, 1, 1, s_a, z_a, s_wb, z_wb, s_b, z_b, False, a_hi)
            z_o = int(rng.integers(0, a_hi // 2 + 1))
            if down:
                ws = rng.integers(-128, 128, (S, Co, 1, 1, Ci), dtype=np.int8)
                s_ws, z_ws, z_s = qp(use_bias, a_hi)
                s_s, bs = out_scale(x, ws, 2, 0, s_x, z_x, s_ws, z_ws, a_hi, use_bias)
                sc = conv_ref(x, ws, bs, 2, 0, s_x, z_x, s_ws, z_ws, s_s, z_s, False, a_hi)
                other, s_r, z_r = sc, s_s, z_s
            else:
                other, s_r, z_r = x, s_x, z_x
            real = (u.astype(np.float64) - z_b) * s_b + (other.astype(np.float64) - z_r) * s_r
            s_o = float(np.float32(real.std() * 6.0 / a_hi * rng.uniform(0.7, 1.5)))       # the Add's output scale: its real values over the range
            ref = orc.qadd_relu(u, s_b, z_b, other, s_r, z_r, s_o, z_o, True, a_hi)
            lv = min(8, a_hi // 2)                  # (A3 has 8 levels in all)
            assert len(np.unique(t)) > lv and len(np.unique(u)) > lv and len(np.unique(ref)) > min(3, a_hi // 4), "degenerate case: outputs saturated"
            # ---- fused kernel
            wa_d, nba = _pack_per_sample(L, wa)
            wb_d, nbb = _pack_per_sample(L, wb)
            dev = lambda v: None if v is None else torch.from_numpy(v).cuda()
            ba_d, bb_d = dev(ba), dev(bb)
            blk = _lib.BlockDesc()
            blk.w_a, blk.w_a_sample_stride, blk.bias_a = wa_d.data_ptr(), nba, (ba_d.data_ptr() if use_bias else None)
            blk.s_wa, blk.z_wa, blk.s_a, blk.z_a = s_wa, z_wa, s_a, z_a
            blk.w_b, blk.w_b_sample_stride, blk.bias_b = wb_d.data_ptr(), nbb, (bb_d.data_ptr() if use_bias else None)
            blk.s_wb, blk.z_wb, blk.s_b, blk.z_b, blk.s_o, blk.z_o = s_wb, z_wb, s_b, z_b, s_o, z_o
            xd = torch.from_numpy(x).cuda()
            y = torch.full((S, B, Ho, Ho, Co), 0xEE, dtype=torch.uint8, device="cuda")
            if down:
                ws_d, nbs = _pack_per_sample(L, ws)
                bs_d = dev(bs)
                dd = _lib.DownDesc()
                dd.blk = blk
                dd.w_s, dd.w_s_sample_stride, dd.bias_s = ws_d.data_ptr(), nbs, (bs_d.data_ptr() if use_bias else None)
                dd.s_ws, dd.z_ws, dd.s_s, dd.z_s = s_ws, z_ws, s_s, z_s
                _lib.check(L.qbnn_block_down_i8_mc(_lib.ptr(xd), xd[0].numel(), s_x, z_x, B, Hi, Ci, a_hi, C.byref(dd), _lib.ptr(y), y[0].numel(), S, st))
            else:
                _lib.check(L.qbnn_block_chain_i8_mc(_lib.ptr(xd), xd[0].numel(), s_x, z_x, B, Hi, Ci, a_hi, C.byref(blk), 1, _lib.ptr(y), y[0].numel(), S, st))
            torch.cuda.synchronize()
            got = y.cpu().numpy()
            assert np.array_equal(got, ref), (Cc, down, a_hi, int((got != ref).sum()))
            if Cc == 24 and down:
                # round 5: the 24 -> 48 block on the 16-wave kernel (csrc/qbnn_c48.hip): stem.0 as MFMA32_N24_TAIL, stem.3 and the shortcut as MFMA32_N24
                wa2, nba2 = _pack_per_sample(L, wa, 4)
                wb2, nbb2 = _pack_per_sample(L, wb, 2)
                ws2, nbs2 = _pack_per_sample(L, ws, 2)
                dd.blk.w_a, dd.blk.w_a_sample_stride, dd.blk.w_b, dd.blk.w_b_sample_stride, dd.blk.w_layout = wa2.data_ptr(), nba2, wb2.data_ptr(), nbb2, 2
                dd.w_s, dd.w_s_sample_stride = ws2.data_ptr(), nbs2
                y2 = torch.full((S, B, Ho, Ho, Co), 0xEE, dtype=torch.uint8, device="cuda")
                _lib.check(L.qbnn_block_down_i8_mc(_lib.ptr(xd), xd[0].numel(), s_x, z_x, B, Hi, Ci, a_hi, C.byref(dd), _lib.ptr(y2), y2[0].numel(), S, st))
                torch.cuda.synchronize()
                assert np.array_equal(y2.cpu().numpy(), ref), ("down 24 -> 48, N24 set", a_hi, int((y2.cpu().numpy() != ref).sum()))
            if Cc == 192 and not down:
                # round 5: QBNN_BLOCK_POOL_OUT -- the block's output leaves as its AvgPool2d(4) (what the head consumes), [S][B][192]
                blk.flags = 1
                yp = torch.full((S, B, Co), 0xEE, dtype=torch.uint8, device="cuda")
                _lib.check(L.qbnn_block_chain_i8_mc(_lib.ptr(xd), xd[0].numel(), s_x, z_x, B, Hi, Ci, a_hi, C.byref(blk), 1, _lib.ptr(yp), yp[0].numel(), S, st))
                torch.cuda.synchronize()
                want = np.stack([orc.avgpool_q(ref[s], 4, z_o, a_hi).reshape(B, Co) for s in range(S)])
                assert np.array_equal(yp.cpu().numpy(), want), ("pooled output", int((yp.cpu().numpy() != want).sum()))
                blk.flags = 0
            if Cc == 96 and not down:
                blk.flags = 1                          # ... and any other geometry refuses the flag
                rc = L.qbnn_block_chain_i8_mc(_lib.ptr(xd), xd[0].numel(), s_x, z_x, B, Hi, Ci, a_hi, C.byref(blk), 1, _lib.ptr(y), y[0].numel(), S, st)
                assert rc != 0 and b"POOL_OUT" in L.qbnn_last_error()
                blk.flags = 0
            if Cc == 48 and not down:
                # round 5: the same block on the 16-wave kernel (csrc/qbnn_c48.hip) -- weights as (24 + 1)-row tile halves (MFMA32_N24)
                wa2, nba2 = _pack_per_sample(L, wa, 2)
                wb2, nbb2 = _pack_per_sample(L, wb, 2)
                blk.w_a, blk.w_a_sample_stride, blk.w_b, blk.w_b_sample_stride, blk.w_layout = wa2.data_ptr(), nba2, wb2.data_ptr(), nbb2, 2
                y2 = torch.full((S, B, Ho, Ho, Co), 0xEE, dtype=torch.uint8, device="cuda")
                _lib.check(L.qbnn_block_chain_i8_mc(_lib.ptr(xd), xd[0].numel(), s_x, z_x, B, Hi, Ci, a_hi, C.byref(blk), 1, _lib.ptr(y2), y2[0].numel(), S, st))
                torch.cuda.synchronize()
                assert np.array_equal(y2.cpu().numpy(), ref), ("N24", a_hi, int((y2.cpu().numpy() != ref).sum()))


@pytest.mark.parametrize("seed", list(range(max(4, int(os.environ.get("QBNN_TEST_SEEDS", "4"))))))
def test_sampler_random_qparams_against_oracle(seed):
    """The weight sampler (qbnn_sample_weights_i8_multi: Philox -> eps_q -> quantized::mul -> quantized::add -> clamp_weight, computed
    in fp32 on exact small integers) with RANDOM quantisation parameters against the oracle's integer / ATen-formula chain: zero points of
    sigma, the product and the sum over +-60, scales over two decades, 8- and 4-bit weights, the fragment layout's fast path (Cin = 48,
    96) and its general path (Cin = 3: whole-K rows), several samples starting at a non-zero global index.  Bit-exact."""
    import quantised_bayesian_nets_amd as q
    from quantised_bayesian_nets_amd.layers import Conv2d as QConv
    from oracle import oracle as orc
    rng = np.random.default_rng(500 + seed)
    for (cin, cout, k) in ((48, 48, 3), (96, 192, 3), (3, 24, 3), (24, 48, 1)):
        w_bits = int(rng.choice([8, 4, 3, 5, 6, 7]))                   # the reference's sweep: experiments/run_all_quant.sh:11-24
        args = types.SimpleNamespace(activation_precision=7, weight_precision=w_bits)
        layer = QConv(cin, cout, (k, k), stride=1, padding=k // 2, bias=False, args=args)
        layer.layer_id = int(rng.integers(0, 21))
        mu = rng.integers(-128, 128, (cout, cin, k, k), dtype=np.int8)                 # OIHW, as the reference's state dict holds it
        sg = rng.integers(-128, 128, (cout, cin, k, k), dtype=np.int8)
        s_w, z_w = float(np.float32(10 ** rng.uniform(-3, -1.5))), int(rng.integers(-60, 61))
        s_sg, z_sg = float(np.float32(10 ** rng.uniform(-4, -2))), int(rng.integers(-128, -60))     # softplus(rho) > 0: the reference's sigma sits above its zero point
        s_mul, z_mul = float(np.float32(s_sg * 128 * 3.0 * rng.uniform(0.5, 2) / 127)), int(rng.integers(-60, 61))
        s_add, z_add = float(np.float32(s_w * rng.uniform(0.8, 1.6))), int(rng.integers(-60, 61))
        st = {"weight": mu, "weight.q_scale": s_w, "weight.q_zero_point": z_w, "std": sg, "std.q_scale": s_sg, "std.q_zero_point": z_sg,
              "scale": 0.1, "zero_point": 3, "add_weight.scale": s_add, "add_weight.zero_point": z_add, "mul_noise.scale": s_mul,
              "mul_noise.zero_point": z_mul}
        layer.load_reference_state(st, "")
        p = orc.sample_params(s_w, z_w, s_sg, z_sg, s_mul, z_mul, s_add, z_add, w_bits)
        S, sb, sd = 5, 254, 77 + seed                                                     # 5 samples: one full group of 4 and a ragged one
        w = layer.sample_weights("cuda", samples=S, seed=sd, sample_begin=sb).cpu().numpy()
        mu_l, sg_l = orc.oihw_to_ohwi(mu), orc.oihw_to_ohwi(sg)
        seen = set()
        for s in range(S):
            ref = orc.sample_weights_i8_philox(mu_l, sg_l, p, sd, layer.layer_id, sb + s)
            seen.update(np.unique(ref).tolist())
            assert np.array_equal(w[s], _pack(layer, ref)), (cin, cout, k, w_bits, s)
        assert len(seen) > min(40, 2 ** w_bits // 2), "degenerate case: the sampled weights barely vary"


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_conv_post_ops_random_qparams_against_separate_kernels(seed):
    """qbnn_conv2d_i8_post_mc (dropout, and dropout + Add + ReLU, in the conv epilogue) with RANDOM quantisation parameters against
    qbnn_conv2d_i8_mc -> qbnn_dropout_q_mc (-> qbnn_add_relu_q_mc), which the other tests tie to the oracle and the reference: mask zero
    points 0..127, conv / residual / sum zero points and scales at random, keep probabilities 0.5..0.95, ragged batch.  Bit-exact."""
    import ctypes as C
    from quantised_bayesian_nets_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(900 + seed)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    S, B = 3, 5
    for (H, Ci, Co, k, stride) in ((32, 24, 24, 3, 1), (16, 48, 96, 3, 2), (8, 96, 96, 3, 1), (16, 48, 96, 1, 2), (4, 192, 192, 3, 1)):
        Ho = H // stride
        x = torch.from_numpy(rng.integers(0, 128, (S, B, H, H, Ci), dtype=np.uint8)).cuda()
        w = rng.integers(-128, 128, (1, Co, k, k, Ci), dtype=np.int8)
        wp, nb = _pack_per_sample(L, w)
        bias = torch.from_numpy((rng.normal(size=Co) * 3).astype(np.float32)).cuda()
        d = _lib.ConvDesc()
        d.B, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.pad = B, H, H, Ci, Co, k, stride, k // 2
        d.s_x, d.z_x = float(np.float32(10 ** rng.uniform(-2, -1))), int(rng.integers(0, 128))
        # a {0, 1} mask quantises to about 1 / s_m; the dropout's output lives on the mask's scale, so the conv's output scale has to be of
        # that order for anything but saturation to come out (as in the reference's calibrated models); the weight scale follows from it
        keep, s_m, z_m, lid = float(np.float32(rng.uniform(0.5, 0.95))), float(np.float32(rng.uniform(1.0, 2.0) / 255.0)), int(rng.integers(0, 128)), int(rng.integers(0, 20))
        d.s_y, d.z_y = float(np.float32(s_m * rng.uniform(0.4, 1.0))), int(rng.integers(20, 100))
        d.z_w = int(rng.integers(-10, 11))
        d.s_w = float(np.float32(d.s_y * 127 / (4 * d.s_x * 74 * 37 * np.sqrt(k * k * Ci))))
        d.relu, d.a_hi, d.has_bias = int(rng.integers(0, 2)), 127, 1
        mult = float(np.float32(1.0) / np.float32(keep))
        other = torch.from_numpy(rng.integers(0, 128, (S, B, Ho, Ho, Co), dtype=np.uint8)).cuda()
        s_b, z_b = float(np.float32(10 ** rng.uniform(-2, -1))), int(rng.integers(0, 128))
        s_a = float(np.float32(s_m * mult))
        s_o, z_o = float(np.float32(max(s_a, s_b) * rng.uniform(1.0, 2.5))), int(rng.integers(0, 64))
        n = B * Ho * Ho * Co
        yc = torch.empty((S, B, Ho, Ho, Co), dtype=torch.uint8, device="cuda")
        yd, ya = torch.empty_like(yc), torch.empty_like(yc)
        _lib.check(L.qbnn_conv2d_i8_mc(_lib.ptr(x), x[0].numel(), _lib.ptr(wp), 0, _lib.ptr(bias), None, 0, _lib.ptr(yc), n, S, C.byref(d), st))
        _lib.check(L.qbnn_dropout_q_mc(_lib.ptr(yc), n, B, Ho * Ho, Co, keep, d.s_y, d.z_y, s_m, z_m, 127, 31 + seed, lid, 7, None, _lib.ptr(yd), n, S, st))
        _lib.check(L.qbnn_add_relu_q_mc(_lib.ptr(yd), n, s_a, z_m, _lib.ptr(other), n, s_b, z_b, _lib.ptr(ya), n, n, s_o, z_o, 127, 1, S, st))
        for add in (0, 1):
            q = _lib.PostDesc(keep, s_m, z_m, lid, add, s_a, s_b, z_b, s_o, z_o)
            y = torch.full_like(yc, 0x5A)
            _lib.check(L.qbnn_conv2d_i8_post_mc(_lib.ptr(x), x[0].numel(), _lib.ptr(wp), 0, _lib.ptr(bias), _lib.ptr(y), n, S, C.byref(d), C.byref(q), None,
                                                _lib.ptr(other) if add else None, n if add else 0, 31 + seed, 7, st))
            torch.cuda.synchronize()
            ref = ya if add else yd
            assert torch.equal(y, ref), (H, Ci, Co, k, add, int((y != ref).sum()))
        assert len(torch.unique(yd)) > 8


@pytest.mark.parametrize("seed", list(range(max(3, int(os.environ.get("QBNN_TEST_SEEDS", "3"))))))
def test_fused_stem_chain_random_qparams_against_oracle(seed):
    """qbnn_stem_chain_i8_mc (layers.0 on the 27-tap patches fused in front of one or two 24-channel identity blocks: the dominant kernel
    of the benchmark) with RANDOM quantisation parameters and per-sample weights against the oracle's conv / conv / conv / add chain.
    Bit-exact."""
    import ctypes as C
    from oracle import oracle as orc
    from quantised_bayesian_nets_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(1300 + seed)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    S, B, a_hi = 2, 3, 127
    dev = lambda v: torch.from_numpy(v).cuda()

    def conv_s(x, w, b, s_x, z_x, s_w, z_w, s_y, z_y, relu):
        return np.stack([orc.conv2d_i8(x[s if x.shape[0] > 1 else 0], w[s], b, 1, 1, s_x, z_x, s_w, z_w, s_y, z_y, relu, a_hi) for s in range(S)])

    def scale_bias(x, w, s_x, z_x, s_w, z_w):
        xf = torch.from_numpy(x[0].astype(np.float64) - z_x).permute(0, 3, 1, 2)
        wf = torch.from_numpy(w[0].astype(np.float64) - z_w).permute(0, 3, 1, 2)
        acc = torch.nn.functional.conv2d(xf, wf, padding=1)
        sd = float((acc - acc.mean(dim=(0, 2, 3), keepdim=True)).std())
        s_y = float(np.float32(s_x * s_w * sd * 4.0 / a_hi * rng.uniform(0.7, 1.5)))
        return s_y, (-acc.mean(dim=(0, 2, 3)).numpy() * s_x * s_w + rng.normal(size=w.shape[1]) * s_y * a_hi / 8).astype(np.float32)

    for n_blocks in (1, 2):
        x = rng.integers(0, a_hi + 1, (1, B, 32, 32, 3), dtype=np.uint8)            # the quantised image, shared by the samples
        s_in, z_in = float(np.float32(10 ** rng.uniform(-2, -1))), int(rng.integers(0, 128))
        w0 = rng.integers(-128, 128, (S, 24, 3, 3, 3), dtype=np.int8)
        s_w0, z_w0, z_y0 = float(np.float32(10 ** rng.uniform(-3, -1.5))), int(rng.integers(-25, 26)), int(rng.integers(10, 60))
        s_y0, b0 = scale_bias(x, w0, s_in, z_in, s_w0, z_w0)
        h = conv_s(x, w0, b0, s_in, z_in, s_w0, z_w0, s_y0, z_y0, True)
        s_h, z_h = s_y0, z_y0
        blks = (_lib.BlockDesc * n_blocks)()
        keep, tails = [], []
        for bi in range(n_blocks):
            wa = rng.integers(-128, 128, (S, 24, 3, 3, 24), dtype=np.int8)
            wb = rng.integers(-128, 128, (S, 24, 3, 3, 24), dtype=np.int8)
            s_wa, z_wa, z_a = float(np.float32(10 ** rng.uniform(-3, -1.5))), int(rng.integers(-25, 26)), int(rng.integers(10, 60))
            s_wb, z_wb, z_b = float(np.float32(10 ** rng.uniform(-3, -1.5))), int(rng.integers(-25, 26)), int(rng.integers(30, 100))
            s_a, ba = scale_bias(h, wa, s_h, z_h, s_wa, z_wa)
            t = conv_s(h, wa, ba, s_h, z_h, s_wa, z_wa, s_a, z_a, True)
            s_b, bb = scale_bias(t, wb, s_a, z_a, s_wb, z_wb)
            u = conv_s(t, wb, bb, s_a, z_a, s_wb, z_wb, s_b, z_b, False)
            real = (u.astype(np.float64) - z_b) * s_b + (h.astype(np.float64) - z_h) * s_h
            s_o, z_o = float(np.float32(real.std() * 6.0 / a_hi * rng.uniform(0.7, 1.5))), int(rng.integers(0, 50))
            h = orc.qadd_relu(u, s_b, z_b, h, s_h, z_h, s_o, z_o, True, a_hi)
            assert len(np.unique(t)) > 8 and len(np.unique(h)) > 8
            wa_d, nba = _pack_per_sample(L, wa)
            wb_d, nbb = _pack_per_sample(L, wb)
            wa_t, nba_t = _pack_per_sample(L, wa, 3)        # QBNN_LAYOUT_MFMA32_TAIL: the 16-wave kernel's operand (two blocks)
            wb_t, nbb_t = _pack_per_sample(L, wb, 3)
            ba_d, bb_d = dev(ba), dev(bb)
            keep += [wa_d, wb_d, ba_d, bb_d, wa_t, wb_t]
            tails.append((wa_t.data_ptr(), nba_t, wb_t.data_ptr(), nbb_t))
            k = blks[bi]
            k.w_a, k.w_a_sample_stride, k.bias_a, k.s_wa, k.z_wa, k.s_a, k.z_a = wa_d.data_ptr(), nba, ba_d.data_ptr(), s_wa, z_wa, s_a, z_a
            k.w_b, k.w_b_sample_stride, k.bias_b, k.s_wb, k.z_wb, k.s_b, k.z_b = wb_d.data_ptr(), nbb, bb_d.data_ptr(), s_wb, z_wb, s_b, z_b
            k.s_o, k.z_o = s_o, z_o
            s_h, z_h = s_o, z_o
        w0_d, nb0 = _pack_per_sample(L, w0)
        b0_d = dev(b0)
        xd = dev(x[0])
        im = torch.empty((B, 1024, 32), dtype=torch.int8, device="cuda")
        _lib.check(L.qbnn_im2col3x3_c3(_lib.ptr(xd), B, 32, 32, z_in, _lib.ptr(im), st))
        y = torch.full((S, B, 32, 32, 24), 0xEE, dtype=torch.uint8, device="cuda")
        _lib.check(L.qbnn_stem_chain_i8_mc(_lib.ptr(im), B, _lib.ptr(w0_d), nb0, _lib.ptr(b0_d), s_in, s_w0, z_w0, s_y0, z_y0, a_hi, blks, n_blocks,
                                           _lib.ptr(y), y[0].numel(), S, st))
        torch.cuda.synchronize()
        got = y.cpu().numpy()
        assert np.array_equal(got, h), (n_blocks, int((got != h).sum()))        # MFMA32 weights: the 8-wave kernel
        if n_blocks == 2:
            # the same call with the blocks' weights as MFMA32_TAIL fragments (7 k-steps): the 16-wave kernel with the magic accumulator start
            for k, (pa, na, pb, nb_) in zip(blks, tails):
                k.w_a, k.w_a_sample_stride, k.w_b, k.w_b_sample_stride, k.w_layout = pa, na, pb, nb_, 3
            y2 = torch.full((S, B, 32, 32, 24), 0xEE, dtype=torch.uint8, device="cuda")
            _lib.check(L.qbnn_stem_chain_i8_mc(_lib.ptr(im), B, _lib.ptr(w0_d), nb0, _lib.ptr(b0_d), s_in, s_w0, z_w0, s_y0, z_y0, a_hi, blks, n_blocks,
                                               _lib.ptr(y2), y2[0].numel(), S, st))
            torch.cuda.synchronize()
            assert np.array_equal(y2.cpu().numpy(), h), ("TAIL", int((y2.cpu().numpy() != h).sum()))


@pytest.mark.parametrize("seed", [0, 1])
def test_head_random_qparams_against_oracle(seed):
    """qbnn_head_i8_mc (AvgPool -> Linear -> DeQuant -> softmax; both its forms: int8-dword dot products when C % 16 == 0, the scalar one
    otherwise) with RANDOM quantisation parameters, per-sample weights, with and without bias against the oracle.  The integers behind the
    probabilities are exact, so the probabilities agree to float rounding of the softmax (1e-6)."""
    import ctypes as C
    from oracle import oracle as orc
    from quantised_bayesian_nets_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(1700 + seed)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    S, B = 3, 7
    for (Cc, k, N, use_bias) in ((192, 4, 10, False), (192, 4, 10, True), (24, 2, 5, True), (100, 1, 16, True)):
        a_hi = int(rng.choice([127, 63]))
        x = rng.integers(0, a_hi + 1, (S, B, k, k, Cc), dtype=np.uint8)
        w = rng.integers(-128, 128, (S, N, Cc), dtype=np.int8)
        bias = (rng.normal(size=N) * 0.5).astype(np.float32) if use_bias else None
        # zero points near the operands' means (nothing else takes the mean off the logits); the output scale from the accumulator's spread
        s_x, z_x = float(np.float32(10 ** rng.uniform(-2, -1))), int(a_hi // 2 + rng.integers(-12, 13))
        s_w, z_w = float(np.float32(10 ** rng.uniform(-3, -2))), int(rng.integers(-4, 5))
        s_y = float(np.float32(s_x * s_w * 74 * (a_hi / 3.46 / k) * np.sqrt(Cc) * 8 / a_hi * rng.uniform(0.7, 1.5)))
        z_y = int(rng.integers(a_hi // 4, 3 * a_hi // 4))
        d = _lib.HeadDesc(B, k, Cc, N, s_x, z_x, s_w, z_w, s_y, z_y, a_hi, int(use_bias))
        xd, wd = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
        bd = None if bias is None else torch.from_numpy(bias).cuda()
        probs = torch.empty((S, B, N), dtype=torch.float32, device="cuda")
        _lib.check(L.qbnn_head_i8_mc(_lib.ptr(xd), xd[0].numel(), _lib.ptr(wd), N * Cc, _lib.ptr(bd), _lib.ptr(probs), S, C.byref(d), st))
        torch.cuda.synchronize()
        for s in range(S):
            pooled = orc.avgpool_q(x[s], k, z_x, a_hi).reshape(B, Cc)
            logits = orc.linear_i8(pooled, w[s], bias, s_x, z_x, s_w, z_w, s_y, z_y, False, a_hi)
            assert len(np.unique(logits)) > 4
            ref = orc.dequant_softmax(logits, s_y, z_y)
            np.testing.assert_allclose(probs[s].cpu().numpy(), ref, rtol=1e-6, atol=1e-9, err_msg=str((Cc, k, N, s)))


@pytest.mark.gpu
def test_checkpoint_file_to_hip_matches_reference(golden_w8):
    """SURVEY 8f row 1 end to end on the device: the file the reference's `utils.save_model` wrote (torch.save of the converted
    qint8 state dict, keys under `module.`, src/utils.py:84-93) -> `checkpoint.load_model` (src/utils.py:112-123) -> the HIP
    path's MC evaluation == the reference's recorded per-sample and mean probabilities."""
    import quantised_bayesian_nets_amd as q
    from quantised_bayesian_nets_amd import checkpoint as ck
    g = golden_w8
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8)
    path = os.path.join(os.path.dirname(__file__), "golden", "resnet_bbb_a7w8_weights.pt")
    m = ck.load_model(q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, args), path)
    x = torch.from_numpy(g["x"]).cuda()
    S, seed = g["probs"].shape[0], g["meta"]["philox_seed"]
    mean, probs = q.mc_predict(m, x, S, seed, return_probs=True)
    np.testing.assert_allclose(probs.cpu().numpy(), g["probs"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(mean.cpu().numpy(), g["mean_probs"], rtol=RTOL, atol=1e-8)
    assert torch.equal(mean, q.mc_predict(_model(g), x, S, seed))          # and bit-identical to the model loaded from the arrays


@pytest.mark.gpu
def test_reloading_a_state_after_a_forward_replaces_every_cached_weight(golden_lenet_mc, golden_w8):
    """A model that has run keeps device copies of its weights (packed MFMA fragments, biases, captured graphs).  Loading a second
    state into the SAME model must drop all of them: state A -> run -> state B -> run == a fresh model loaded with B."""
    import quantised_bayesian_nets_amd as q
    g = golden_lenet_mc
    la = types.SimpleNamespace(activation_precision=7, weight_precision=8, p=0.2)
    rng = np.random.default_rng(11)
    state_b = {}
    for k, v in g["state"].items():
        v = np.asarray(v)
        if k.endswith(".weight") and v.dtype == np.int8:
            state_b[k] = np.clip(v.astype(np.int32) + rng.integers(-9, 10, v.shape), -128, 127).astype(np.int8)
        elif k.endswith(".bias") and v.size:
            state_b[k] = (v * 1.25).astype(np.float32)
        else:
            state_b[k] = v
    x = torch.rand(128, 1, 28, 28, generator=torch.Generator().manual_seed(3)).cuda()
    m = q.ModelFactory.get_model("conv_lenet_mc", [1, 1, 28, 28], 10, True, la).load_reference_state(g["state"])
    gp = q.GraphedPredictor(m, 6)
    pa, ga = q.mc_predict(m, x, 6, 21), gp(x, 21)
    m.load_reference_state(state_b)
    pb, gb = q.mc_predict(m, x, 6, 21), gp(x, 21)
    fresh = q.ModelFactory.get_model("conv_lenet_mc", [1, 1, 28, 28], 10, True, la).load_reference_state(state_b)
    want = q.mc_predict(fresh, x, 6, 21)
    assert torch.equal(pa, ga) and not torch.equal(pa, pb)
    assert torch.equal(pb, want) and torch.equal(gb, want)
    # the int8 BBB ResNet through a captured graph: same check (the graph holds raw pointers to the packed mu / sigma)
    r = _model(golden_w8)
    gr = q.GraphedPredictor(r, 3)
    xr = torch.randn(8, 3, 32, 32, generator=torch.Generator().manual_seed(4)).cuda()
    a0 = gr(xr, 5)
    st2 = dict(golden_w8["state"])
    k0 = "layers.9.weight"
    st2[k0] = np.clip(np.asarray(st2[k0]).astype(np.int32) + 5, -128, 127).astype(np.int8)
    r.load_reference_state(st2)
    a1 = gr(xr, 5)
    r2 = _model(dict(golden_w8, state=st2))
    assert torch.equal(a1, q.mc_predict(r2, xr, 3, 5)) and not torch.equal(a0, a1)


@pytest.mark.gpu
def test_graphed_predictor_queued_replays_keep_their_own_seeds(golden_lenet_mc):
    """Replays queued back to back without a host synchronisation in between (the launch-bound case the class exists for) must
    each run with THEIR seed and sample offset: the noise words travel in a fresh host buffer per call."""
    import quantised_bayesian_nets_amd as q
    g = golden_lenet_mc
    la = types.SimpleNamespace(activation_precision=7, weight_precision=8, p=0.2)
    m = q.ModelFactory.get_model("conv_lenet_mc", [1, 1, 28, 28], 10, True, la).load_reference_state(g["state"])
    gp = q.GraphedPredictor(m, 8)
    x = torch.rand(128, 1, 28, 28, generator=torch.Generator().manual_seed(9)).cuda()
    gp(x, 1)
    torch.cuda.synchronize()
    seeds = [(101, 0), (2 ** 35 + 7, 3), (55, 9), (101, 0), (7, 1), (8, 2), (9, 3), (10, 4)]
    outs = [gp(x, s, sample_begin=b) for s, b in seeds]                   # no synchronisation between the calls
    torch.cuda.synchronize()
    for (s, b), o in zip(seeds, outs):
        with q.mc_context(8, s, b):
            want = m.forward_mc(x).double().mean(0)
        np.testing.assert_allclose(o.cpu().numpy(), want.cpu().numpy(), rtol=1e-6, atol=1e-9)
    assert torch.equal(outs[0], outs[3]) and not torch.equal(outs[0], outs[1])


@pytest.mark.gpu
def test_fused_float_mlp_equals_layerwise_and_reference(golden_mlp_f32, monkeypatch):
    """BASELINE config 0 on the fused path (qbnn_mlp_bbb_f32_mc: one sampler launch + one launch for the whole 4 x 100 network)
    against the layer-by-layer kernels -- same Philox weights, so only the fp32 summation order of the dot products may differ
    (a few 1e-6 absolute on O(1) outputs) -- and against the reference's recorded (mu, var) and MC reduction (1e-5, as test_float_bbb_mlp_matches_reference);
    ragged batches and another input width too."""
    import quantised_bayesian_nets_amd as q
    g = golden_mlp_f32
    m = q.ModelFactory.get_model("linear_bbb", [g["in_dim"]], 1, False, types.SimpleNamespace(sigma_prior=-2.0)).load_reference_state(g["state"])
    x = torch.from_numpy(g["x"]).cuda()
    S = g["mu"].shape[0]
    with q.mc_context(S, g["seed"], 0):
        mu, var = m.forward_mc(x)
    np.testing.assert_allclose(mu.cpu().numpy(), g["mu"], rtol=1e-5, atol=g["mu_atol"])
    np.testing.assert_allclose(var.cpu().numpy(), g["var"], rtol=1e-5, atol=0)
    monkeypatch.setenv("QBNN_MLP_LAYERWISE", "1")
    with q.mc_context(S, g["seed"], 0):
        mu_l, var_l = m.forward_mc(x)
    monkeypatch.delenv("QBNN_MLP_LAYERWISE")
    np.testing.assert_allclose(mu.cpu().numpy(), mu_l.cpu().numpy(), rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(var.cpu().numpy(), var_l.cpu().numpy(), rtol=2e-5, atol=1e-8)
    gen = torch.Generator().manual_seed(5)
    for B, in_dim in ((1, g["in_dim"]), (33, g["in_dim"]), (1000, g["in_dim"])):
        xb = torch.randn(B, in_dim, generator=gen).cuda()
        with q.mc_context(3, 77, 4):
            a, b = m.forward_mc(xb)
        monkeypatch.setenv("QBNN_MLP_LAYERWISE", "1")
        with q.mc_context(3, 77, 4):
            c, d = m.forward_mc(xb)
        monkeypatch.delenv("QBNN_MLP_LAYERWISE")
        assert a.shape == (3, B, 1)
        np.testing.assert_allclose(a.cpu().numpy(), c.cpu().numpy(), rtol=1e-5, atol=5e-6)
        np.testing.assert_allclose(b.cpu().numpy(), d.cpu().numpy(), rtol=2e-5, atol=1e-8)


@pytest.mark.gpu
def test_linear_mc_int8_matches_reference(golden_mlp_mc_q):
    """Row a6+: `linear_mc` (mcdropout/models_mc.py:10-73, src/models/__init__.py:25-26), converted int8: in-kernel Philox masks, the
    per-element quantised dropout between the LinearReLUs and in front of both heads -- every layer of sample 0 bit for bit, all samples'
    (mu, var) and the regression reduction against the reference; injected masks; a 1000-row batch against the oracle."""
    import quantised_bayesian_nets_amd as q
    from oracle import oracle as orc
    g = golden_mlp_mc_q
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, p=g["meta"]["p"])
    m = q.ModelFactory.get_model("linear_mc", [g["meta"]["in_dim"]], 1, True, args).load_reference_state(g["state"])
    x = torch.from_numpy(g["x"]).cuda()
    S, seed = g["mu"].shape[0], g["meta"]["philox_seed"]
    rec = {}
    with q.mc_context(S, seed, 0):
        mu, var = m.forward_mc(x, record=rec)
    assert len(g["rec"]) == 10
    for k, v in g["rec"].items():
        assert np.array_equal(rec[k][0].cpu().numpy().reshape(v.shape), v), k
    np.testing.assert_allclose(mu.cpu().numpy(), g["mu"], rtol=1e-6, atol=0)          # a dequantised integer: one fp32 product
    np.testing.assert_allclose(var.cpu().numpy(), g["var"], rtol=RTOL, atol=0)
    mean, pv = q.mc_predict_regression(m, x, S, seed)
    np.testing.assert_allclose(mean.cpu().numpy(), g["mean"], rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(pv.cpu().numpy(), g["pred_var"], rtol=RTOL, atol=1e-9)
    with q.mc_context(1, seed, 2):                                                     # the single stochastic forward (Trainer.infer contract)
        mu2, var2 = m(x)
    assert torch.equal(mu2, mu[2]) and torch.equal(var2, var[2])
    keep = np.float32(1.0) - np.float32(g["meta"]["p"])
    B = x.shape[0]
    masks = [torch.from_numpy(np.stack([(orc.fill_uniform(B * 100, seed, di, s) < keep).astype(np.float32).reshape(B, 100) for s in (1, 3)]))
             for di in range(4)]
    with q.mc_context(2, 999, 0):
        mu_i, var_i = m.forward_mc(x, masks=masks)
    assert torch.equal(mu_i[0], mu[1]) and torch.equal(mu_i[1], mu[3]) and torch.equal(var_i[1], var[3])
    gen = torch.Generator().manual_seed(5)
    xb = torch.randn(1000, g["meta"]["in_dim"], generator=gen)
    net = orc.Int8MLPMCOracle(g["state"], 7)
    with q.mc_context(3, seed, 250):
        mub, varb = m.forward_mc(xb.cuda())
    mo, vo = net.forward(xb.numpy(), seed, 252)
    np.testing.assert_allclose(mub[2].cpu().numpy(), mo, rtol=1e-6, atol=0)
    np.testing.assert_allclose(varb[2].cpu().numpy(), vo, rtol=RTOL, atol=0)


@pytest.mark.gpu
def test_float_mc_dropout_graphs_match_reference(golden_mc_f32):
    """Rows a6+ / a7 with q=False: `linear_mc`, `conv_lenet_mc`, `conv_resnet_mc` as float graphs with the FloatFunctional
    BernoulliDropout (dropout.py:15-40) -- in-kernel Philox masks == injected masks bit for bit, and the outputs against the reference's
    recorded ones.  Tolerance = 1e-5 relative (north_star) plus twice the reference's own oneDNN-vs-ATen spread (recorded in the fixture)
    absolute; the MLP: four times its AVX-512-vs-AVX2 spread."""
    import quantised_bayesian_nets_amd as q
    from oracle import oracle as orc
    g = golden_mc_f32
    seed, p = g["meta"]["philox_seed"], g["meta"]["p"]
    args = types.SimpleNamespace(p=p)
    in_size = {"linear_mc": [g["meta"].get("in_dim", 13)], "conv_lenet_mc": [1, 28, 28], "conv_resnet_mc": [1, 3, 32, 32]}[g["model"]]
    out_size = 1 if g["model"] == "linear_mc" else 10
    m = q.ModelFactory.get_model(g["model"], in_size, out_size, False, args).load_reference_state(g["state"])
    assert len(m.dropouts()) == g["meta"]["n_dropouts"]
    x = torch.from_numpy(g["x"]).cuda()
    keep = np.float32(1.0) - np.float32(p)
    B = x.shape[0]
    if g["model"] == "linear_mc":
        S = g["mu"].shape[0]
        with q.mc_context(S, seed, 0):
            mu, var = m.forward_mc(x)
        mu_atol = 4.0 * g["refspread"]["max_abs"]          # four times the reference's AVX-512-vs-AVX2 distance from itself (conftest._mlp_f32)
        np.testing.assert_allclose(mu.cpu().numpy(), g["mu"], rtol=RTOL, atol=mu_atol)
        np.testing.assert_allclose(var.cpu().numpy(), g["var"], rtol=RTOL, atol=0)
        mean, pv = q.mc_predict_regression(m, x, S, seed)
        np.testing.assert_allclose(mean.cpu().numpy(), g["mean"], rtol=RTOL, atol=mu_atol)
        pv_err = np.abs(pv.cpu().numpy().astype(np.float64) - g["pred_var"])
        assert (pv_err <= RTOL * np.abs(g["pred_var"]) + pred_var_atol(g["mu"], mu_atol, RTOL).reshape(g["pred_var"].shape)).all(), pv_err.max()
        masks = [torch.from_numpy(np.stack([(orc.fill_uniform(B * 100, seed, di, s) < keep).astype(np.float32).reshape(B, 100) for s in range(S)]))
                 for di in range(4)]
        with q.mc_context(S, 4242, 0):
            mu_i, var_i = m.forward_mc(x, masks=masks)
        assert torch.equal(mu_i, mu) and torch.equal(var_i, var)
        with q.mc_context(1, seed, 3):
            mu3, _ = m(x)
        assert torch.equal(mu3, mu[3])
        return
    S = g["probs"].shape[0]
    with q.mc_context(S, seed, 0):
        probs = m.forward_mc(x)
    atol = 2 * g["refspread"]["max_abs"] + 1e-7
    np.testing.assert_allclose(probs.cpu().numpy(), g["probs"], rtol=RTOL, atol=atol)
    mean = q.mc_predict(m, x, S, seed)
    np.testing.assert_allclose(mean.cpu().numpy(), g["mean_probs"], rtol=RTOL, atol=atol)
    chans = [d_c for d_c in _mc_f32_mask_channels(g["model"])]
    masks = [torch.from_numpy(np.stack([(orc.fill_uniform(B * c, seed, di, s) < keep).astype(np.float32).reshape(B, c) for s in range(S)]))
             for di, c in enumerate(chans)]
    with q.mc_context(S, 777, 0):
        assert torch.equal(m.forward_mc(x, masks=masks), probs)
    with q.mc_context(1, seed, 1):
        assert torch.equal(m(x), probs[1])
    # a bigger batch against the oracle's fp32 forward (fp64 accumulation: 1e-5 relative + the same absolute floor)
    gen = torch.Generator().manual_seed(9)
    xb = torch.randn(16, *x.shape[1:], generator=gen) if g["model"] == "conv_resnet_mc" else torch.rand(16, *x.shape[1:], generator=gen)
    net = orc.F32MCOracle(g["state"])
    fwd = net.lenet if "lenet" in g["model"] else net.resnet
    with q.mc_context(2, seed, 40):
        pb = m.forward_mc(xb.cuda())
    np.testing.assert_allclose(pb[1].cpu().numpy(), fwd(xb.numpy(), seed, 41), rtol=RTOL, atol=atol)


def _mc_f32_mask_channels(model):
    if model == "conv_lenet_mc":
        return [20, 50, 500]
    out = [24]
    for planes, down in ((24, False), (48, True), (96, True), (192, True)):
        out += [planes, planes] + ([planes] if down else []) + [planes, planes]
    return out


@pytest.mark.gpu
def test_graphed_predictor_survives_a_layout_switch(golden_lenet_bbb):
    """An eager call that switches the stochastic layers' packed layout (`record=`: the any-geometry kernels' row-major form) frees the
    fragment-layout mu / sigma a captured graph points at.  layers.state_epoch() moves, and the next replay captures again instead of
    sampling from freed memory (advisor finding, round 3); two predictors on one model stay independent."""
    import quantised_bayesian_nets_amd as q
    from quantised_bayesian_nets_amd import layers as ql
    g = golden_lenet_bbb
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8)
    m = q.ModelFactory.get_model("conv_lenet_bbb", [1, 1, 28, 28], 10, True, args).load_reference_state(g["state"])
    x = torch.rand(64, 1, 28, 28, generator=torch.Generator().manual_seed(8)).cuda()
    gp, gp2 = q.GraphedPredictor(m, 5), q.GraphedPredictor(m, 3)
    want, want2 = q.mc_predict(m, x, 5, 77), q.mc_predict(m, x, 3, 78)
    assert torch.equal(gp(x, 77), want) and torch.equal(gp2(x, 78), want2)
    e0 = ql.state_epoch()
    with q.mc_context(2, 1, 0):
        m.forward_mc(x, record={})                      # row-major layouts: the packed fragment tensors are dropped
    junk = [torch.full((1 << 20,), 0x5a, dtype=torch.uint8, device="cuda") for _ in range(8)]     # reuse the freed blocks
    assert ql.state_epoch() != e0
    assert torch.equal(gp(x, 77), want) and torch.equal(gp2(x, 78), want2)
    assert torch.equal(gp(x, 77), want)                  # and the re-captured graph replays
    assert not hasattr(m.load_reference_state, "__wrapped__") and "load_reference_state" not in m.__dict__      # no monkey-patched loader
    del junk


@pytest.mark.gpu
@pytest.mark.parametrize("layerwise", [False, True], ids=["fused", "layerwise"])
def test_float_bbb_mlp_every_input_width(golden_mlp_f32_width, layerwise, monkeypatch):
    """BASELINE config 0 at SURVEY 8(d) C1's other input widths (in_dim 1, 4, 6, 8, 11; 13 is the benchmark's): the fused two-launch MLP
    (rows padded to 4 floats: 1 and 6 are the widths that padding has to get right) and the layer-by-layer path, per-sample (mu, var)
    and the regression reduction against the reference; same measured tolerance as in_dim 13."""
    import quantised_bayesian_nets_amd as q
    g = golden_mlp_f32_width
    if layerwise:
        monkeypatch.setenv("QBNN_MLP_LAYERWISE", "1")
    m = q.ModelFactory.get_model("linear_bbb", [g["in_dim"]], 1, False, types.SimpleNamespace(sigma_prior=-2.0)).load_reference_state(g["state"])
    x = torch.from_numpy(g["x"]).cuda()
    S = g["mu"].shape[0]
    with q.mc_context(S, g["seed"], 0):
        mu, var = m.forward_mc(x)
    np.testing.assert_allclose(mu.cpu().numpy(), g["mu"], rtol=1e-5, atol=g["mu_atol"])
    np.testing.assert_allclose(var.cpu().numpy(), g["var"], rtol=1e-5, atol=0)
    mean, pv = q.mc_predict_regression(m, x, S, g["seed"])
    np.testing.assert_allclose(mean.cpu().numpy(), g["mean"], rtol=1e-5, atol=g["mu_atol"])
    pv_err = np.abs(pv.cpu().numpy().astype(np.float64) - g["pred_var"])
    assert (pv_err <= 1e-5 * np.abs(g["pred_var"]) + pred_var_atol(g["mu"], g["mu_atol"]).reshape(g["pred_var"].shape)).all(), pv_err.max()


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 5: parity at the sizes bench.py times (verdict round 4, item 4) -- every secondary workload has an oracle check at its bench size
@pytest.mark.gpu
@pytest.mark.parametrize("qseed", [None, 3])
def test_resnet_mc_int8_bench_size_against_oracle(qseed):
    """`conv_resnet_mc` int8 at the size `bench.py --workload resnet_mc` times (B = 256): the DROP instantiations of every fused block
    kernel -- the 16-wave layer-1 kernel, the weights-stationary 24 -> 48 block and 48-channel chain, the wide down blocks, the ring
    chains with their one-bit mask tables -- on full work-item ranges, one MC sample at a global index beyond the first launch's
    against the CPU oracle (Int8ResNetMCOracle, pinned to the reference by tests/golden/make_golden_resnet_mc.py).  qseed: the same with
    random output scales / zero points of every conv, mask and Add (the reference-calibrated ones sit in a narrow band).
    Probabilities at 1e-5 relative: the integer path is bit-exact, the softmax is fp32 on both sides."""
    import quantised_bayesian_nets_amd as q
    from conftest import load_golden
    from oracle import oracle as orc
    g = load_golden("resnet_mc_a7w8.npz")
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, p=g["meta"]["p"])
    state = g["state"] if qseed is None else _perturb_activation_qparams(g["state"], qseed)
    m = q.ModelFactory.get_model("conv_resnet_mc", [1, 3, 32, 32], 10, True, args).load_reference_state(state)
    x = torch.randn(256, 3, 32, 32, generator=torch.Generator().manual_seed(31))
    seed, first, S = 77, 40, 3
    assert m._can_fuse_blocks(x.cuda(), None)
    with q.mc_context(S, seed, first):
        p = m.forward_mc(x.cuda()).cpu().numpy()
    net = orc.Int8ResNetMCOracle(state, 7)
    for s in (0, S - 1):
        np.testing.assert_allclose(p[s], net.forward(x.numpy(), seed, first + s), rtol=RTOL, atol=1e-8)


@pytest.mark.gpu
def test_resnet_mc_f32_bench_size_against_oracle():
    """The float MC-Dropout ResNet at the size `bench.py --workload resnet_mc_f32` times (B = 256): one MC sample against the oracle's
    fp32 forward with fp64 accumulation (F32MCOracle.resnet), 1e-5 relative + twice the reference's own oneDNN-vs-ATen spread recorded in
    the fixture (the tolerance of test_float_mc_dropout_graphs_match_reference)."""
    import quantised_bayesian_nets_amd as q
    from conftest import _npz
    from oracle import oracle as orc
    g = _npz("resnet_mc_f32.npz")
    m = q.ModelFactory.get_model("conv_resnet_mc", [1, 3, 32, 32], 10, False, types.SimpleNamespace(p=g["meta"]["p"])).load_reference_state(g["state"])
    x = torch.randn(256, 3, 32, 32, generator=torch.Generator().manual_seed(33))
    seed = g["meta"]["philox_seed"]
    with q.mc_context(2, seed, 8):
        p = m.forward_mc(x.cuda())
    atol = 2 * g["refspread"]["max_abs"] + 1e-7
    np.testing.assert_allclose(p[1].cpu().numpy(), orc.F32MCOracle(g["state"]).resnet(x.numpy(), seed, 9), rtol=RTOL, atol=atol)


@pytest.mark.gpu
def test_lenet_bbb_bench_size_every_sample_against_oracle(golden_lenet_bbb):
    """The int8 BBB LeNet at the size `bench.py --workload lenet_bbb` times: B = 128, S = 100 (global sample indices 0..99) on its fast
    path (the sampler writing the fragment layouts, the one-MFMA conv 1, the fused 20 -> 50 conv, the int8 GEMMs), EVERY sample's
    probabilities against the oracle, as test_lenet_mc_full_sample_count_against_oracle does for config 1."""
    import quantised_bayesian_nets_amd as q
    from oracle import oracle as orc
    g = golden_lenet_bbb
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8)
    m = q.ModelFactory.get_model("conv_lenet_bbb", [1, 1, 28, 28], 10, True, args).load_reference_state(g["state"])
    S, seed = 100, 4321
    x = torch.rand(128, 1, 28, 28, generator=torch.Generator().manual_seed(35))
    assert m._can_run_fast(x.cuda(), None)
    with q.mc_context(S, seed, 0):
        p = m.forward_mc(x.cuda()).cpu().numpy()
    net = orc.Int8LeNetBBBOracle(g["state"], 7, 8)
    ref = np.stack([net.forward(x.numpy(), seed, s) for s in range(S)])
    np.testing.assert_allclose(p, ref, rtol=RTOL, atol=1e-8)
    mean = q.mc_predict(m, x.cuda(), S, seed).cpu().numpy()
    np.testing.assert_allclose(mean, ref.astype(np.float64).mean(0), rtol=RTOL, atol=1e-8)


@pytest.mark.gpu
def test_sampler_n24_layout_draws_the_same_weights(golden_w8):
    """The weight sampler writing QBNN_LAYOUT_MFMA32_N24 fragments (the 16-wave 48-channel kernel's operand) draws the SAME weight for
    every logical element (n, k) as in the MFMA32 layout -- the Philox counter is the element's index in the reference's OHWI order, not
    its position in a layout -- single-layer and all-layers-in-one-launch entry points, W8 and W4, a sample range across 256."""
    import quantised_bayesian_nets_amd as q
    from quantised_bayesian_nets_amd import layers as ql
    for w_bits in (8, 4):
        args = types.SimpleNamespace(activation_precision=7, weight_precision=w_bits)
        m = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, args).load_reference_state(golden_w8["state"])
        layer = m.layers[4][1].stem[3]          # 48 -> 48, 3x3
        cout, k, krow = 48, 432, 144
        KS = 3 * 160 // 32

        def logical(w, tr):
            f = w.cpu().numpy().reshape(w.shape[0], 48 // tr if tr == 24 else 2, KS, 2, 32, 16)
            dense = f.transpose(0, 1, 4, 2, 3, 5).reshape(w.shape[0], -1, 32, KS * 32)      # [S][tile][row][padded k]
            kp = (np.arange(k) // krow) * 160 + np.arange(k) % krow
            return np.stack([dense[:, n // tr, n % tr][:, kp] for n in range(cout)], axis=1), dense
        got = {}
        for layout, tr in ((ql.LAYOUT_MFMA32, 32), (ql.LAYOUT_MFMA32_N24, 24)):
            layer.set_layout(layout)
            with q.mc_context(5, 77, 254):
                one = layer.sample_weights(torch.device("cuda"))
                ql.sample_all_weights([layer], torch.device("cuda"))
                multi = layer.sample_weights(torch.device("cuda"))
            assert torch.equal(one, multi)
            got[layout], dense = logical(one, tr)
            if tr == 24:
                assert (dense[:, :, 24].max() == 1) and not dense[:, :, 25:].any()
        assert np.array_equal(got[ql.LAYOUT_MFMA32], got[ql.LAYOUT_MFMA32_N24])
        lo, hi = (-8, 7) if w_bits == 4 else (-128, 127)
        assert got[ql.LAYOUT_MFMA32].min() >= lo and got[ql.LAYOUT_MFMA32].max() <= hi
        layer.set_layout(ql.LAYOUT_MFMA32)
        # QBNN_LAYOUT_MFMA32_TAIL on a 24 -> 24 3x3 conv (72-byte kernel rows: 2 full k-steps each + the three tails in a seventh)
        l24 = m.layers[3][0].stem[3]
        k24 = 216
        draws = {}
        for layout in (ql.LAYOUT_MFMA32, ql.LAYOUT_MFMA32_TAIL):
            l24.set_layout(layout)
            with q.mc_context(5, 77, 254):
                one = l24.sample_weights(torch.device("cuda"))
                ql.sample_all_weights([l24], torch.device("cuda"))
                assert torch.equal(one, l24.sample_weights(torch.device("cuda")))
            KS24 = 9 if layout == ql.LAYOUT_MFMA32 else 7
            assert one.shape[1] == KS24 * 1024
            dense = one.cpu().numpy().reshape(5, KS24, 2, 32, 16).transpose(0, 3, 1, 2, 4).reshape(5, 32, KS24 * 32)      # [S][row][packed k]
            kk = np.arange(k24)
            kh, j = kk // 72, kk % 72
            kp = kh * 96 + j if layout == ql.LAYOUT_MFMA32 else np.where(j < 64, kh * 64 + j, 192 + kh * 8 + (j - 64))
            draws[layout] = dense[:, :24][:, :, kp]
            ones = np.zeros(KS24 * 32, np.int8); ones[kp] = 1
            assert np.array_equal(dense[0, 24], ones) and not dense[:, 25:].any()
        assert np.array_equal(draws[ql.LAYOUT_MFMA32], draws[ql.LAYOUT_MFMA32_TAIL])
        l24.set_layout(ql.LAYOUT_MFMA32)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(3, 24, 3, 1, 1, 32), (24, 24, 3, 1, 1, 16), (24, 48, 3, 2, 1, 16), (48, 96, 1, 2, 0, 8), (96, 192, 3, 1, 1, 4), (20, 50, 5, 1, 2, 14),
                                  # round 6: the LDS-tiled forms (csrc/qbnn_q8t.hip) -- the ResNet's seven 3 x 3 geometries, the 3-channel stem, ragged image groups (B = 3
                                  # against 2 / 8 images per block), ragged channel groups (40 of 2 x 32, 100 of 2 x 96) and a 3 x 3 conv no tiled form matches
                                  (24, 24, 3, 1, 1, 32), (24, 48, 3, 2, 1, 32), (48, 48, 3, 1, 1, 16), (48, 96, 3, 2, 1, 16), (96, 96, 3, 1, 1, 8), (96, 192, 3, 2, 1, 8),
                                  (192, 192, 3, 1, 1, 4), (3, 32, 3, 1, 1, 32), (3, 8, 3, 1, 1, 32), (24, 40, 3, 1, 1, 32), (96, 100, 3, 1, 1, 8), (192, 24, 3, 1, 1, 4), (48, 48, 3, 1, 1, 8)])
def test_qat_int8_conv_entry_points_against_numpy(case):
    """qbnn_grid_to_i8_mc + qbnn_conv2d_q8_f32_mc through the C ABI (round 5: the QAT convs on the int8 matrix pipe) against the same sum in numpy:
    fake-quantised operands with PER-SAMPLE scales / zero points (activations on 7-bit grids, weights on int8 grids with non-zero zero points),
    every tile form (128 x 32 for Cout <= 32, 64 x 64, the byte-gather form for Cin = 3), stride 2, 1 x 1 and 5 x 5 kernels, ragged pixel counts, and the
    fused tail Z / c + b, bn, ReLU with the per-workgroup (min, max) partials.  y = fl32(fl64(N) * fl64(s_x) * fl64(s_w)) with N the exact integer sum:
    compared at 1e-6 relative (one fp32 rounding of the tail's four steps)."""
    import ctypes as C
    from quantised_bayesian_nets_amd import _lib
    L = _lib.lib()
    cin, cout, k, stride, pad, H = case
    rng = np.random.default_rng(cin * 1000 + cout)
    S, B = 3, 3
    st = _lib.current_stream()
    s_x = (10 ** rng.uniform(-2, -1, S)).astype(np.float32)
    z_x = rng.integers(0, 128, S)
    s_w = (10 ** rng.uniform(-3, -2, S)).astype(np.float32)
    z_w = rng.integers(-40, 41, S).astype(np.int32)
    q_x = rng.integers(0, 128, (S, B, H, H, cin))
    q_w = rng.integers(-128, 128, (S, cout, k, k, cin))
    xf = ((q_x - z_x[:, None, None, None, None]).astype(np.float32) * s_x[:, None, None, None, None]).astype(np.float32)      # what a FakeQuantize leaves
    wf = ((q_w - z_w[:, None, None, None, None]).astype(np.float32) * s_w[:, None, None, None, None]).astype(np.float32)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    xd, wd, sxd, swd, zwd = dev(xf), dev(wf.reshape(S, -1)), dev(s_x), dev(s_w), dev(z_w)
    n_x, n_w = xf[0].size, wf[0].size
    xq = torch.empty((S, n_x), dtype=torch.int8, device="cuda")
    wq = torch.empty((S, n_w), dtype=torch.int8, device="cuda")
    _lib.check(L.qbnn_grid_to_i8_mc(_lib.ptr(xd), n_x, n_x, _lib.ptr(sxd), None, _lib.ptr(xq), S, st))
    _lib.check(L.qbnn_grid_to_i8_mc(_lib.ptr(wd), n_w, n_w, _lib.ptr(swd), _lib.ptr(zwd), _lib.ptr(wq), S, st))
    assert np.array_equal(xq.cpu().numpy().reshape(q_x.shape), q_x - z_x[:, None, None, None, None])      # the centred activation integers
    assert np.array_equal(wq.cpu().numpy().reshape(q_w.shape), q_w)                                        # the raw weight integers
    div = dev((rng.uniform(0.5, 2.0, cout)).astype(np.float32))
    bias = dev((rng.normal(size=cout) * 0.1).astype(np.float32))
    alpha = dev((rng.uniform(0.5, 1.5, cout)).astype(np.float32))
    beta = dev((rng.normal(size=cout) * 0.1).astype(np.float32))
    Ho = (H + 2 * pad - k) // stride + 1
    y = torch.full((S, B, Ho, Ho, cout), float("nan"), dtype=torch.float32, device="cuda")
    nblk = int(L.qbnn_conv2d_q8_blocks(B, H, H, cin, cout, k, stride, pad))
    mm = torch.full((S * nblk * 2,), float("nan"), dtype=torch.float32, device="cuda")
    _lib.check(L.qbnn_conv2d_q8_f32_mc(_lib.ptr(xq), n_x, _lib.ptr(wq), n_w, _lib.ptr(sxd), _lib.ptr(swd), _lib.ptr(zwd), _lib.ptr(div), _lib.ptr(bias),
                                       _lib.ptr(alpha), _lib.ptr(beta), _lib.ptr(y), y[0].numel(), B, H, H, cin, cout, k, stride, pad, 1, S, _lib.ptr(mm), st))
    torch.cuda.synchronize()
    got = y.cpu().numpy()
    for s in range(S):
        m_x = torch.from_numpy((q_x[s] - z_x[s]).astype(np.float64)).permute(0, 3, 1, 2)
        m_w = torch.from_numpy((q_w[s] - int(z_w[s])).astype(np.float64)).permute(0, 3, 1, 2)
        N = torch.nn.functional.conv2d(m_x, m_w, stride=stride, padding=pad).permute(0, 2, 3, 1).numpy()      # exact in fp64: |N| < 2^53
        v = (N * (float(s_x[s]) * float(s_w[s]))).astype(np.float32)
        v = (v / div.cpu().numpy()).astype(np.float32)
        v = (v + bias.cpu().numpy()).astype(np.float32)
        v = (v * alpha.cpu().numpy()).astype(np.float32)
        v = np.maximum((v + beta.cpu().numpy()).astype(np.float32), 0)
        np.testing.assert_allclose(got[s], v, rtol=1e-6, atol=1e-7)
        part = mm.cpu().numpy().reshape(S, nblk, 2)[s]
        assert np.isclose(part[:, 0].min(), got[s].min()) and np.isclose(part[:, 1].max(), got[s].max())


# ------------------------------------------------------------------------------------------ round 6: advisor findings of round 5
@pytest.mark.parametrize("name,model,B", [("mlp_bbb_qat.npz", "linear_bbb", 7), ("mlp_bbb_qat.npz", "linear_bbb", 1), ("lenet_bbb_qat.npz", "conv_lenet_bbb", 7),
                                          ("lenet_bbb_qat.npz", "conv_lenet_bbb", 3)])
def test_qat_eval_odd_batches_against_oracle(name, model, B):
    """The QAT Linears run through qbnn_conv2d_q8_f32_mc (1 x 1 conv form): its 128 x 32 tile must take the heads' outputs -- Linear(100, 1) with
    y_ss = B, Linear(500, 10) with y_ss = 10 B -- for ANY batch (a ragged last batch: B % 4 != 0, odd B), where round 5 answered QBNN_E_INVALID.
    Live-observer evaluation of the MLP / LeNet against the CPU oracle (conv_qat.py:139-167, linear_qat.py:18-41), sample after sample."""
    import os
    import quantised_bayesian_nets_amd as q
    from oracle import oracle as orc
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
    st = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
    args = types.SimpleNamespace(sigma_prior=-2.0, activation_precision=7, weight_precision=8, qat_eval=True)
    shape = {"linear_bbb": [13], "conv_lenet_bbb": [1, 28, 28]}[model]
    m = q.ModelFactory.get_model(model, shape, 1 if model == "linear_bbb" else 10, True, args).load_reference_state(st)
    gen = torch.Generator().manual_seed(40 + B)
    x = torch.randn(B, 13, generator=gen) if model == "linear_bbb" else torch.rand(B, 1, 28, 28, generator=gen)
    S, seed = 3, 17
    net = orc.QATOracle(st)
    with q.mc_context(S, seed, 0):
        out = m.forward_mc(x.cuda())
    for s in range(S):
        if model == "linear_bbb":
            mu, var = net.mlp(x.numpy(), seed, s)
            np.testing.assert_allclose(out[0][s].cpu().numpy(), mu, rtol=1e-5, atol=4e-6)
            np.testing.assert_allclose(out[1][s].cpu().numpy(), var, rtol=1e-5, atol=1e-8)
        else:
            np.testing.assert_allclose(out[s].cpu().numpy(), net.lenet(x.numpy(), seed, s), rtol=1e-5, atol=2e-6)


def test_int8_grid_output_needs_a_grid_of_at_most_128_steps():
    """q - z spans +-(qmax - qmin): qbnn_fake_quant_ex_f32_mc's int8 output (the activation operand of the int8-pipe QAT convs) is refused for a
    grid wider than 128 steps instead of wrapping silently, the Python FakeQuantize then leaves no `_grid` (its consumer takes the fp64 path),
    and the QAT constructors hold the reference's bit-width contract (quant_utils.py:120-121)."""
    import quantised_bayesian_nets_amd as q
    from quantised_bayesian_nets_amd import _lib, models_qat
    L = _lib.lib()
    S, n = 2, 64
    x = torch.randn(S, n, device="cuda")
    y = torch.empty_like(x)
    q8 = torch.empty((S, n), dtype=torch.int8, device="cuda")
    sc = torch.full((S,), 0.05, device="cuda")
    zp = torch.full((S,), 100, dtype=torch.int32, device="cuda")
    st = _lib.current_stream()
    assert L.qbnn_fake_quant_ex_f32_mc(_lib.ptr(x), n, _lib.ptr(y), n, n, _lib.ptr(sc), _lib.ptr(zp), 0, 255, 0, _lib.ptr(q8), S, st) == -1
    assert b"128 steps" in L.qbnn_last_error()
    assert L.qbnn_fake_quant_ex_f32_mc(_lib.ptr(x), n, _lib.ptr(y), n, n, _lib.ptr(sc), _lib.ptr(zp), 0, 255, 0, None, S, st) == 0      # fp32 output only: any grid
    assert L.qbnn_fake_quant_ex_f32_mc(_lib.ptr(x), n, _lib.ptr(y), n, n, _lib.ptr(sc), _lib.ptr(zp), 0, 127, 0, _lib.ptr(q8), S, st) == 0
    torch.cuda.synchronize()
    got = q8.cpu().numpy().astype(np.int32)
    want = np.clip(np.rint(x.cpu().numpy() * np.float32(1.0 / np.float32(0.05))) + 100, 0, 127) - 100
    assert np.array_equal(got, want.astype(np.int32))
    with pytest.raises(AssertionError):
        q.ModelFactory.get_model("linear_bbb", [13], 1, True, types.SimpleNamespace(sigma_prior=-2.0, activation_precision=8, weight_precision=8, qat_eval=True))
    with q.mc_context(S, 1, 0):
        fq = models_qat.FakeQuantize(0, 255)
        assert getattr(fq(torch.randn(S, 4, 8, device="cuda")), "_grid", None) is None
        fq7 = models_qat.FakeQuantize(0, 127)
        assert getattr(fq7(torch.randn(S, 4, 8, device="cuda")), "_grid", None) is not None


def test_prepared_multi_launch_refuses_what_prepare_did_not_bake(golden_ensemble):
    """qbnn_block_*_i8_multi_launch picks its kernel from (a_hi, w_layout, n_blocks, with_stem): the library remembers what _multi_prepare baked into
    each dev_args block and answers QBNN_E_INVALID for anything else (another fragment layout would read the weights scrambled), for a device
    block it never prepared, and for QBNN_BLOCK_POOL_OUT on the multi-call forms (they write the full map)."""
    import quantised_bayesian_nets_amd as q
    from quantised_bayesian_nets_amd import _lib
    from conftest import synth_ensemble_members
    n = 4
    members = synth_ensemble_members(golden_ensemble, n)
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, model="conv_resnet_sgld", samples=n)
    net = q.ModelFactory.get_model("conv_resnet_sgld", [1, 3, 32, 32], 10, True, args, training_mode=False).load_reference_state(members)
    x = torch.randn(8, 3, 32, 32, generator=torch.Generator().manual_seed(2)).cuda()
    with q.mc_context(n, 0, 0):
        want = net.forward_mc(x)
    (plan,) = net._plans.values()
    assert plan["dev_steps"], "the prepared-launch path is the default"
    L, st, M, B, a_hi = _lib.lib(), _lib.current_stream(), plan["M"], 8, plan["a_hi"]
    stem, dargs = plan["steps"][0], _lib.ptr(plan["dev_steps"][0])
    lay = stem[1][0].blocks[0].w_layout
    assert stem[0] == "stem"
    assert L.qbnn_block_chain_i8_multi_launch(dargs, M, 1, B, 32, 24, a_hi, lay, 2, 1, st) == 0
    for bad in (dict(w_layout=0 if lay != 0 else 3), dict(a_hi=a_hi // 2), dict(B=B + 1), dict(M=M + 1)):
        kw = dict(M=M, B=B, a_hi=a_hi, w_layout=lay)
        kw.update(bad)
        assert L.qbnn_block_chain_i8_multi_launch(dargs, kw["M"], 1, kw["B"], 32, 24, kw["a_hi"], kw["w_layout"], 2, 1, st) == -1, bad
        assert b"_multi_prepare" in L.qbnn_last_error()
    stray = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    assert L.qbnn_block_chain_i8_multi_launch(_lib.ptr(stray), M, 1, B, 32, 24, a_hi, lay, 2, 1, st) == -1
    down, ddown = plan["steps"][1], _lib.ptr(plan["dev_steps"][1])
    dlay = down[1][0].desc.contents.blk.w_layout
    assert down[0] == "down" and L.qbnn_block_down_i8_multi_launch(ddown, M, B, down[2], down[3], dlay + 1, 1, st) == -1
    # POOL_OUT on a multi-call form: refused at argument-building time
    chain = [s for s in plan["steps"] if s[0] == "chain"][-1]
    chain[1][0].blocks[0].flags = 1
    try:
        assert L.qbnn_block_chain_i8_multi(chain[1], M, 0, B, chain[2], chain[3], a_hi, 1, st) == -1 and b"POOL_OUT" in L.qbnn_last_error()
        buf = torch.empty(int(L.qbnn_chain_multi_args_bytes(M, 1)), dtype=torch.uint8, device="cuda")
        assert L.qbnn_block_chain_i8_multi_prepare(chain[1], M, 0, B, a_hi, 1, _lib.ptr(buf), st) == -1 and b"POOL_OUT" in L.qbnn_last_error()
    finally:
        chain[1][0].blocks[0].flags = 0
    with q.mc_context(n, 0, 0):                 # the plan still runs, same bits
        assert torch.equal(net.forward_mc(x), want)


def test_qat_resnet_graph_replay_equals_eager():
    """The QAT evaluation pass (live observers, ~100 launches since the weight pipelines come batched) as ONE captured HIP graph: replays with new
    seeds equal eager evaluations of a model whose observers have the same history (GraphedPredictor's warm-up pass runs eagerly with seed 0; the
    capture itself executes nothing) -- bit for bit, observer states included."""
    import quantised_bayesian_nets_amd as q
    from conftest import load_golden
    g = load_golden("resnet_bbb_qat.npz")
    args = types.SimpleNamespace(sigma_prior=-2.0, activation_precision=7, weight_precision=8, qat_eval=True)
    mk = lambda: q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, args).load_reference_state(g["state"])
    x = torch.randn(37, 3, 32, 32, generator=torch.Generator().manual_seed(6)).cuda()
    S = 4
    mg, me = mk(), mk()
    gp = q.GraphedPredictor(mg, S)
    a = [gp(x, seed).cpu().numpy() for seed in (5, 6)]
    with q.mc_context(S, 0, 0):
        me.forward_mc(x)
    b = [q.mc_predict(me, x, S, seed).cpu().numpy() for seed in (5, 6)]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    sg, se = mg.prepared_state(), me.prepared_state()
    for k in sg:
        if k.endswith("min_val") or k.endswith("max_val"):
            assert float(sg[k]) == float(se[k]), k
