"""GPU parity tests, rows a6-a7 on the small graphs: int8 LeNet / MLP (BBB and MC-Dropout), the MC-Dropout ResNet, their dedicated kernels (run with -m gpu on an MI355X): the HIP path, called through the C ABI of libqbnn_hip.so, against
(a) the golden vectors recorded from the real reference and (b) the CPU oracle on the same seeded inputs.
Integer tensors: bit-exact.  fp32 probabilities / moments: 1e-5 relative (BASELINE.json north_star)."""
import ctypes as C
import os
import types

import numpy as np
import pytest
import torch

from gpu_common import RTOL, _pack_per_sample      # noqa: F401

pytestmark = pytest.mark.gpu



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
