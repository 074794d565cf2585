"""GPU parity tests, rows a3-a6, a9, a10: the int8 Bayes-by-backprop ResNet-18 (sampler, layers, fused blocks, whole model, bit-width sweep, random quantisation parameters, A/B switches) (run with -m gpu on an MI355X): the HIP path, called through the C ABI of libqbnn_hip.so, against
(a) the golden vectors recorded from the real reference and (b) the CPU oracle on the same seeded inputs.
Integer tensors: bit-exact.  fp32 probabilities / moments: 1e-5 relative (BASELINE.json north_star)."""
import ctypes as C
import os
import types

import numpy as np
import pytest
import torch

from gpu_common import RTOL, _model, _pack_per_sample      # noqa: F401

pytestmark = pytest.mark.gpu



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


@pytest.mark.parametrize("a_bits,w_bits", [(3, 8), (4, 8), (5, 8), (6, 8), (7, 3), (7, 5), (7, 6), (7, 7)], ids=lambda v: str(v))
def test_bit_width_sweep_full_batch_against_oracle(a_bits, w_bits):
    """EVERY point of the reference's sweep (experiments/run_all_quant.sh:11-37) away from the two BASELINE points, at the full batch: A3 ... A6
    move every activation clamp (src/utils.py:25-30: [0, 7] ... [0, 63]) and the accumulator bound of the 1.5 * 2^23 start, W3 ... W7 the
    sampled-weight clamp ([-4, 3] ... [-64, 63], src/utils.py:32-37).  Fused path and per-block launches, B = 256, two samples, against
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
            # the case must exercise the requantisation, not saturate it: each tensor uses more than a third of the levels its clamp leaves it (capped at 8 / 3;
            # a ReLU-fused output lives on [z, a_hi]: A3 with z = 5 has three levels in all)
            need = lambda levels, cap: min(cap, max(1, levels // 3))
            assert len(np.unique(t)) > need(a_hi - z_a + 1, 8) and len(np.unique(u)) > need(a_hi + 1, 8) and len(np.unique(ref)) > need(a_hi - z_o + 1, 3) - 1, \
                "degenerate case: outputs saturated"
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
