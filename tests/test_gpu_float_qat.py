"""GPU parity tests, rows a1, a2, f3, f4: float BBB graphs, QAT evaluation with live observers (int8-pipe convs, batched weight pipelines), native prepare -> calibrate -> convert (run with -m gpu on an MI355X): the HIP path, called through the C ABI of libqbnn_hip.so, against
(a) the golden vectors recorded from the real reference and (b) the CPU oracle on the same seeded inputs.
Integer tensors: bit-exact.  fp32 probabilities / moments: 1e-5 relative (BASELINE.json north_star)."""
import ctypes as C
import os
import types

import numpy as np
import pytest
import torch

from gpu_common import RTOL      # noqa: F401

pytestmark = pytest.mark.gpu



def pred_var_atol(mu_ref, mu_atol, rtol=1e-5):
    """Absolute tolerance per row for the predictive variance mean_s(var_s) + var_s(mu_s) of a regression model whose per-sample means
    are only known to rtol / mu_atol (north_star: 1e-5 relative on the moments): the variance ACROSS samples subtracts nearly equal
    numbers, so a perturbation d_s of mu_s moves it by mean_s(2 (mu_s - m)(d_s - mean d)) <= 2 std_s(mu) max|d| -- first order, derived,
    instead of a looser relative tolerance on the sum."""
    mu_ref = np.asarray(mu_ref, dtype=np.float64)
    dmax = rtol * np.abs(mu_ref).max(axis=0) + mu_atol
    S = mu_ref.shape[0]
    return (2.0 * mu_ref.std(axis=0) * dmax + dmax ** 2) * (S / max(S - 1.0, 1.0))      # (the unbiased estimator divides by S - 1)


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


def _mc_f32_mask_channels(model):
    if model == "conv_lenet_mc":
        return [20, 50, 500]
    out = [24]
    for planes, down in ((24, False), (48, True), (96, True), (192, True)):
        out += [planes, planes] + ([planes] if down else []) + [planes, planes]
    return out


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


@pytest.mark.parametrize("switch", ["QBNN_F32_TAPMASK=0", "QBNN_QAT_PRESAMPLE=0", "QBNN_Q8_TILED=0", "QBNN_QAT_WBATCH=0", "QBNN_QAT_ADDFQ=0", "QBNN_F32_WBATCH=0"])
def test_float_path_switches_give_the_same_bits(switch, tmp_path):
    """The fp32 / fp64 conv's two gather forms (per-row tap masks against per-element bounds compares), the QAT weight pipelines on side
    streams against in line, the QAT 3 x 3 convs LDS-tiled (round 6, csrc/qbnn_q8t.hip) against the gather forms of round 5 (same integer sums, same tail),
    all layers' weight pipelines in four launches (qbnn_qat_weights_mc) against ~15 launches per layer, and the Add's FakeQuantize from its operands' integers
    (qbnn_fake_quant_add_q8_mc) against the stored fp32 sum, and the float ResNet's 21 weight draws in one launch (qbnn_sample_weights_f32_batch) against one per layer: the float BBB ResNet and the QAT evaluation, ragged batch of 70, give bit-identical probabilities either way."""
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
