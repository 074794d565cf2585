"""QAT -> int8 conversion (quantised_bayesian_nets_amd/convert.py) against what the reference's own convert() produced
for the same QAT layers (tests/golden/make_golden_convert.py).  CPU-only host logic; everything must match exactly."""
import os
import types

import numpy as np
import pytest
import torch

from quantised_bayesian_nets_amd import convert as cv

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN = os.path.join(GOLDEN_DIR, "convert_layers_a7w8.npz")


def _layers():
    d = np.load(GOLDEN)
    names = sorted({"/".join(k.split("/")[:2]) for k in d.files})
    return d, names


def _observer(a):
    return cv.ObserverState(np.float32(a[0]), np.float32(a[1]), int(a[2]), int(a[3]))


@pytest.mark.parametrize("idx", range(10))
def test_convert_layer_matches_reference(idx):
    d, names = _layers()
    p = names[idx] + "/"
    bn = None
    if p + "bn.running_mean" in d.files:
        bn = dict(running_mean=d[p + "bn.running_mean"], running_var=d[p + "bn.running_var"], eps=float(d[p + "bn.eps"]),
                  weight=d[p + "bn.weight"], bias=d[p + "bn.bias"])
    bias = d[p + "bias"] if p + "bias" in d.files else None
    got = cv.convert_layer(d[p + "mu"], d[p + "rho"], bias, _observer(d[p + "obs.weight"]), _observer(d[p + "obs.std"]),
                           _observer(d[p + "obs.act"]), _observer(d[p + "obs.add"]), _observer(d[p + "obs.mul"]), bn)
    expect = {k[len(p + "expect/"):]: d[k] for k in d.files if k.startswith(p + "expect/")}
    assert set(expect) == set(got), (sorted(expect), sorted(got))
    for k, v in expect.items():
        g = np.asarray(got[k])
        if np.issubdtype(np.asarray(v).dtype, np.integer) or v.dtype == np.int8:
            assert np.array_equal(g, v), (p, k)
        else:
            assert np.array_equal(g.astype(np.float64), np.asarray(v).astype(np.float64)), (p, k, g, v)


def test_convert_qat_module_duck_typing():
    """convert_qat_module reads a reference-style QAT module object (attributes only)."""
    d, names = _layers()
    p = [n for n in names if n.endswith("layers.3.0.stem.3") and n.startswith("w8")][0] + "/"

    def fq(a):
        o = types.SimpleNamespace(min_val=torch.tensor(np.float32(a[0])), max_val=torch.tensor(np.float32(a[1])), quant_min=int(a[2]), quant_max=int(a[3]))
        return types.SimpleNamespace(activation_post_process=o, quant_min=int(a[2]), quant_max=int(a[3]))

    bn = types.SimpleNamespace(running_mean=torch.from_numpy(d[p + "bn.running_mean"]), running_var=torch.from_numpy(d[p + "bn.running_var"]),
                               eps=float(d[p + "bn.eps"]), weight=torch.from_numpy(d[p + "bn.weight"]), bias=torch.from_numpy(d[p + "bn.bias"]))
    mod = types.SimpleNamespace(weight=torch.from_numpy(d[p + "mu"]), std=torch.from_numpy(d[p + "rho"]), bias=None, bn=bn,
                                weight_fake_quant=fq(d[p + "obs.weight"]), std_fake_quant=fq(d[p + "obs.std"]),
                                activation_post_process=fq(d[p + "obs.act"]),
                                add_weight=types.SimpleNamespace(activation_post_process=fq(d[p + "obs.add"])),
                                mul_noise=types.SimpleNamespace(activation_post_process=fq(d[p + "obs.mul"])))
    got = cv.convert_qat_module(mod)
    assert np.array_equal(got["weight"], d[p + "expect/weight"]) and np.array_equal(got["std"], d[p + "expect/std"])
    assert got["zero_point"] == int(d[p + "expect/zero_point"])


def test_from_float_accepts_a_qat_module():
    """Conv2d.from_float on a reference-style QAT module: the reference's own swap seam (quant_utils.py:62-99)."""
    import quantised_bayesian_nets_amd as q
    d, names = _layers()
    p = [n for n in names if n.endswith("layers.4.0.shortcut.0") and n.startswith("w8")][0] + "/"

    def fq(a):
        o = types.SimpleNamespace(min_val=torch.tensor(np.float32(a[0])), max_val=torch.tensor(np.float32(a[1])), quant_min=int(a[2]), quant_max=int(a[3]))
        return types.SimpleNamespace(activation_post_process=o, quant_min=int(a[2]), quant_max=int(a[3]))

    bn = types.SimpleNamespace(running_mean=torch.from_numpy(d[p + "bn.running_mean"]), running_var=torch.from_numpy(d[p + "bn.running_var"]),
                               eps=float(d[p + "bn.eps"]), weight=torch.from_numpy(d[p + "bn.weight"]), bias=torch.from_numpy(d[p + "bn.bias"]))
    mod = types.SimpleNamespace(weight=torch.from_numpy(d[p + "mu"]), std=torch.from_numpy(d[p + "rho"]), bias=None, bn=bn,
                                weight_fake_quant=fq(d[p + "obs.weight"]), std_fake_quant=fq(d[p + "obs.std"]),
                                activation_post_process=fq(d[p + "obs.act"]),
                                add_weight=types.SimpleNamespace(activation_post_process=fq(d[p + "obs.add"])),
                                mul_noise=types.SimpleNamespace(activation_post_process=fq(d[p + "obs.mul"])),
                                in_channels=24, out_channels=48, kernel_size=(1, 1), stride=(2, 2), padding=(0, 0), dilation=(1, 1),
                                groups=1, padding_mode="zeros", args=types.SimpleNamespace(activation_precision=7, weight_precision=8))
    layer = q.Conv2d.from_float(mod)
    assert np.array_equal(layer.weight.int_repr(), d[p + "expect/weight"])
    assert layer.zero_point == int(d[p + "expect/zero_point"]) and abs(layer.scale - float(d[p + "expect/scale"])) == 0
    assert np.array_equal(layer.bias_.numpy(), d[p + "expect/bias_"])


def _prepared_fixture():
    d = np.load(os.path.join(GOLDEN_DIR, "resnet_bbb_prepared_a7w8.npz"))
    return {k[len("state/"):]: d[k] for k in d.files}


def test_model_level_convert_matches_reference():
    """`convert_model_state` (the reference's quant_utils.convert, src/quant_utils.py:62-99, walked over a whole
    prepared state dict) turns the calibrated PREPARED conv_resnet_bbb recorded from the reference into exactly the converted
    state the reference's own convert() produced from that model (the `state/` of resnet_bbb_a7w8.npz): every key, every int8
    weight, every scale / zero point."""
    import types
    from conftest import load_golden
    from quantised_bayesian_nets_amd.convert import convert_model_state, convert_model
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8)
    out = convert_model_state(_prepared_fixture(), args)
    ref = load_golden("resnet_bbb_a7w8.npz")["state"]
    assert sorted(out) == sorted(ref)
    for k, v in ref.items():
        a, b = np.asarray(out[k]).reshape(-1), np.asarray(v).reshape(-1)
        assert np.array_equal(a.astype(np.float64), b.astype(np.float64)), k
    # and the whole pipeline: prepared state -> the package's int8 model, whose layers report the reference's converted state
    m = convert_model(_prepared_fixture(), "conv_resnet_bbb", [1, 3, 32, 32], 10, args)
    for name, layer in zip(m.stochastic_layer_names(), m.stochastic_layers()):
        for k, v in layer.reference_state(name + ".").items():
            assert np.array_equal(np.asarray(v).astype(np.float64), np.asarray(ref[k]).astype(np.float64)), k
    assert m.quant.zero_point == int(ref["quant.zero_point"].reshape(-1)[0])
    # a QAT model of this package hands over the same dict (observers as they stand)
    import quantised_bayesian_nets_amd as q
    aq = types.SimpleNamespace(sigma_prior=-2.0, activation_precision=7, weight_precision=8, qat_eval=True)
    mq = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, aq).load_reference_state(_prepared_fixture())
    out2 = convert_model_state(mq.prepared_state(), args)
    for k, v in ref.items():
        assert np.array_equal(np.asarray(out2[k]).reshape(-1).astype(np.float64), np.asarray(v).reshape(-1).astype(np.float64)), k


def test_checkpoint_reader_reads_the_reference_wire_format(tmp_path):
    """`checkpoint.load_model` reads the file the reference's `utils.save_model` wrote (torch.save of a state
    dict with qint8 tensors, src/utils.py:84-93; fixture recorded from the reference with a `module.` prefix) exactly like
    `utils.load_model` (src/utils.py:112-123), and `save_model` writes a file that round-trips."""
    import types
    import torch
    from conftest import load_golden
    import quantised_bayesian_nets_amd as q
    from quantised_bayesian_nets_amd import checkpoint as ck
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8)
    ref = load_golden("resnet_bbb_a7w8.npz")["state"]
    path = os.path.join(GOLDEN_DIR, "resnet_bbb_a7w8_weights.pt")
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert all(k.startswith("module.") for k in raw) and raw["module.layers.0.weight"].dtype == torch.qint8
    flat = ck.load_state(path)
    for k, v in ref.items():
        assert np.array_equal(np.asarray(flat[k]).reshape(-1).astype(np.float64), np.asarray(v).reshape(-1).astype(np.float64)), k
    m = ck.load_model(q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, args), path)
    for name, layer in zip(m.stochastic_layer_names(), m.stochastic_layers()):
        for k, v in layer.reference_state(name + ".").items():
            assert np.array_equal(np.asarray(v).astype(np.float64), np.asarray(ref[k]).astype(np.float64)), k
    # writer: the same keys / dtypes as the reference's file (without the DataParallel prefix), and it reads back identically
    out = ck.save_model(m, str(tmp_path / "weights.pt"))
    mine = torch.load(out, map_location="cpu", weights_only=False)
    for k, v in raw.items():
        k2 = k.replace("module.", "")
        if v is None:
            continue
        assert k2 in mine, k2
        if v.is_quantized:
            assert mine[k2].is_quantized and torch.equal(mine[k2].int_repr(), v.int_repr()) and mine[k2].q_scale() == v.q_scale() and \
                mine[k2].q_zero_point() == v.q_zero_point(), k2
        else:
            assert np.array_equal(mine[k2].detach().numpy().reshape(-1).astype(np.float64), v.detach().numpy().reshape(-1).astype(np.float64)), k2
    m2 = ck.load_model(q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, args), out)
    assert np.array_equal(m2.layers[9].weight.int_repr(), m.layers[9].weight.int_repr())


def test_load_ensemble_reads_one_file_per_member(golden_ensemble, tmp_path):
    """reference models_sgld.py:245-261: `weights_<n>.pt` per member (fixtures saved by the reference's own modules, keys under
    `main_net.`), natural order, the last args.samples."""
    import shutil
    import types
    import quantised_bayesian_nets_amd as q
    src = os.path.join(GOLDEN_DIR, "ensemble_ckpt")
    for f in os.listdir(src):
        shutil.copy(os.path.join(src, f), tmp_path / f)
    shutil.copy(os.path.join(src, "weights_1.pt"), tmp_path / "weights_0.pt")       # an older sample: must NOT be picked (last 2 of 3)
    (tmp_path / "args.pt").write_bytes(b"")
    args = types.SimpleNamespace(activation_precision=7, weight_precision=8, model="conv_resnet_sgld", samples=2, save=str(tmp_path))
    net = q.ModelFactory.get_model("conv_resnet_sgld", [1, 3, 32, 32], 10, True, args, training_mode=False)
    net.load_ensemble(args)
    assert net.sample_names == ["weights_1.pt", "weights_2.pt"]
    for i, mem in enumerate(net.ensemble):
        g = golden_ensemble["members"][i]
        for name, layer in zip(mem.stochastic_layer_names(), mem.stochastic_layers()):
            assert np.array_equal(layer.weight.int_repr(), g[name + ".weight"]), (i, name)
            assert layer.weight.q_scale() == float(g[name + ".weight.q_scale"]) and layer.scale == float(g[name + ".scale"])
            if (name + ".bias") in g and np.asarray(g[name + ".bias"]).size:
                assert np.array_equal(layer.bias_.numpy(), g[name + ".bias"])
            else:
                assert layer.bias_ is None
        assert mem.quant.scale == float(np.asarray(g["quant.scale"]).reshape(-1)[0])


def test_prepare_model_state_builds_the_reference_prepared_keys():
    """SURVEY 8f row 4, prepare side: `prepare_model_state` (fusion + observer insertion of quant_utils.prepare_model,
    src/quant_utils.py:112-147) turns the float conv_resnet_bbb fixture into exactly the key set the reference's prepared model
    has (713 entries), BatchNorm tensors under `<conv>.bn.*`, every observer fresh."""
    from quantised_bayesian_nets_amd.convert import prepare_model_state
    d = np.load(os.path.join(GOLDEN_DIR, "resnet_bbb_f32.npz"))
    fstate = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
    ref = np.load(os.path.join(GOLDEN_DIR, "resnet_bbb_prepare_calibrate.npz"))
    prep = prepare_model_state(fstate)
    assert sorted(prep) == sorted(str(k) for k in ref["fresh_keys"])
    for k in ref.files:
        if k.startswith("fresh/"):
            assert np.array_equal(np.asarray(prep[k[len("fresh/"):]]), ref[k]), k          # +inf / -inf
    assert np.array_equal(prep["layers.4.0.stem.0.bn.running_var"], fstate["layers.4.0.stem.1.running_var"])
    assert np.array_equal(prep["layers.0.weight"], fstate["layers.0.weight"])
    # the package's QAT model loads it (observers unseen)
    import types
    import quantised_bayesian_nets_amd as q
    aq = types.SimpleNamespace(sigma_prior=-2.0, activation_precision=7, weight_precision=8, qat_eval=True)
    m = q.ModelFactory.get_model("conv_resnet_bbb", [1, 3, 32, 32], 10, True, aq).load_reference_state(prep)
    assert m.layers[0].weight_fake_quant.min_max() == (float("inf"), float("-inf"))


_NON_BBB = [("mlp_mc_qat.npz", "linear_mc"), ("lenet_mc_qat.npz", "conv_lenet_mc"), ("resnet_mc_qat.npz", "conv_resnet_mc"), ("resnet_sgld_qat.npz", "conv_resnet_sgld")]


@pytest.mark.parametrize("name,model", _NON_BBB)
def test_model_level_convert_of_the_non_bbb_graphs_matches_reference(name, model):
    """SURVEY 8f row 4 widened to quant_utils.prepare_model's `prepare_qat` branch (:139-140): `convert_model_state` on the prepared MC-Dropout graphs
    and the SGHMC member template gives exactly what the reference's own quant_utils.convert (:62-99, through torch's from_float) makes of the same
    prepared state -- every key, every int8 tensor bit for bit, every scale (fixtures: tests/golden/make_golden_qat_mc.py, `converted/*`)."""
    import types
    from quantised_bayesian_nets_amd.convert import convert_model_state
    d = np.load(os.path.join(GOLDEN_DIR, name))
    st = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
    want = {k[len("converted/"):]: d[k] for k in d.files if k.startswith("converted/")}
    got = convert_model_state(st, types.SimpleNamespace(activation_precision=7, weight_precision=8))
    assert sorted(got) == sorted(want)
    n_int8 = 0
    for k, w in want.items():
        g = np.asarray(got[k])
        if w.dtype == np.int8:
            assert g.dtype == np.int8 and np.array_equal(w, g), k
            n_int8 += 1
        else:
            np.testing.assert_allclose(g.astype(np.float64).reshape(-1), np.asarray(w, np.float64).reshape(-1), rtol=1e-7, atol=0, err_msg=k)
    assert n_int8 >= 4


@pytest.mark.parametrize("fname,qname", [("mlp_mc_f32.npz", "mlp_mc_qat.npz"), ("lenet_mc_f32.npz", "lenet_mc_qat.npz"), ("resnet_mc_f32.npz", "resnet_mc_qat.npz")])
def test_prepare_model_state_of_the_mc_dropout_graphs_builds_the_reference_prepared_keys(fname, qname):
    """The prepare side of the same branch: the float MC-Dropout state -> the key set of the reference's prepared model (conv + BatchNorm fused:
    `<conv>.bn.*`; a weight and an output FakeQuantize per layer; two per BernoulliDropout; QuantStub; one per Add), every observer fresh.
    The reference's keys are those of the prepared states recorded in the *_mc_qat.npz fixtures (floating-point entries, as flat() keeps them)."""
    from quantised_bayesian_nets_amd.convert import prepare_model_state
    d = np.load(os.path.join(GOLDEN_DIR, fname))
    fstate = {k[len("state/"):]: d[k] for k in d.files if k.startswith("state/")}
    ref = np.load(os.path.join(GOLDEN_DIR, qname))
    ref_keys = sorted(k[len("state/"):] for k in ref.files if k.startswith("state/"))
    prep = prepare_model_state(fstate)
    mine = sorted(k for k, v in prep.items() if np.asarray(v).dtype.kind == "f" and not k.endswith(".scale"))
    assert mine == ref_keys
    for k in mine:
        if k.endswith("min_val"):
            assert np.isposinf(prep[k]) and np.isneginf(prep[k.replace("min_val", "max_val")])
