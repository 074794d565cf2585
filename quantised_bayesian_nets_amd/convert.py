"""QAT -> int8 conversion of Bayes-by-backprop layers, restated natively.

What the reference does at `quant_utils.convert` time for one stochastic layer (conv_q.py:127-177, :214-225,
linear_q.py:105-145, bbb/conv.py:70-80), expressed on plain tensors:

  1. ConvBn* only -- fold BatchNorm into the distribution parameters:
        c = gamma * rsqrt(running_var + eps);  mu' = mu * c;  rho' = softplusinv(softplus(rho) * c);
        bias' = (bias - running_mean) * rsqrt(running_var + eps) * gamma + beta
  2. one more observer step (MovingAverageMinMax, c = 0.01) of the weight / std FakeQuantize on mu' / softplus(rho'),
  3. per-tensor-affine qparams from the observers (min/max widened to include 0; zero_point = qmin - round(min/scale)),
  4. torch.quantize_per_tensor(mu', s_w, z_w, qint8), same for softplus(rho'),
  5. output (scale, zero_point) from the activation observer; add_weight / mul_noise (scale, zero_point) from theirs.

Host-side only (runs once per model, on CPU tensors); torch is used for the elementwise float ops so that softplus /
rsqrt / log / exp round exactly as in the reference's own conversion.  Validated bit-for-bit against the reference's
convert() on real layers: tests/golden/make_golden_convert.py -> tests/test_convert.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .layers import QFunctional, QuantizedParam

AVERAGING_CONSTANT = 0.01      # torch MovingAverageMinMaxObserver default, used by the reference's qconfig (quant_utils.py:129-138)
_EPS = torch.finfo(torch.float32).eps


def softplusinv(x):
    """reference bbb/utils_bbb.py:7-8."""
    return torch.log(torch.exp(x) - 1.)


def fold_bn(mu, rho, bias, bn_rm, bn_rv, bn_eps, bn_w, bn_b):
    """reference bbb/conv.py:70-80 (fuse_conv_bn_weights)."""
    if bias is None:
        bias = torch.zeros_like(bn_rm)
    rs = torch.rsqrt(bn_rv + bn_eps)
    c = (bn_w * rs).reshape([-1] + [1] * (mu.dim() - 1))
    return mu * c, softplusinv(F.softplus(rho) * c), (bias - bn_rm) * rs * bn_w + bn_b


class ObserverState:
    """min/max state of a MovingAverageMinMaxObserver inside a FakeQuantize (per-tensor affine)."""

    def __init__(self, min_val, max_val, quant_min, quant_max):
        self.min_val = torch.as_tensor(min_val, dtype=torch.float32).reshape(())
        self.max_val = torch.as_tensor(max_val, dtype=torch.float32).reshape(())
        self.quant_min, self.quant_max = int(quant_min), int(quant_max)

    @classmethod
    def from_fake_quant(cls, fq):
        """Duck-typed read of a torch.quantization.FakeQuantize (or a bare observer) instance."""
        obs = getattr(fq, "activation_post_process", fq)
        return cls(obs.min_val.detach().clone(), obs.max_val.detach().clone(), getattr(fq, "quant_min", obs.quant_min),
                   getattr(fq, "quant_max", obs.quant_max))

    def observe(self, x):
        mn, mx = torch.aminmax(x.detach().to(torch.float32))
        if self.min_val == float("inf") and self.max_val == float("-inf"):
            self.min_val, self.max_val = mn, mx
        else:
            self.min_val = self.min_val + AVERAGING_CONSTANT * (mn - self.min_val)
            self.max_val = self.max_val + AVERAGING_CONSTANT * (mx - self.max_val)
        return self

    def qparams(self):
        mn = torch.min(self.min_val, torch.zeros_like(self.min_val))
        mx = torch.max(self.max_val, torch.zeros_like(self.max_val))
        scale = (mx - mn) / float(self.quant_max - self.quant_min)
        scale = torch.max(scale, torch.tensor(_EPS))
        zp = self.quant_min - torch.round(mn / scale).to(torch.int)
        zp = torch.clamp(zp, self.quant_min, self.quant_max)
        return float(scale), int(zp)


def quantize_qint8(x, scale, zero_point):
    """torch.quantize_per_tensor(x, scale, zp, torch.qint8).int_repr(): clamp(rne(x * (1/scale)) + zp, -128, 127)."""
    return torch.quantize_per_tensor(x.to(torch.float32), float(scale), int(zero_point), torch.qint8).int_repr().numpy()


def convert_layer(mu, rho, bias, w_obs, std_obs, act_obs, add_obs, mul_obs, bn=None):
    """One QAT BBB layer -> the converted int8 layer's state (reference key names, flat numpy).
    bn: None or dict(running_mean, running_var, eps, weight, bias)."""
    mu, rho = torch.as_tensor(mu, dtype=torch.float32), torch.as_tensor(rho, dtype=torch.float32)
    bias = None if bias is None else torch.as_tensor(bias, dtype=torch.float32)
    if bn is not None:
        mu, rho, bias = fold_bn(mu, rho, bias, torch.as_tensor(bn["running_mean"]), torch.as_tensor(bn["running_var"]), float(bn["eps"]),
                                torch.as_tensor(bn["weight"]), torch.as_tensor(bn["bias"]))
    sigma = F.softplus(rho)
    s_w, z_w = w_obs.observe(mu).qparams()
    s_s, z_s = std_obs.observe(sigma).qparams()
    s_y, z_y = act_obs.qparams()
    s_a, z_a = add_obs.qparams()
    s_m, z_m = mul_obs.qparams()
    out = {"weight": quantize_qint8(mu, s_w, z_w), "weight.q_scale": np.float64(s_w), "weight.q_zero_point": np.int64(z_w),
           "std": quantize_qint8(sigma, s_s, z_s), "std.q_scale": np.float64(s_s), "std.q_zero_point": np.int64(z_s),
           "scale": np.float32(s_y), "zero_point": np.int64(z_y),
           "add_weight.scale": np.float32(s_a), "add_weight.zero_point": np.int64(z_a),
           "mul_noise.scale": np.float32(s_m), "mul_noise.zero_point": np.int64(z_m)}
    if bias is not None:
        out["bias_"] = bias.detach().numpy()
    return out


def convert_qat_module(mod):
    """Duck-typed entry for a reference QAT module instance (conv_qat.Conv2d / ConvBn2d / ConvReLU2d / ConvBnReLU2d,
    linear_qat.Linear / LinearReLU): reads its tensors and observer states, returns the converted layer state."""
    bn = None
    if hasattr(mod, "bn"):
        b = mod.bn
        bn = dict(running_mean=b.running_mean.detach(), running_var=b.running_var.detach(), eps=b.eps, weight=b.weight.detach(), bias=b.bias.detach())
    bias = None if getattr(mod, "bias", None) is None else mod.bias.detach()
    return convert_layer(mod.weight.detach(), mod.std.detach(), bias,
                         ObserverState.from_fake_quant(mod.weight_fake_quant), ObserverState.from_fake_quant(mod.std_fake_quant),
                         ObserverState.from_fake_quant(mod.activation_post_process),
                         ObserverState.from_fake_quant(mod.add_weight.activation_post_process),
                         ObserverState.from_fake_quant(mod.mul_noise.activation_post_process), bn)


# ---------------------------------------------------------------------------------------------- model level
_OBS = ".activation_post_process."


def _obs(state, key, bounds):
    """ObserverState of the FakeQuantize stored under `key` (…activation_post_process.{min_val,max_val}) in a prepared state dict."""
    return ObserverState(np.asarray(state[key + "min_val"]).reshape(()), np.asarray(state[key + "max_val"]).reshape(()), bounds[0], bounds[1])


def convert_model_state(prepared, args, bn_eps=1e-5):
    """The reference's model-level `quant_utils.convert` (src/quant_utils.py:62-99) on a flat state dict.

    `prepared`: flat {key: numpy} state of a PREPARED (QAT) BBB model -- the keys of the reference model's state_dict after
    `prepare_model` (quant_utils.py:112-147) and calibration: per stochastic layer `<p>.weight` (mu), `<p>.std` (rho),
    optional `<p>.bias`, `<p>.bn.{weight,bias,running_mean,running_var}` for ConvBn*, and the min / max of its five
    FakeQuantize observers; `quant.activation_post_process...`; one `...add.add.activation_post_process...` per BasicBlock.
    (models_qat.py holds the same dict, with the observers as they stand after its live-observer evaluations.)
    The walk is what `convert` does: every swappable module is replaced through its class's `from_float`
    (conv_q.py:127-177, linear_q.py:105-145), QuantStub -> Quantize and FloatFunctional -> QFunctional take their observer's
    qparams.  The quantisation ranges come from `args` exactly as in prepare_model (:122-123): weight-like observers (weight,
    std, add_weight, mul_noise) use INT_BOUNDS[weight_precision], activations UINT_BOUNDS[activation_precision].
    Returns the converted model's flat state (reference key names), i.e. what `load_reference_state` ingests and what the
    reference would have saved with `utils.save_model` after `postprocess_model` (quant_utils.py:100-109)."""
    from .quant import INT_BOUNDS, UINT_BOUNDS
    wb, ab = INT_BOUNDS[args.weight_precision], UINT_BOUNDS[args.activation_precision]
    out = {}
    tag = ".weight_fake_quant" + _OBS + "min_val"
    layers = [k[:-len(tag)] for k in prepared if k.endswith(tag)]
    if layers and not any((p + ".std") in prepared for p in layers):
        return _convert_deterministic_state(prepared, layers, wb, ab, bn_eps)
    for p in layers:
        bn = None
        if (p + ".bn.running_mean") in prepared:
            bn = dict(running_mean=prepared[p + ".bn.running_mean"], running_var=prepared[p + ".bn.running_var"], eps=bn_eps,
                      weight=prepared[p + ".bn.weight"], bias=prepared[p + ".bn.bias"])
        st = convert_layer(prepared[p + ".weight"], prepared[p + ".std"], prepared.get(p + ".bias"),
                           _obs(prepared, p + ".weight_fake_quant" + _OBS, wb), _obs(prepared, p + ".std_fake_quant" + _OBS, wb),
                           _obs(prepared, p + _OBS + "activation_post_process.", ab),
                           _obs(prepared, p + ".add_weight" + _OBS + "activation_post_process.", wb),
                           _obs(prepared, p + ".mul_noise" + _OBS + "activation_post_process.", wb), bn)
        out.update({p + "." + k: v for k, v in st.items()})
        if (p + ".std_prior") in prepared:
            out[p + ".std_prior"] = np.asarray(prepared[p + ".std_prior"])
    # QuantStub -> Quantize, Add's FloatFunctional -> QFunctional: (scale, zero_point) of their activation observer
    for k in prepared:
        if k.endswith(_OBS + "activation_post_process.min_val") and not any(k.startswith(p + ".") for p in layers):
            base = k[:-len(_OBS + "activation_post_process.min_val")]
            s, z = _obs(prepared, base + _OBS + "activation_post_process.", ab).qparams()
            if base == "quant":
                out["quant.scale"], out["quant.zero_point"] = np.asarray([s], np.float32), np.asarray([z], np.int64)
            else:
                out[base + ".scale"], out[base + ".zero_point"] = np.float32(s), np.int64(z)
    return out


def convert_deterministic_layer(w, bias, w_obs, act_obs, bn=None):
    """One torch.ao.nn.qat layer (Linear / LinearReLU / Conv2d / ConvBn2d / ConvBnReLU2d: deterministic weights) -> the converted int8 layer's
    state, as torch's own `from_float` does it (torch.ao.nn.quantized.modules.conv._ConvNd.from_float / linear.Linear.from_float, reached through
    the reference's quant_utils.convert :62-99): fold the BatchNorm (fuse_conv_bn_weights), ONE more step of the weight observer on the folded
    weight, per-tensor-affine qint8 with the integers clamped to the observer's [quant_min, quant_max] (a no-op at 8 bits), output
    (scale, zero point) from the activation observer."""
    w = torch.as_tensor(w, dtype=torch.float32)
    bias = None if bias is None else torch.as_tensor(bias, dtype=torch.float32)
    if bn is not None:
        rm, rv = torch.as_tensor(bn["running_mean"]), torch.as_tensor(bn["running_var"])
        if bias is None:
            bias = torch.zeros_like(rm)
        rs = torch.rsqrt(rv + float(bn["eps"]))
        w = w * (torch.as_tensor(bn["weight"]) * rs).reshape([-1] + [1] * (w.dim() - 1))
        bias = (bias - rm) * rs * torch.as_tensor(bn["weight"]) + torch.as_tensor(bn["bias"])
    s_w, z_w = w_obs.observe(w).qparams()
    s_y, z_y = act_obs.qparams()
    out = {"weight": np.clip(quantize_qint8(w, s_w, z_w), w_obs.quant_min, w_obs.quant_max).astype(np.int8), "weight.q_scale": np.float64(s_w),
           "weight.q_zero_point": np.int64(z_w), "scale": np.float32(s_y), "zero_point": np.int64(z_y)}
    if bias is not None:
        out["bias"] = bias.detach().numpy()
    return out


def _convert_deterministic_state(prepared, layers, wb, ab, bn_eps):
    """convert_model_state for the models quant_utils.prepare_model prepared with `prepare_qat` (:139-140): the MC-Dropout graphs and the SGHMC
    member templates.  Besides the weighted layers: QuantStub -> Quantize, Add's FloatFunctional -> QFunctional, and each BernoulliDropout's two
    FloatFunctionals (mcdropout/dropout.py:9-13) -> QFunctional -- `mul_mask` with its observer's qparams, `mul_scalar` (never observed:
    FloatFunctional.mul_scalar has no observer call) with the (1.0, 0) torch gives an observer that saw nothing; `p` / `multiplier` pass through."""
    out = {}
    for p in layers:
        bn = None
        if (p + ".bn.running_mean") in prepared:
            bn = dict(running_mean=prepared[p + ".bn.running_mean"], running_var=prepared[p + ".bn.running_var"], eps=bn_eps,
                      weight=prepared[p + ".bn.weight"], bias=prepared[p + ".bn.bias"])
        st = convert_deterministic_layer(prepared[p + ".weight"], prepared.get(p + ".bias"), _obs(prepared, p + ".weight_fake_quant" + _OBS, wb),
                                         _obs(prepared, p + _OBS + "activation_post_process.", ab), bn)
        out.update({p + "." + k: v for k, v in st.items()})
    for k in prepared:
        if k.endswith(_OBS + "activation_post_process.min_val") and not any(k.startswith(p + ".") for p in layers):
            base = k[:-len(_OBS + "activation_post_process.min_val")]
            seen = np.isfinite(float(np.asarray(prepared[k])))
            s, z = _obs(prepared, base + _OBS + "activation_post_process.", ab).qparams() if seen else (1.0, 0)
            if base.endswith("quant") and not base.endswith("fake_quant"):
                out[base + ".scale"], out[base + ".zero_point"] = np.asarray([s], np.float32), np.asarray([z], np.int64)
            else:
                out[base + ".scale"], out[base + ".zero_point"] = np.float32(s), np.int64(z)
        elif k.endswith(".p") or k.endswith(".multiplier"):
            out[k] = np.asarray(prepared[k], np.float32).reshape(1)
    return out


def convert_model(prepared, model_name, input_size, output_size, args):
    """prepared state (or a models_qat model holding one) -> the converted int8 model of this package, ready for the HIP path:
    `ModelFactory.get_model(model_name, ..., q=True, args)` loaded with `convert_model_state(prepared, args)`."""
    from .models import ModelFactory
    import types
    if hasattr(prepared, "prepared_state"):
        prepared = prepared.prepared_state()
    a = types.SimpleNamespace(**{k: v for k, v in vars(args).items() if k != "qat_eval"})
    if "sgld" in model_name:
        # the SGHMC template (`main_net.`) converts to ONE ensemble member (models_sgld.py:245-261 loads such states member by member)
        member = {k[len("main_net."):] if k.startswith("main_net.") else k: v for k, v in convert_model_state(prepared, a).items()}
        a.samples = 1
        return ModelFactory.get_model(model_name, input_size, output_size, True, a, training_mode=False).load_reference_state([member])
    model = ModelFactory.get_model(model_name, input_size, output_size, True, a)
    return model.load_reference_state(convert_model_state(prepared, a))


# ---------------------------------------------------------------------------------------------- prepare
def prepare_model_state(float_state):
    """The state-dict side of the reference's `quant_utils.prepare_model` (src/quant_utils.py:112-147) for a BBB model:

      * fusion (`model.fuse_model()` -> `fuse_bbb_modules`, models_bbb.py:10-29, :186-188, :249): a stochastic conv followed by its
        BatchNorm becomes one ConvBn(ReLU)2d module at the conv's index -- the BatchNorm's tensors move from `<seq>.<i+1>.*` to
        `<seq>.<i>.bn.*` (the later indices become Identity);
      * observer insertion (`prepare` x2 + `convert(model, QAT_MODULE_MAPPINGS)`): every stochastic layer gets the five FakeQuantize
        of conv_qat.py:21-23, :75-77 (weight, std, mul_noise, add_weight, output), the QuantStub and every BasicBlock's `Add` one
        activation FakeQuantize each -- all fresh (min = +inf, max = -inf, scale 1, zero point 0).

    `float_state`: flat {key: numpy} state of the float model (`weight` = mu, `std` = rho, BatchNorm parameters / statistics).
    Returns the prepared model's flat state: what `models_qat` loads (live-observer evaluation = calibration) and what
    `convert_model_state` converts."""
    out = {}
    inf = np.float32(np.inf)
    eps = np.asarray([np.finfo(np.float32).eps], np.float32)

    def observer(prefix):           # a FakeQuantize holding a fresh MovingAverageMinMaxObserver
        out[prefix + ".scale"] = np.asarray([1.0], np.float32)
        out[prefix + ".zero_point"] = np.asarray([0], np.int32)
        out[prefix + ".activation_post_process.eps"] = eps
        out[prefix + ".activation_post_process.min_val"] = inf
        out[prefix + ".activation_post_process.max_val"] = -inf

    layers = [k[:-len(".std")] for k in float_state if k.endswith(".std")]
    bbb = bool(layers)
    if not bbb:
        # the `prepare_qat` branch (quant_utils.py:139-140): every nn.Linear / nn.Conv2d (a weight of 2 or 4 dimensions) becomes a
        # torch.ao.nn.qat layer with a weight and an output FakeQuantize; a BernoulliDropout (`<d>.p`) gets one per FloatFunctional
        layers = [k[:-len(".weight")] for k, v in float_state.items() if k.endswith(".weight") and np.asarray(v).ndim in (2, 4)]
    bn_of = {}
    for p in layers:
        head, _, idx = p.rpartition(".")
        if idx.isdigit():
            cand = f"{head}.{int(idx) + 1}"
            if (cand + ".running_mean") in float_state:
                bn_of[cand] = p
    for k, v in float_state.items():
        moved = False
        for bn, conv in bn_of.items():
            if k.startswith(bn + "."):
                out[conv + ".bn." + k[len(bn) + 1:]] = np.asarray(v)        # incl. num_batches_tracked, as the reference's fused module keeps it
                moved = True
                break
        if not moved:
            out[k] = np.asarray(v)
    for p in layers:
        for fq in (("weight_fake_quant", "std_fake_quant", "activation_post_process", "add_weight.activation_post_process",
                    "mul_noise.activation_post_process") if bbb else ("weight_fake_quant", "activation_post_process")):
            observer(p + "." + fq)
    for k in float_state:
        if not bbb and k.endswith(".multiplier") and (k[:-len("multiplier")] + "p") in float_state:
            observer(k[:-len(".multiplier")] + ".mul_mask.activation_post_process")
            observer(k[:-len(".multiplier")] + ".mul_scalar.activation_post_process")
    root = "main_net." if any(k.startswith("main_net.") for k in float_state) else ""
    observer(root + "quant.activation_post_process")
    for p in layers:                                        # one Add per BasicBlock: the blocks are the parents of `<blk>.stem.0`
        if p.endswith(".stem.0"):
            observer(p[:-len(".stem.0")] + ".add.add.activation_post_process")
    return out


def calibrate(qat_model, batches, seed=0):
    """Calibration as the reference's evaluation-mode forwards do it: every forward of the prepared model updates every
    observer (they are never disabled: SURVEY 3.3).  Runs one stochastic forward per batch on the GPU (`models_qat`, live
    observers); afterwards `convert_model_state(qat_model.prepared_state(), args)` / `convert_model(qat_model, ...)` gives the
    int8 model.  (The reference's TRAIN-mode forwards -- local reparametrisation, BatchNorm batch statistics -- are training and
    out of scope; a model calibrated there arrives as a prepared state dict instead.)"""
    from .layers import mc_context
    for i, x in enumerate(batches):
        with mc_context(1, seed, i):
            qat_model.forward_mc(x)
    return qat_model
