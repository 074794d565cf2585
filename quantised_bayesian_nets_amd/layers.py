"""int8 Bayes-by-backprop layers behind the reference's layer API.

Host-side mirror of reference src/models/stochastic/bbb/quantized/conv_q.py (Conv2d :22-177, ConvReLU2d :179-225)
and linear_q.py (Linear :13-145, LinearReLU :147-185): same constructor arguments, attribute names
(`weight`, `std`, `bias_`, `scale`, `zero_point`, `add_weight`, `mul_noise`, `std_prior`, `args`) and state-dict
keys.  The arithmetic runs in libqbnn_hip.so (HIP, gfx950); there is no CPU path.

Difference by design: a forward call processes S Monte-Carlo samples at once.  Activations travel as
`MCQTensor` ([S, B, H, W, C] uint8, NHWC) instead of one torch quint8 tensor per sample, and the weight noise comes
from the Philox stream (seed, layer_id, sample) instead of torch's global generator.
"""
import contextlib
import os
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .quant import INT_BOUNDS, UINT_BOUNDS, make_sample_params

LAYOUT_MFMA32, LAYOUT_ROWMAJOR, LAYOUT_MFMA32_N24, LAYOUT_MFMA32_TAIL, LAYOUT_MFMA32_N24_TAIL = 0, 1, 2, 3, 4      # include/qbnn.h: QBNN_LAYOUT_*


def w16_enabled():
    """The 16-wave layer-1 kernel (csrc/qbnn_w16.hip) is in use: its blocks' weights are packed as MFMA32_TAIL fragments (QBNN_W16=0: the 8-wave
    kernels of round 2 on MFMA32 weights, for A/B checks)."""
    return os.environ.get("QBNN_W16", "1") != "0"

# State epoch: bumped whenever device-side parameter images may be dropped or replaced (a state dict loaded into any layer, a packed
# layout switched).  A captured HIP graph holds raw pointers to those images; mc.GraphedPredictor records the epoch at capture and
# captures again when it has moved (a replay over freed / repacked buffers would be silently wrong).
_STATE_EPOCH = [0]


def bump_state_epoch():
    _STATE_EPOCH[0] += 1


def state_epoch():
    return _STATE_EPOCH[0]

# bench.py sets this to a list: kernel launches then append (key, meta, start_event, end_event), HIP events recorded on the
# launch stream (torch's current stream).  PROFILE_FILTER (a predicate on meta) limits the events to the launches of
# interest, and back-to-back profiled launches share one event (end of one = start of the next): every event is a marker
# packet that drains the queue, ~4 us each, which at 24 events per 4 ms step was 2.5 % of the measured step.
PROFILE = None
PROFILE_FILTER = None
_prev_end = None


@contextlib.contextmanager
def timed(key, meta=None):
    global _prev_end
    if PROFILE is None or (PROFILE_FILTER is not None and not PROFILE_FILTER(meta)):
        _prev_end = None                 # another launch goes between: the next profiled one needs its own start event
        yield
        return
    e0 = _prev_end
    if e0 is None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    yield
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    _prev_end = e1
    PROFILE.append((key, meta, e0, e1))


# ------------------------------------------------------------------ MC context
class _MCState:
    samples, seed, sample_begin, eps = 1, 0, 0, None


_MC = _MCState()


@contextlib.contextmanager
def mc_context(samples, seed, sample_begin=0, eps=None):
    """Number of MC samples a layer call evaluates, and the Philox (seed, first global sample index).
    `eps`: optional {layer_id: fp32 tensor [S, n_weights] in OHWI order} -- parity mode (injected noise)."""
    old = (_MC.samples, _MC.seed, _MC.sample_begin, _MC.eps)
    _MC.samples, _MC.seed, _MC.sample_begin, _MC.eps = int(samples), int(seed), int(sample_begin), eps
    try:
        yield
    finally:
        _MC.samples, _MC.seed, _MC.sample_begin, _MC.eps = old


class MCQTensor:
    """Quantised activations of S MC samples: `data` uint8 [S or 1, B, ...] channels-last, per-tensor (scale, zp).
    S == 1 with shared=True means "identical for every sample" (the network input)."""

    def __init__(self, data, scale, zero_point, shared=False):
        assert data.dtype == torch.uint8
        self.data, self.scale, self.zero_point, self.shared = data, float(scale), int(zero_point), bool(shared)

    @property
    def samples(self):
        return self.data.shape[0]

    def sample_stride(self):
        return 0 if self.shared else self.data[0].numel()

    def q_scale(self):
        return self.scale

    def q_zero_point(self):
        return self.zero_point

    def int_repr(self):
        return self.data

    def dequantize(self):
        return (self.data.to(torch.float32) - float(self.zero_point)) * np.float32(self.scale)


# ---- the reference's own tensor type at the layer seam ---------------------------------------------------------------
def is_torch_quantized(x):
    return isinstance(x, torch.Tensor) and x.is_quantized


def mcq_from_torch(x):
    """torch quint8 tensor as the reference's layers pass it (NCHW-logical `(N, C, H, W)`, or `(N, F)`; conv_q.py:107-125,
    linear_q.py:80-94) -> MCQTensor of ONE sample on the GPU ([1, N, H, W, C] / [1, N, F] uint8, channels-last).
    A CPU tensor (where the reference keeps its int8 model) is uploaded: the arithmetic itself only exists on the MI355X."""
    if x.dtype != torch.quint8:
        raise TypeError("expected a quint8 activation tensor")
    if x.qscheme() not in (torch.per_tensor_affine, torch.per_tensor_symmetric):
        raise RuntimeError("Unsupported qscheme: activations are quantised per tensor")
    if not torch.cuda.is_available():
        raise RuntimeError("qbnn layers run on an MI355X only (no CPU fallback)")
    q = x.int_repr().to("cuda")
    if q.dim() == 4:
        q = q.permute(0, 2, 3, 1)
    elif q.dim() != 2:
        raise ValueError("Input shape must be `(N, C, H, W)` or `(N, F)`!")
    return MCQTensor(q.contiguous().unsqueeze(0), x.q_scale(), x.q_zero_point())


def torch_from_mcq(y, device):
    """MCQTensor holding one sample -> torch quint8 tensor in the reference's layout, on `device` (where the input lived)."""
    if y.data.shape[0] != 1:
        raise RuntimeError("a torch quantised tensor carries one MC sample: call the layer under mc_context(samples=1)")
    q = y.data[0]
    if q.dim() == 4:
        q = q.permute(0, 3, 1, 2)
    return torch._make_per_tensor_quantized_tensor(q.contiguous().to(device), y.scale, y.zero_point)


class QFunctional:
    """Stand-in for torch.nn.quantized.QFunctional: only the (scale, zero_point) the reference reads."""

    def __init__(self, scale=1.0, zero_point=0):
        self.scale, self.zero_point = float(scale), int(zero_point)


class QuantizedParam:
    """A per-tensor-affine qint8 parameter (reference: torch qint8 tensor): integer values + (scale, zp)."""

    def __init__(self, int_repr, scale, zero_point):
        self._int = np.ascontiguousarray(int_repr, dtype=np.int8)
        self._scale, self._zp = float(scale), int(zero_point)

    def int_repr(self):
        return self._int

    def q_scale(self):
        return self._scale

    def q_zero_point(self):
        return self._zp

    @property
    def shape(self):
        return self._int.shape


def _as_qparam(t):
    if isinstance(t, QuantizedParam):
        return t
    if isinstance(t, torch.Tensor) and t.is_quantized:      # a reference module's qint8 tensor
        return QuantizedParam(t.int_repr().cpu().numpy(), t.q_scale(), t.q_zero_point())
    raise TypeError("expected a quantised parameter")


def _stream():
    return _lib.current_stream()


# ------------------------------------------------------------------ base class
class _BBBInt8(nn.Module):
    layout = LAYOUT_MFMA32

    def _init_common(self, w_shape, bias, args):
        self.weight = QuantizedParam(np.zeros(w_shape, np.int8), 1.0, 0)
        self.std = QuantizedParam(np.zeros(w_shape, np.int8), 1.0, 0)
        self.bias_ = torch.zeros(w_shape[0], dtype=torch.float) if bias else None
        self.std_prior = nn.Parameter(torch.ones((1,)), requires_grad=False)
        self.scale, self.zero_point = 1.0, 0
        self.add_weight, self.mul_noise = QFunctional(), QFunctional()
        self.args = args
        self.layer_id = 0
        self._packed = None
        # deterministic = a standard torch.nn.quantized Conv2d / Linear (ensemble members, reference sgld/models_sgld.py):
        # the fixed qint8 weight is used as the "sampled" weight of the single sample, its qparams as add_weight's
        self.deterministic = False

    def bias(self):
        return self.bias_

    # ---- state (reference keys: conv_q.py:72-99, linear_q.py:40-67)
    def load_reference_state(self, state, prefix):
        g = lambda k: state[prefix + k]
        self.weight = QuantizedParam(g("weight"), g("weight.q_scale"), g("weight.q_zero_point"))
        if self.deterministic:
            self.std = QuantizedParam(np.zeros(self.weight.shape, np.int8), 1.0, 0)
            b = state.get(prefix + "bias", None)
            self.bias_ = None if b is None or np.asarray(b).size == 0 else torch.from_numpy(np.asarray(b, np.float32).copy())
            self.scale, self.zero_point = float(g("scale")), int(g("zero_point"))
            self.add_weight = QFunctional(self.weight.q_scale(), self.weight.q_zero_point())
            self.mul_noise = QFunctional()
            self._packed = None
            bump_state_epoch()
            return self
        self.std = QuantizedParam(g("std"), g("std.q_scale"), g("std.q_zero_point"))
        b = state.get(prefix + "bias_", None)
        self.bias_ = None if b is None or np.asarray(b).size == 0 else torch.from_numpy(np.asarray(b, np.float32).copy())
        self.scale, self.zero_point = float(g("scale")), int(g("zero_point"))
        self.add_weight = QFunctional(g("add_weight.scale"), g("add_weight.zero_point"))
        self.mul_noise = QFunctional(g("mul_noise.scale"), g("mul_noise.zero_point"))
        if (prefix + "std_prior") in state:          # carried through conversion untouched (conv_q.py:160, linear_q.py:131)
            self.std_prior.data = torch.from_numpy(np.asarray(state[prefix + "std_prior"], np.float32).reshape(-1).copy())
        self._packed = None
        bump_state_epoch()
        return self

    def reference_state(self, prefix=""):
        d = {prefix + "weight": self.weight.int_repr(), prefix + "weight.q_scale": np.float64(self.weight.q_scale()),
             prefix + "weight.q_zero_point": np.int64(self.weight.q_zero_point()),
             prefix + "std": self.std.int_repr(), prefix + "std.q_scale": np.float64(self.std.q_scale()),
             prefix + "std.q_zero_point": np.int64(self.std.q_zero_point()),
             prefix + "scale": np.float32(self.scale), prefix + "zero_point": np.int64(self.zero_point),
             prefix + "add_weight.scale": np.float32(self.add_weight.scale),
             prefix + "add_weight.zero_point": np.int64(self.add_weight.zero_point),
             prefix + "mul_noise.scale": np.float32(self.mul_noise.scale),
             prefix + "mul_noise.zero_point": np.int64(self.mul_noise.zero_point)}
        if self.bias_ is not None:
            d[prefix + "bias_"] = self.bias_.cpu().numpy()
        return d

    @classmethod
    def _from_converted(cls, mod, *ctor):
        q = cls(*ctor, args=getattr(mod, "args", None))
        q.weight, q.std = _as_qparam(mod.weight), _as_qparam(mod.std)
        b = mod.bias() if callable(getattr(mod, "bias", None)) else getattr(mod, "bias_", None)
        q.bias_ = None if b is None else b.detach().float().cpu().clone()
        q.scale, q.zero_point = float(mod.scale), int(mod.zero_point)
        q.add_weight = QFunctional(mod.add_weight.scale, mod.add_weight.zero_point)
        q.mul_noise = QFunctional(mod.mul_noise.scale, mod.mul_noise.zero_point)
        return q

    # ---- device-side parameter images
    def _logical_ohwi(self, qp):
        w = qp.int_repr()
        return np.ascontiguousarray(w.transpose(0, 2, 3, 1)) if w.ndim == 4 else w

    def _ensure_packed(self, device):
        if self._packed is not None and self._packed["device"] == device:
            return self._packed
        L = _lib.lib()
        mu, sg = self._logical_ohwi(self.weight), self._logical_ohwi(self.std)
        cout, k = mu.shape[0], int(np.prod(mu.shape[1:]))
        krow = self._krow(mu)
        nbytes = L.qbnn_packed_weight_bytes(cout, k, krow, self.layout)
        out = []
        for src in (mu, sg):
            dst = np.zeros(nbytes, np.int8)
            src = np.ascontiguousarray(src.reshape(cout, k))
            _lib.check(L.qbnn_pack_weights_host(src.ctypes.data_as(C.c_void_p), cout, k, krow, self.layout, dst.ctypes.data_as(C.c_void_p)))
            out.append(torch.from_numpy(dst).to(device))
        wp = getattr(self.args, "weight_precision", 8)
        sp = make_sample_params(self.weight.q_scale(), self.weight.q_zero_point(), self.std.q_scale(), self.std.q_zero_point(),
                                self.mul_noise.scale, self.mul_noise.zero_point, self.add_weight.scale,
                                self.add_weight.zero_point, wp)
        bias = None if self.bias_ is None else self.bias_.to(device=device, dtype=torch.float32).contiguous()
        self._packed = dict(device=device, mu=out[0], sigma=out[1], cout=cout, k=k, krow=krow, nbytes=int(nbytes), sp=sp, bias=bias)
        return self._packed

    def set_layout(self, layout, krow=None):
        """Packed layout of this layer's mu / sigma / sampled weights (the small networks switch between the row-major form of the
        any-geometry kernel and the fragment form of their own kernels); a change drops the packed copies."""
        if layout != self.layout or krow != getattr(self, "krow_override", None):
            self.layout, self.krow_override = layout, krow
            self._packed, self._presampled = None, None
            bump_state_epoch()

    def _krow(self, w_ohwi):
        """Bytes of one kernel row (kw, c) in the OHWI weight: the unit the packed K axis is padded by.  The 3-channel
        first layer runs as a 1x1 conv over 27-tap im2col patches, so its whole K is one row."""
        if getattr(self, "krow_override", None):
            return int(self.krow_override)
        if w_ohwi.ndim == 4 and w_ohwi.shape[3] % 8 == 0:
            return int(w_ohwi.shape[2] * w_ohwi.shape[3])
        return int(np.prod(w_ohwi.shape[1:]))

    def sample_weights(self, device, samples=None, seed=None, sample_begin=None, eps=None):
        """W_q for S samples in this layer's packed layout: int8 [S, nbytes]  (conv_q.py:113-119 chain, fused)."""
        if self.deterministic:
            pk = self._ensure_packed(device)
            if (_MC.samples if samples is None else samples) != 1:
                raise RuntimeError("a deterministic (ensemble member) layer evaluates exactly one sample per call")
            return pk["mu"].reshape(1, -1)
        if samples is None and eps is None and _MC.eps is None and getattr(self, "_presampled", None) is not None:
            w, self._presampled = self._presampled, None      # produced by sample_all_weights() for this MC batch
            return w
        S = _MC.samples if samples is None else samples
        seed = _MC.seed if seed is None else seed
        sb = _MC.sample_begin if sample_begin is None else sample_begin
        if eps is None and _MC.eps is not None:
            eps = _MC.eps.get(self.layer_id)
        pk = self._ensure_packed(device)
        w = torch.empty((S, pk["nbytes"]), dtype=torch.int8, device=device)
        if eps is not None:
            eps = eps.to(device=device, dtype=torch.float32).contiguous()
            assert eps.numel() == S * pk["cout"] * pk["k"]
        with timed("sample_weights_i8"):
            _lib.check(_lib.lib().qbnn_sample_weights_i8(_lib.ptr(pk["mu"]), _lib.ptr(pk["sigma"]), pk["cout"], pk["k"], pk["krow"], self.layout,
                                                         C.byref(pk["sp"]), seed, self.layer_id, sb, S, _lib.ptr(eps),
                                                         _lib.ptr(w), pk["nbytes"], _stream()))
        return w

    def _a_hi(self):
        return UINT_BOUNDS[getattr(self.args, "activation_precision", 7)][1]


def sample_all_weights(layers, device):
    """W_q of every layer in `layers` for the current mc_context in ONE kernel launch (qbnn_sample_weights_i8_multi);
    each layer's next sample_weights() call returns its slab.  Same values as per-layer sampling."""
    if _MC.eps is not None:
        return                                    # parity mode (injected eps): per-layer path
    tab = (_lib.SamplerLayer * len(layers))()
    outs = []
    for t, layer in zip(tab, layers):
        pk = layer._ensure_packed(device)
        w = torch.empty((_MC.samples, pk["nbytes"]), dtype=torch.int8, device=device)
        outs.append(w)
        t.mu_packed, t.sigma_packed, t.w_out, t.w_sample_stride = pk["mu"].data_ptr(), pk["sigma"].data_ptr(), w.data_ptr(), pk["nbytes"]
        t.cout, t.k, t.krow, t.layout, t.layer_id, t.params = pk["cout"], pk["k"], pk["krow"], layer.layout, layer.layer_id, pk["sp"]
    with timed("sample_weights_i8_multi"):
        _lib.check(_lib.lib().qbnn_sample_weights_i8_multi(tab, len(layers), _MC.seed, _MC.sample_begin, _MC.samples, _stream()))
    for layer, w in zip(layers, outs):
        layer._presampled = w


# ------------------------------------------------------------------ conv
class Conv2d(_BBBInt8):
    """reference conv_q.Conv2d (conv_q.py:22-177): sampled int8 conv, output quint8 in [0, 255] then the model's
    clamp_activation -- fused here, with the optional residual Add+ReLU of BasicBlock."""
    relu = False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=False,
                 padding_mode='zeros', args=None):
        super().__init__()
        if padding_mode != 'zeros':
            raise NotImplementedError("Currently only zero-padding is supported by quantized conv")
        if in_channels % groups != 0:
            raise ValueError('in_channels must be divisible by groups')
        if out_channels % groups != 0:
            raise ValueError('out_channels must be divisible by groups')
        pair = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = pair(kernel_size), pair(stride), pair(padding), pair(dilation)
        self.groups, self.padding_mode = groups, padding_mode
        if groups != 1 or self.dilation != (1, 1):
            raise NotImplementedError("groups/dilation other than 1 are not on the hot path")
        self._init_common((out_channels, in_channels) + self.kernel_size, bias, args)

    def _get_name(self):
        return 'QuantizedConv2d'

    def forward(self, x, residual=None, add_qparams=None):
        """x: MCQTensor (S samples at once), or -- the reference's own call contract, conv_q.py:107-125 -- a torch quint8
        `(N, C, H, W)` tensor: then ONE stochastic forward (the mc_context's first sample) and a torch quint8 tensor back, so
        the layer can sit inside the reference's own graph (`clamp_activation(layer(x))`, models_bbb.py:229-238).
        residual/add_qparams: fuse `Add` + ReLU of BasicBlock (models_bbb.py:179-182)."""
        if is_torch_quantized(x):
            if len(x.shape) != 4:
                raise ValueError("Input shape must be `(N, C, H, W)`!")
            if residual is not None:
                raise NotImplementedError("the fused residual takes MCQTensor operands")
            with mc_context(1, _MC.seed, _MC.sample_begin, _MC.eps):
                return torch_from_mcq(self.forward(mcq_from_torch(x)), x.device)
        if x.data.dim() != 5:
            raise ValueError("Input shape must be `(S, N, H, W, C)`!")
        dev = x.data.device
        if dev.type != "cuda":
            raise RuntimeError("qbnn layers run on an MI355X only (no CPU fallback)")
        S = _MC.samples
        w = self.sample_weights(dev)
        im2col = None
        if self.layout == LAYOUT_MFMA32 and self.in_channels == 3 and self.kernel_size == (3, 3) and self.stride == (1, 1) and self.padding == (1, 1):
            # the network's first conv runs on the pre-gathered 27-tap patches (K = 27 -> 32), as in ConvNetwork_ResNet.forward_mc
            _, B, H, W, _c = x.data.shape
            im2col = torch.empty((B, H * W, 32), dtype=torch.int8, device=dev)
            src = x.data if x.data.shape[0] == 1 else None
            if src is None:
                raise NotImplementedError("a 3-channel first conv takes an input shared by the samples")
            with timed("im2col3x3_c3"):
                _lib.check(_lib.lib().qbnn_im2col3x3_c3(_lib.ptr(src), B, H, W, x.zero_point, _lib.ptr(im2col), _stream()))
        return self._conv(x, w, S, residual, add_qparams, im2col=im2col)

    def _generic(self, x, w, S, H, W, Cin, Cout, ks, st, pd, out_shape):
        """Any-geometry path (qbnn_conv2d_i8_generic_mc) with per-sample row-major sampled weights: the small nets
        (LeNet, MLP) whose channel counts do not fit the MFMA tiling."""
        pk = self._ensure_packed(x.data.device)
        y = torch.empty((S,) + out_shape, dtype=torch.uint8, device=x.data.device)
        d = _lib.ConvDesc()
        d.B, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.pad = x.data.shape[1], H, W, Cin, Cout, ks, st, pd
        d.s_x, d.z_x = x.scale, x.zero_point
        d.s_w, d.z_w = self.add_weight.scale, self.add_weight.zero_point
        d.s_y, d.z_y = self.scale, self.zero_point
        d.relu, d.a_hi, d.has_bias = int(self.relu), self._a_hi(), int(pk["bias"] is not None)
        with timed("conv_generic_i8 bbb %d->%d k%d" % (Cin, Cout, ks)):
            _lib.check(_lib.lib().qbnn_conv2d_i8_generic_mc(_lib.ptr(x.data), x.sample_stride(), _lib.ptr(w), w.shape[1], _lib.ptr(pk["bias"]),
                                                            _lib.ptr(y), y[0].numel(), S, C.byref(d), _stream()))
        return MCQTensor(y, self.scale, self.zero_point)

    def _conv(self, x, w, S, residual=None, add_qparams=None, im2col=None, w_shared=False, post=None):
        """post: optional dict(desc=_lib.PostDesc, masks=fp32 [S, B, Cout] or None, other=MCQTensor or None, out_qparams=(scale, zp)) --
        the MC-Dropout graphs' dropout (+ Add + ReLU) in the conv's store pass (qbnn_conv2d_i8_post_mc)."""
        if self.layout == LAYOUT_ROWMAJOR:
            if residual is not None or im2col is not None or post is not None:
                raise NotImplementedError("the generic conv path has no fused residual / im2col")
            _, B, H, W, Cin = x.data.shape
            ks, st, pd = self.kernel_size[0], self.stride[0], self.padding[0]
            Ho, Wo = (H + 2 * pd - ks) // st + 1, (W + 2 * pd - ks) // st + 1
            return self._generic(x, w, S, H, W, Cin, self.out_channels, ks, st, pd, (B, Ho, Wo, self.out_channels))
        if self.layout != LAYOUT_MFMA32:
            raise RuntimeError("the layer-level conv kernel takes MFMA32 weights: this layer is packed for a fused block kernel "
                               "(the model switches layouts per path: ConvNetwork_ResNet._apply_layouts)")
        pk = self._ensure_packed(x.data.device)
        _, B, H, W, Cin = x.data.shape
        ks, st, pd = self.kernel_size[0], self.stride[0], self.padding[0]
        Ho, Wo = (H + 2 * pd - ks) // st + 1, (W + 2 * pd - ks) // st + 1
        y = torch.empty((S, B, Ho, Wo, self.out_channels), dtype=torch.uint8, device=x.data.device)
        d = _lib.ConvDesc()
        d.B, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.pad = B, H, W, Cin, self.out_channels, ks, st, pd
        d.s_x, d.z_x = x.scale, x.zero_point
        d.s_w, d.z_w = self.add_weight.scale, self.add_weight.zero_point
        d.s_y, d.z_y = self.scale, self.zero_point
        d.relu, d.a_hi, d.has_bias = int(self.relu), self._a_hi(), int(pk["bias"] is not None)
        xin, xss = x.data, x.sample_stride()
        if im2col is not None:       # layer 0: K = 27 taps of the 3-channel image, pre-gathered once per batch
            xin, xss = im2col, 0
            d.Cin, d.ksize, d.pad, d.x_is_centered_im2col = 32, 1, 0, 1
        res_ss = 0
        if residual is not None:
            d.has_res = 1
            d.s_r, d.z_r = residual.scale, residual.zero_point
            d.s_o, d.z_o = add_qparams
            res_ss = residual.sample_stride()
            assert residual.data.shape[1:] == y.shape[1:]
        key = "conv_i8 %dx%d %d->%d k%d s%d%s" % (H, W, Cin, self.out_channels, ks, st, " +res" if residual is not None else "")
        meta = dict(fused=False, convs=[(H, 3 if im2col is not None else Cin, self.out_channels, 3 if im2col is not None else ks, st, pk["cout"] * pk["k"])])
        if post is not None:
            assert residual is None and im2col is None
            other = post.get("other")
            masks = post.get("masks")
            if masks is not None:
                masks = masks.to(device=y.device, dtype=torch.float32).contiguous()
                assert masks.numel() == S * B * self.out_channels
            if other is not None:
                assert other.data.shape[1:] == y.shape[1:]
            with timed(key + " +post", meta):
                _lib.check(_lib.lib().qbnn_conv2d_i8_post_mc(_lib.ptr(xin), xss, _lib.ptr(w), 0 if w_shared else w.shape[1], _lib.ptr(pk["bias"]),
                                                             _lib.ptr(y), y[0].numel(), S, C.byref(d), C.byref(post["desc"]), _lib.ptr(masks),
                                                             _lib.ptr(None if other is None else other.data),
                                                             0 if other is None else other.sample_stride(), _MC.seed, _MC.sample_begin, _stream()))
            return MCQTensor(y, post["out_qparams"][0], post["out_qparams"][1])
        with timed(key, meta):
            _lib.check(_lib.lib().qbnn_conv2d_i8_mc(_lib.ptr(xin), xss, _lib.ptr(w), 0 if w_shared else w.shape[1], _lib.ptr(pk["bias"]),
                                                    _lib.ptr(None if residual is None else residual.data), res_ss,
                                                    _lib.ptr(y), y[0].numel(), S, C.byref(d), _stream()))
        if residual is not None:
            return MCQTensor(y, add_qparams[0], add_qparams[1])
        return MCQTensor(y, self.scale, self.zero_point)

    @classmethod
    def from_float(cls, mod):
        """Accepts an already converted reference module (conv_q.Conv2d / ConvReLU2d instance)."""
        if hasattr(mod, 'weight_fake_quant'):
            # a reference QAT module (conv_qat.Conv2d / ConvBn2d / ConvReLU2d / ConvBnReLU2d): native restatement of
            # conv_q.py:127-177 (BN folding, observer step, qparams, weight quantisation) in convert.py
            from .convert import convert_qat_module
            st = convert_qat_module(mod)
            q = cls(mod.in_channels, mod.out_channels, mod.kernel_size, mod.stride, mod.padding, mod.dilation, mod.groups,
                    "bias_" in st, mod.padding_mode, args=getattr(mod, "args", None))
            return q.load_reference_state(st, "")
        return cls._from_converted(mod, mod.in_channels, mod.out_channels, mod.kernel_size, mod.stride, mod.padding,
                                   mod.dilation, mod.groups, mod.bias() is not None, mod.padding_mode)


class ConvReLU2d(Conv2d):
    """reference conv_q.ConvReLU2d (conv_q.py:179-225)."""
    relu = True

    def _get_name(self):
        return 'QuantizedConvReLU2d'


# ------------------------------------------------------------------ linear
class Linear(_BBBInt8):
    """reference linear_q.Linear (linear_q.py:13-145)."""
    relu = False
    layout = LAYOUT_ROWMAJOR

    def __init__(self, in_features, out_features, bias_=False, args=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self._init_common((out_features, in_features), bias_, args)

    def _get_name(self):
        return 'QuantizedLinear'

    def forward(self, x):
        """reference linear_q.Linear.forward (linear_q.py:80-94) / LinearReLU.forward (:154-173) for S samples:
        x MCQTensor [S or 1, B, in_features] -> [S, B, out_features]; or a torch quint8 `(N, in_features)` tensor -> one
        stochastic forward, torch quint8 `(N, out_features)` back (the reference's call contract)."""
        if is_torch_quantized(x):
            with mc_context(1, _MC.seed, _MC.sample_begin, _MC.eps):
                return torch_from_mcq(self.forward(mcq_from_torch(x)), x.device)
        d = x.data
        if d.device.type != "cuda":
            raise RuntimeError("qbnn layers run on an MI355X only (no CPU fallback)")
        S = _MC.samples
        w = self.sample_weights(d.device)
        B = d.shape[1]
        flat = MCQTensor(d.reshape(d.shape[0], B, -1), x.scale, x.zero_point, shared=x.shared)
        return Conv2d._generic(self, flat, w, S, 1, 1, self.in_features, self.out_features, 1, 1, 0, (B, self.out_features))

    @classmethod
    def from_float(cls, mod):
        if hasattr(mod, 'weight_fake_quant'):
            from .convert import convert_qat_module      # reference linear_q.py:105-145, restated in convert.py
            st = convert_qat_module(mod)
            q = cls(mod.in_features, mod.out_features, "bias_" in st, args=getattr(mod, "args", None))
            return q.load_reference_state(st, "")
        return cls._from_converted(mod, mod.in_features, mod.out_features, mod.bias() is not None)


class LinearReLU(Linear):
    """reference linear_q.LinearReLU (linear_q.py:147-185)."""
    relu = True

    def _get_name(self):
        return 'QuantizedLinearReLU'
