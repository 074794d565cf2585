"""MC-Dropout models behind the reference's API (BASELINE config 2).

Mirror of reference src/models/stochastic/mcdropout/dropout.py (`BernoulliDropout`, :6-46) and
models_mc.py (`LinearNetwork`, :10-73; `ConvNetwork_LeNet`, :75-111; `ConvNetwork_ResNet`, :162-226) in their converted int8 form (quant_utils.prepare_model -> convert):
deterministic torch.nn.quantized Conv2d / Linear(ReLU) layers with an always-on quantised Bernoulli dropout.
All S MC samples of the active mc_context are evaluated per call; masks come from the Philox uniform stream
(seed, dropout index, global sample index) instead of torch's global generator.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .layers import MCQTensor, QFunctional, QuantizedParam, _MC, bump_state_epoch, mc_context, timed
from .quant import UINT_BOUNDS, check_bits


def _a_hi(args):
    return UINT_BOUNDS[getattr(args, "activation_precision", 7)][1]


class BernoulliDropout(nn.Module):
    """reference mcdropout/dropout.py:6-46.  Always stochastic (no `training` check); p <= 0 is the identity;
    4-D inputs drop whole channels (one draw per (sample, image, channel)); the mask is quantised with mul_mask's own
    (scale, zero_point) and the 1/(1-p) rescale only changes the scale (quantized::mul_scalar)."""

    def __init__(self, p=0.0):
        super().__init__()
        self.p = nn.Parameter(torch.ones((1,)) * p, requires_grad=False)
        self.multiplier = nn.Parameter(torch.ones((1,)) / (1.0 - self.p), requires_grad=False)
        self.mul_mask = QFunctional()
        self.mul_scalar = QFunctional()
        self.layer_id = 0           # Philox tensor id: index of this dropout among the model's dropouts
        self.args = None

    def forward(self, x, masks=None):
        if float(self.p) <= 0.0:
            return x
        d = x.data
        if d.device.type != "cuda":
            raise RuntimeError("qbnn layers run on an MI355X only (no CPU fallback)")
        S = _MC.samples
        B, Cc = d.shape[1], d.shape[-1]
        HW = int(np.prod(d.shape[2:-1])) if d.dim() > 3 else 1
        y = torch.empty((S,) + tuple(d.shape[1:]), dtype=torch.uint8, device=d.device)
        keep = float(np.float32(1.0) - np.float32(self.p.item()))
        if masks is not None:
            masks = masks.to(device=d.device, dtype=torch.float32).contiguous()
            assert masks.numel() == S * B * Cc
        with timed("dropout_q"):
            _lib.check(_lib.lib().qbnn_dropout_q_mc(_lib.ptr(d), x.sample_stride(), B, HW, Cc, keep, x.scale, x.zero_point,
                                                    self.mul_mask.scale, self.mul_mask.zero_point, _a_hi(self.args), _MC.seed,
                                                    self.layer_id, _MC.sample_begin, _lib.ptr(masks), _lib.ptr(y), y[0].numel(), S,
                                                    _lib.current_stream()))
        # mul_scalar: integers and zero point unchanged, scale (a double in torch) times 1/(1-p)
        return MCQTensor(y, self.mul_mask.scale * float(np.float32(self.multiplier.item())), self.mul_mask.zero_point)

    def extra_repr(self):
        return 'p={}, quant={}'.format(self.p.item(), True)


class _QDeterministic(nn.Module):
    """A converted standard quantised layer (torch.nn.quantized.Conv2d / Linear / LinearReLU): fixed qint8 weight."""
    relu = False

    def _init(self, w_shape, args):
        self._weight = QuantizedParam(np.zeros(w_shape, np.int8), 1.0, 0)
        self.bias_ = None
        self.scale, self.zero_point = 1.0, 0
        self.args = args
        self._dev = None

    def weight(self):
        return self._weight

    def bias(self):
        return self.bias_

    def load_reference_state(self, state, prefix):
        self._weight = QuantizedParam(state[prefix + "weight"], state[prefix + "weight.q_scale"], state[prefix + "weight.q_zero_point"])
        b = state.get(prefix + "bias", None)
        self.bias_ = None if b is None or np.asarray(b).size == 0 else torch.from_numpy(np.asarray(b, np.float32).copy())
        self.scale, self.zero_point = float(state[prefix + "scale"]), int(state[prefix + "zero_point"])
        self._dev = None
        self._pk = None          # the packed-fragment copy too: it is keyed by (krow, device) only
        bump_state_epoch()
        return self

    def _device_params(self, device, w_ohwi):
        if self._dev is None or self._dev["device"] != device:
            self._dev = dict(device=device, w=torch.from_numpy(np.ascontiguousarray(w_ohwi)).to(device),
                             bias=None if self.bias_ is None else self.bias_.to(device=device, dtype=torch.float32).contiguous())
        return self._dev

    def _packed_mfma(self, device, w_rows, krow):
        """The fixed weight as QBNN_LAYOUT_MFMA32 fragments (w_rows: int8 [Cout, K] in the kernel's K order)."""
        key = ("mfma", krow)
        if getattr(self, "_pk", None) is None or self._pk["key"] != key or self._pk["device"] != device:
            L = _lib.lib()
            cout, k = w_rows.shape
            nbytes = L.qbnn_packed_weight_bytes(cout, k, krow, 0)        # QBNN_LAYOUT_MFMA32
            dst = np.zeros(nbytes, np.int8)
            src = np.ascontiguousarray(w_rows, dtype=np.int8)
            _lib.check(L.qbnn_pack_weights_host(src.ctypes.data_as(C.c_void_p), cout, k, krow, 0, dst.ctypes.data_as(C.c_void_p)))
            self._pk = dict(key=key, device=device, w=torch.from_numpy(dst).to(device),
                            bias=None if self.bias_ is None else self.bias_.to(device=device, dtype=torch.float32).contiguous())
        return self._pk

    def _desc(self, x, B, H, Cin, Cout, ks, pad):
        c = _lib.ConvDesc()
        c.B, c.H, c.W, c.Cin, c.Cout, c.ksize, c.stride, c.pad = B, H, H, Cin, Cout, ks, 1, pad
        c.s_x, c.z_x = x.scale, x.zero_point
        c.s_w, c.z_w = self._weight.q_scale(), self._weight.q_zero_point()
        c.s_y, c.z_y = self.scale, self.zero_point
        c.relu, c.a_hi, c.has_bias = int(self.relu), _a_hi(self.args), int(self.bias_ is not None)
        return c

    def _run(self, x, w_ohwi, H, W, Cin, Cout, ks, stride, pad, out_shape):
        d = x.data
        if d.device.type != "cuda":
            raise RuntimeError("qbnn layers run on an MI355X only (no CPU fallback)")
        S = x.samples if not x.shared else 1
        dev = self._device_params(d.device, w_ohwi)
        y = torch.empty((S,) + out_shape, dtype=torch.uint8, device=d.device)
        c = _lib.ConvDesc()
        c.B, c.H, c.W, c.Cin, c.Cout, c.ksize, c.stride, c.pad = d.shape[1], H, W, Cin, Cout, ks, stride, pad
        c.s_x, c.z_x = x.scale, x.zero_point
        c.s_w, c.z_w = self._weight.q_scale(), self._weight.q_zero_point()
        c.s_y, c.z_y = self.scale, self.zero_point
        c.relu, c.a_hi, c.has_bias = int(self.relu), _a_hi(self.args), int(dev["bias"] is not None)
        with timed("conv_generic_i8 %d->%d k%d" % (Cin, Cout, ks)):
            _lib.check(_lib.lib().qbnn_conv2d_i8_generic_mc(_lib.ptr(d), x.sample_stride(), _lib.ptr(dev["w"]), 0, _lib.ptr(dev["bias"]),
                                                            _lib.ptr(y), y[0].numel(), S, C.byref(c), _lib.current_stream()))
        return MCQTensor(y, self.scale, self.zero_point, shared=x.shared)


class QConv2d(_QDeterministic):
    """torch.nn.quantized.Conv2d as the reference's convert() produces it for models_mc.py:83,86."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False, args=None):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding = in_channels, out_channels, kernel_size, stride, padding
        self._init((out_channels, in_channels, kernel_size, kernel_size), args)

    # geometries of the tuned LDS-tiled layer kernel (qbnn_conv2d_i8_mc): (H, Cin, Cout, k, stride), pad = (k - 1) / 2
    _TUNED = {(32, 24, 24, 3, 1), (32, 24, 48, 3, 2), (32, 24, 48, 1, 2), (16, 48, 48, 3, 1), (16, 48, 96, 3, 2), (16, 48, 96, 1, 2),
              (8, 96, 96, 3, 1), (8, 96, 192, 3, 2), (8, 96, 192, 1, 2), (4, 192, 192, 3, 1)}

    def _tuned(self, layout=0):
        """This conv as a deterministic layers.Conv2d: the fixed qint8 weight is the one 'sample', shared by all MC samples.  One object per
        packed layout (MFMA32 for the layer-level kernel and most fused ones, MFMA32_TAIL for the blocks behind the fused stem)."""
        if getattr(self, "_fast", None) is None:
            self._fast = {}
        if layout not in self._fast:
            from .layers import Conv2d as _Conv, ConvReLU2d as _ConvReLU
            m = (_ConvReLU if self.relu else _Conv)(self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding,
                                                    bias=self.bias_ is not None, args=self.args)
            m.deterministic = True
            st = {"weight": self._weight.int_repr(), "weight.q_scale": self._weight.q_scale(), "weight.q_zero_point": self._weight.q_zero_point(),
                  "scale": self.scale, "zero_point": self.zero_point}
            if self.bias_ is not None:
                st["bias"] = self.bias_.cpu().numpy()
            m.load_reference_state(st, "")
            m.layout = layout
            self._fast[layout] = m
        return self._fast[layout]

    def load_reference_state(self, state, prefix):
        self._fast = None
        return super().load_reference_state(state, prefix)

    def is_tuned(self, x):
        _, B, H, W, Cin = x.data.shape
        ks = self.kernel_size
        return H == W and self.padding == (ks - 1) // 2 and (H, Cin, self.out_channels, ks, self.stride) in self._TUNED

    def forward_post(self, x, drop, masks=None, add=None):
        """conv -> BernoulliDropout `drop` (-> quantized::add with the MCQTensor `other` -> ReLU; add = (other, s_o, z_o)) in ONE launch:
        the dropout and the Add run in the conv kernel's epilogue (qbnn_conv2d_i8_post_mc).  x must not be sample-shared."""
        assert self.is_tuned(x) and not x.shared and float(drop.p) > 0.0
        m = self._tuned()
        pk = m._ensure_packed(x.data.device)
        q = _lib.PostDesc()
        q.keep_prob = float(np.float32(1.0) - np.float32(drop.p.item()))
        q.s_m, q.z_m, q.drop_layer_id = drop.mul_mask.scale, drop.mul_mask.zero_point, drop.layer_id
        s_d = drop.mul_mask.scale * float(np.float32(drop.multiplier.item()))      # mul_scalar: only the scale changes
        out_q = (s_d, drop.mul_mask.zero_point)
        other = None
        if add is not None:
            other, s_o, z_o = add
            q.add, q.s_a, q.s_b, q.z_b, q.s_o, q.z_o = 1, s_d, other.scale, other.zero_point, s_o, z_o
            out_q = (s_o, z_o)
        return m._conv(x, pk["mu"].reshape(1, -1), x.samples, w_shared=True, post=dict(desc=q, masks=masks, other=other, out_qparams=out_q))

    _SMALL = {(14, 20, 50, 5, 1, 2)}        # qbnn_conv_pool_drop_i8_mc geometries: (H, Cin, Cout, k, stride, pad)

    def is_small(self, x):
        _, B, H, W, Cin = x.data.shape
        return (H == W and (H, Cin, self.out_channels, self.kernel_size, self.stride, self.padding) in self._SMALL
                and _a_hi(self.args) <= 127 and 0 <= x.zero_point <= 127 and 0 <= self.zero_point <= 127)

    def forward_pool_drop_flat(self, x, pool, drop, masks=None, drop_in=None, masks_in=None):
        """[BernoulliDropout `drop_in` on x ->] conv -> clamp -> MaxPool2d(2,2) -> BernoulliDropout -> Flatten (NHWC order) in ONE launch:
        rows [S, B, ld] with a 16-byte multiple pitch (bytes beyond the map are 0).  Returns (MCQTensor of the rows, row width)."""
        _, B, H, W, Cin = x.data.shape
        S = _MC.samples
        din, s_in, z_in = None, 0.0, 0
        if drop_in is not None and float(drop_in.p) > 0.0:
            din = _lib.DropoutDesc(float(np.float32(1.0) - np.float32(drop_in.p.item())), drop_in.mul_mask.scale, drop_in.mul_mask.zero_point,
                                   drop_in.layer_id)
            s_in, z_in = x.scale, x.zero_point
            x = MCQTensor(x.data, drop_in.mul_mask.scale * float(np.float32(drop_in.multiplier.item())), drop_in.mul_mask.zero_point, shared=x.shared)
            if masks_in is not None:
                masks_in = masks_in.to(device=x.data.device, dtype=torch.float32).contiguous()
                assert masks_in.numel() == S * B * Cin
        w = self._weight.int_repr().transpose(0, 2, 3, 1).reshape(self.out_channels, -1)
        pk = self._packed_mfma(x.data.device, w, self.kernel_size * Cin)
        Ho = H // 2 if pool else H
        width = Ho * Ho * self.out_channels
        ld = (width + 15) // 16 * 16
        y = torch.empty((S, B, ld), dtype=torch.uint8, device=x.data.device)
        c = self._desc(x, B, H, Cin, self.out_channels, self.kernel_size, self.padding)
        dd, out_q = None, (self.scale, self.zero_point)
        if drop is not None and float(drop.p) > 0.0:
            dd = _lib.DropoutDesc(float(np.float32(1.0) - np.float32(drop.p.item())), drop.mul_mask.scale, drop.mul_mask.zero_point, drop.layer_id)
            out_q = (drop.mul_mask.scale * float(np.float32(drop.multiplier.item())), drop.mul_mask.zero_point)
        if masks is not None:
            masks = masks.to(device=y.device, dtype=torch.float32).contiguous()
            assert masks.numel() == S * B * self.out_channels
        with timed("conv_pool_drop_i8 %d->%d k%d" % (Cin, self.out_channels, self.kernel_size)):
            _lib.check(_lib.lib().qbnn_conv_pool_drop_i8_mc(_lib.ptr(x.data), x.sample_stride(), _lib.ptr(pk["w"]), 0, _lib.ptr(pk["bias"]),
                                                            _lib.ptr(y), y[0].numel(), ld, S, C.byref(c), int(pool),
                                                            None if dd is None else C.byref(dd), _lib.ptr(masks if dd is not None else None),
                                                            None if din is None else C.byref(din), _lib.ptr(masks_in if din is not None else None),
                                                            s_in, z_in, _MC.seed, _MC.sample_begin, _lib.current_stream()))
        return MCQTensor(y, out_q[0], out_q[1]), width

    def forward(self, x):
        _, B, H, W, Cin = x.data.shape
        ks, st, pd = self.kernel_size, self.stride, self.padding
        Ho, Wo = (H + 2 * pd - ks) // st + 1, (W + 2 * pd - ks) // st + 1
        if self.is_tuned(x):
            m = self._tuned()
            pk = m._ensure_packed(x.data.device)
            y = m._conv(x, pk["mu"].reshape(1, -1), 1 if x.shared else x.samples, w_shared=True)
            return MCQTensor(y.data, self.scale, self.zero_point, shared=x.shared)
        w = self._weight.int_repr().transpose(0, 2, 3, 1)
        return self._run(x, w, H, W, Cin, self.out_channels, ks, st, pd, (B, Ho, Wo, self.out_channels))


class QLinear(_QDeterministic):
    """torch.nn.quantized.Linear (models_mc.py:93).  `nhwc_from` = (C, H, W) when the input is a flattened conv map: the
    reference flattens NCHW (src/utils.py:40-47), the build's activations are NHWC, so the weight columns are permuted."""

    def __init__(self, in_features, out_features, bias=False, args=None, nhwc_from=None):
        super().__init__()
        self.in_features, self.out_features, self.nhwc_from = in_features, out_features, nhwc_from
        self._init((out_features, in_features), args)

    def forward(self, x):
        d = x.data
        B = d.shape[1]
        w = self._weight.int_repr()
        if self.nhwc_from is not None:
            c, h, wd = self.nhwc_from
            w = w.reshape(self.out_features, c, h, wd).transpose(0, 2, 3, 1).reshape(self.out_features, -1)
        flat = MCQTensor(d.reshape(d.shape[0], B, -1), x.scale, x.zero_point, shared=x.shared)
        return self._run(flat, w, 1, 1, self.in_features, self.out_features, 1, 1, 0, (B, self.out_features))


    def forward_rows(self, x, width, drop=None, masks=None, dense_out=False):
        """x: MCQTensor of rows [S|1, B, ldx] (ldx % 16 == 0, `width` = in_features real bytes per row) -> Linear(ReLU) -> clamp
        [-> BernoulliDropout, one draw per element] on the GEMM kernel; rows [S, B, ldy] out (ldy = out_features if dense_out)."""
        d = x.data
        S = _MC.samples if not (x.shared and drop is None) else 1
        B, ldx = d.shape[1], d.shape[2]
        assert width == self.in_features and ldx % 16 == 0
        w = self._weight.int_repr()
        if self.nhwc_from is not None:
            c, h, wd = self.nhwc_from
            w = w.reshape(self.out_features, c, h, wd).transpose(0, 2, 3, 1).reshape(self.out_features, -1)
        pk = self._packed_mfma(d.device, w, self.in_features)
        ldy = self.out_features if dense_out else (self.out_features + 15) // 16 * 16
        y = torch.empty((S, B, ldy), dtype=torch.uint8, device=d.device)
        c = self._desc(x, B, 1, self.in_features, self.out_features, 1, 0)
        dd, out_q = None, (self.scale, self.zero_point)
        if drop is not None and float(drop.p) > 0.0:
            dd = _lib.DropoutDesc(float(np.float32(1.0) - np.float32(drop.p.item())), drop.mul_mask.scale, drop.mul_mask.zero_point, drop.layer_id)
            out_q = (drop.mul_mask.scale * float(np.float32(drop.multiplier.item())), drop.mul_mask.zero_point)
        if masks is not None:
            masks = masks.to(device=y.device, dtype=torch.float32).contiguous()
            assert masks.numel() == S * B * self.out_features
        with timed("linear_i8 %d->%d" % (self.in_features, self.out_features)):
            _lib.check(_lib.lib().qbnn_linear_i8_mc(_lib.ptr(d), x.sample_stride(), ldx, _lib.ptr(pk["w"]), 0, _lib.ptr(pk["bias"]), _lib.ptr(y),
                                                    y[0].numel(), ldy, S, C.byref(c), None if dd is None else C.byref(dd),
                                                    _lib.ptr(masks if dd is not None else None), _MC.seed, _MC.sample_begin, _lib.current_stream()))
        return MCQTensor(y, out_q[0], out_q[1], shared=(x.shared and drop is None)), self.out_features


class QLinearReLU(QLinear):
    relu = True


class MaxPool2dQ(nn.Module):
    def __init__(self, args=None):
        super().__init__()
        self.args = args

    def forward(self, x):
        d = x.data
        S, B, H, W, Cc = d.shape
        y = torch.empty((S, B, H // 2, W // 2, Cc), dtype=torch.uint8, device=d.device)
        with timed("maxpool2_q"):
            _lib.check(_lib.lib().qbnn_maxpool2_q_mc(_lib.ptr(d), x.sample_stride(), B, H, W, Cc, _a_hi(self.args), _lib.ptr(y),
                                                     y[0].numel(), S, _lib.current_stream()))
        return MCQTensor(y, x.scale, x.zero_point, shared=x.shared)


class LinearNetwork(nn.Module):
    """reference mcdropout/models_mc.py:10-73 (`linear_mc`, the MC-Dropout regression MLP), converted int8 form: QuantStub ->
    3 x LinearReLU(100) with a per-element BernoulliDropout after the first two -> heads `mu` / `log_var` = [BernoulliDropout,
    Linear(100, 1)] -> DeQuant -> (mu, exp(log_var)).  Mask draw order = execution order: layers.2, layers.5, mu.0, log_var.0."""

    def __init__(self, input_size, output_size, q, args):
        super().__init__()
        if not q:
            raise NotImplementedError("this class is the converted int8 graph; ModelFactory routes q=False to models_mc_f32.LinearNetwork")
        check_bits(args)
        self.args, self.q = args, q
        self.input_size = 1
        for i in input_size:
            self.input_size *= int(i)
        self.output_size = int(output_size)
        ident = nn.Identity
        self.layers = nn.ModuleList([QLinearReLU(self.input_size, 100, bias=True, args=args), ident(), BernoulliDropout(args.p),
                                     QLinearReLU(100, 100, bias=True, args=args), ident(), BernoulliDropout(args.p),
                                     QLinearReLU(100, 100, bias=True, args=args), ident()])
        self.mu = nn.ModuleList([BernoulliDropout(args.p), QLinear(100, 1, bias=True, args=args)])
        self.log_var = nn.ModuleList([BernoulliDropout(args.p), QLinear(100, 1, bias=True, args=args)])
        for i, m in enumerate(self.dropouts()):
            m.layer_id, m.args = i, args
        from .models import QuantStub
        self.quant = QuantStub()

    def dropouts(self):
        return [self.layers[2], self.layers[5], self.mu[0], self.log_var[0]]

    def load_reference_state(self, state):
        for n, m in (("layers.0.", self.layers[0]), ("layers.3.", self.layers[3]), ("layers.6.", self.layers[6]), ("mu.1.", self.mu[1]),
                     ("log_var.1.", self.log_var[1])):
            m.load_reference_state(state, n)
        for n, d in zip(("layers.2.", "layers.5.", "mu.0.", "log_var.0."), self.dropouts()):
            _load_dropout(d, state, n)
        self.quant.scale = float(np.asarray(state["quant.scale"]).reshape(-1)[0])
        self.quant.zero_point = int(np.asarray(state["quant.zero_point"]).reshape(-1)[0])
        return self

    def forward_mc(self, x, record=None, masks=None):
        """All S samples of the current mc_context -> (mu [S,B,1], var [S,B,1]) fp32.  masks: optional list of fp32 [S, B, 100] in draw
        order (parity mode); record: optional dict that receives every layer's quint8 output."""
        if x.device.type != "cuda":
            raise RuntimeError("qbnn models run on an MI355X only (no CPU fallback)")
        rec = (lambda k, v: record.__setitem__(k, v)) if record is not None else (lambda k, v: None)
        masks = list(masks) if masks is not None else None
        take = lambda: masks.pop(0) if masks is not None else None
        x2 = x.to(torch.float32).reshape(x.shape[0], -1).contiguous()
        B, K = x2.shape
        xq = torch.empty((1, B, K), dtype=torch.uint8, device=x.device)
        _lib.check(_lib.lib().qbnn_quantize_input_nchw(_lib.ptr(x2), B, K, 1, 1, self.quant.scale, self.quant.zero_point, _a_hi(self.args),
                                                       _lib.ptr(xq), _lib.current_stream()))
        h = MCQTensor(xq, self.quant.scale, self.quant.zero_point, shared=True); rec("quant.out", h.data)
        h = self.layers[0](h); rec("layers.0.out", h.data)
        h = self.layers[2](h, take()); rec("layers.2.out", h.data)
        h = self.layers[3](h); rec("layers.3.out", h.data)
        h = self.layers[5](h, take()); rec("layers.5.out", h.data)
        h = self.layers[6](h); rec("layers.6.out", h.data)
        hm = self.mu[0](h, take()); rec("mu.0.out", hm.data)
        qm = self.mu[1](hm); rec("mu.1.out", qm.data)
        hv = self.log_var[0](h, take()); rec("log_var.0.out", hv.data)
        qv = self.log_var[1](hv); rec("log_var.1.out", qv.data)
        return qm.dequantize(), torch.exp(qv.dequantize())

    def forward(self, x):
        with mc_context(1, _MC.seed, _MC.sample_begin, _MC.eps):
            mu, var = self.forward_mc(x)
        return mu[0], var[0]


class ConvNetwork_LeNet(nn.Module):
    """reference mcdropout/models_mc.py:75-111 (`conv_lenet_mc`), converted int8 form."""
    fast_tail = True        # layers.3 .. 10 on qbnn_conv_pool_drop_i8_mc + qbnn_linear_i8_mc (False: one any-geometry launch per op)

    def __init__(self, input_size, output_size, q, args):
        super().__init__()
        if not q:
            raise NotImplementedError("this class is the converted int8 graph; ModelFactory routes q=False to models_mc_f32.ConvNetwork_LeNet")
        check_bits(args)
        self.args, self.q = args, q
        self.init_channels = input_size[0]
        self.output_size = int(output_size)
        ident = nn.Identity
        self.layers = nn.ModuleList([QConv2d(self.init_channels, 20, 5, padding=2, args=args), BernoulliDropout(args.p), MaxPool2dQ(args),
                                     QConv2d(20, 50, 5, padding=2, args=args), BernoulliDropout(args.p), MaxPool2dQ(args),
                                     ident(),                                  # Flatten
                                     QLinearReLU(50 * 7 * 7, 500, args=args, nhwc_from=(50, 7, 7)), ident(),
                                     BernoulliDropout(args.p),
                                     QLinear(500, output_size, args=args)])
        for i, m in enumerate(self.dropouts()):
            m.layer_id, m.args = i, args
        from .models import QuantStub
        self.quant = QuantStub()

    def dropouts(self):
        return [m for m in self.layers if isinstance(m, BernoulliDropout)]

    def load_reference_state(self, state):
        for i in (0, 3, 7, 10):
            self.layers[i].load_reference_state(state, f"layers.{i}.")
        for i in (1, 4, 9):
            d = self.layers[i]
            d.mul_mask = QFunctional(state[f"layers.{i}.mul_mask.scale"], state[f"layers.{i}.mul_mask.zero_point"])
            d.p.data = torch.from_numpy(np.asarray(state[f"layers.{i}.p"], np.float32).reshape(1).copy())
            d.multiplier.data = torch.from_numpy(np.asarray(state[f"layers.{i}.multiplier"], np.float32).reshape(1).copy())
        self.quant.scale = float(np.asarray(state["quant.scale"]).reshape(-1)[0])
        self.quant.zero_point = int(np.asarray(state["quant.zero_point"]).reshape(-1)[0])
        return self

    def forward_mc(self, x, record=None, masks=None):
        """All S samples of the current mc_context -> softmax probabilities [S, B, classes].
        masks: optional {dropout index: fp32 [S, B, C]} (parity mode)."""
        if x.device.type != "cuda":
            raise RuntimeError("qbnn models run on an MI355X only (no CPU fallback)")
        S = _MC.samples
        x = x.to(torch.float32).contiguous()
        B, Cc, H, W = x.shape
        xq = torch.empty((1, B, H, W, Cc), dtype=torch.uint8, device=x.device)
        a_hi = _a_hi(self.args)
        _lib.check(_lib.lib().qbnn_quantize_input_nchw(_lib.ptr(x), B, Cc, H, W, self.quant.scale, self.quant.zero_point, a_hi,
                                                       _lib.ptr(xq), _lib.current_stream()))
        h = MCQTensor(xq, self.quant.scale, self.quant.zero_point, shared=True)
        if record is not None:
            record["quant.out"] = h.data
        di = 0
        skip = -1
        if record is None and self.fast_tail:
            # layers 0..2 (conv on the sample-shared input, pool, dropout) as below, then the rest in three launches
            h = self.layers[2](self.layers[0](h))            # conv + pool on the sample-shared input: once per batch
            if self.layers[3].is_small(h) and all(0 <= self.layers[i].mul_mask.zero_point <= 127 for i in (1, 4, 9)):
                rows, width = self.layers[3].forward_pool_drop_flat(h, True, self.layers[4], None if masks is None else masks[1],
                                                                    drop_in=self.layers[1], masks_in=None if masks is None else masks[0])
                rows, width = self.layers[7].forward_rows(rows, width, self.layers[9], None if masks is None else masks[2])
                rows, _ = self.layers[10].forward_rows(rows, width, dense_out=True)
                probs = torch.empty((S, B, self.output_size), dtype=torch.float32, device=x.device)
                _lib.check(_lib.lib().qbnn_dequant_softmax_mc(_lib.ptr(rows.data), rows.sample_stride(), B, self.output_size, rows.scale,
                                                              rows.zero_point, _lib.ptr(probs), S, _lib.current_stream()))
                return probs
            h = MCQTensor(xq, self.quant.scale, self.quant.zero_point, shared=True)
        for i, layer in enumerate(self.layers):
            if isinstance(layer, nn.Identity) or i == skip:
                continue
            if isinstance(layer, BernoulliDropout):
                m = None if masks is None else masks[di]
                di += 1
                nxt = self.layers[i + 1] if i + 1 < len(self.layers) else None
                if record is None and isinstance(nxt, MaxPool2dQ):
                    # dropout -> max-pool == max-pool -> dropout, bit for bit: the mask is constant over a pooling window (one
                    # draw per image and channel) and the quantised multiply is monotone in x for a factor >= 0, so it commutes
                    # with max.  Pooling first quarters the dropout's traffic, and pools a sample-shared tensor once.
                    h = layer(nxt(h), m)
                    skip = i + 1
                    continue
                h = layer(h, m)
            else:
                h = layer(h)
            if record is not None:
                record[f"layers.{i}.out"] = h.data
        d = h.data                                   # [S, B, classes]
        probs = torch.empty((S, B, self.output_size), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().qbnn_dequant_softmax_mc(_lib.ptr(d), h.sample_stride(), B, self.output_size, h.scale, h.zero_point,
                                                      _lib.ptr(probs), S, _lib.current_stream()))
        return probs

    def forward(self, x):
        with mc_context(1, _MC.seed, _MC.sample_begin, _MC.eps):
            return self.forward_mc(x)[0]


class QConvReLU2d(QConv2d):
    """torch.nn.intrinsic.quantized.ConvReLU2d (conv + folded BN + ReLU after convert)."""
    relu = True


def _load_dropout(d, state, prefix):
    d.mul_mask = QFunctional(state[prefix + "mul_mask.scale"], state[prefix + "mul_mask.zero_point"])
    d.p.data = torch.from_numpy(np.asarray(state[prefix + "p"], np.float32).reshape(1).copy())
    d.multiplier.data = torch.from_numpy(np.asarray(state[prefix + "multiplier"], np.float32).reshape(1).copy())


def add_relu_q(a, b, s_o, z_o, a_hi, relu=True):
    """quantized::add + clamp + ReLU + clamp on two MCQTensors (either may be shared by the samples)."""
    S = _MC.samples
    da, db = a.data, b.data
    n = da[0].numel()
    y = torch.empty((S,) + tuple(da.shape[1:]), dtype=torch.uint8, device=da.device)
    with timed("add_relu_q"):
        _lib.check(_lib.lib().qbnn_add_relu_q_mc(_lib.ptr(da), a.sample_stride(), a.scale, a.zero_point, _lib.ptr(db), b.sample_stride(),
                                                 b.scale, b.zero_point, _lib.ptr(y), n, n, s_o, z_o, a_hi, int(relu), S, _lib.current_stream()))
    return MCQTensor(y, s_o, z_o)


def _fill_drop(d, drop, mask, keep):
    """qbnn_drop_desc of a BernoulliDropout behind a conv of a fused block; mask: optional fp32 [S, B, C] (kept alive in `keep`)."""
    d.keep_prob = float(np.float32(1.0) - np.float32(drop.p.item()))
    d.s_m, d.z_m, d.layer_id = drop.mul_mask.scale, drop.mul_mask.zero_point, drop.layer_id
    d.s_out = drop.mul_mask.scale * float(np.float32(drop.multiplier.item()))      # mul_scalar: only the scale changes
    if mask is not None:
        mask = mask.to(device="cuda", dtype=torch.float32).contiguous()
        keep.append(mask)
    d.mask_in = _lib.ptr(mask)


def _fill_block(d, blk, dev, layout=0):
    """qbnn_block_desc of a converted MC-Dropout BasicBlock: the fixed qint8 weights as `layout` fragments (MFMA32; MFMA32_TAIL for the two
    blocks behind the fused stem on the 16-wave kernel), sample stride 0."""
    ca, cb = blk.stem[0]._tuned(layout), blk.stem[4]._tuned(layout)
    d.w_layout = layout
    pa, pb = ca._ensure_packed(dev), cb._ensure_packed(dev)
    d.w_a, d.w_a_sample_stride, d.bias_a = pa["mu"].data_ptr(), 0, (pa["bias"].data_ptr() if pa["bias"] is not None else None)
    d.s_wa, d.z_wa, d.s_a, d.z_a = ca.add_weight.scale, ca.add_weight.zero_point, blk.stem[0].scale, blk.stem[0].zero_point
    d.w_b, d.w_b_sample_stride, d.bias_b = pb["mu"].data_ptr(), 0, (pb["bias"].data_ptr() if pb["bias"] is not None else None)
    d.s_wb, d.z_wb, d.s_b, d.z_b = cb.add_weight.scale, cb.add_weight.zero_point, blk.stem[4].scale, blk.stem[4].zero_point
    d.s_o, d.z_o = blk.add.scale, blk.add.zero_point


def run_identity_chain_drop(blocks, x, masks, stem=None):
    """Identity MC-Dropout blocks in ONE fused kernel (qbnn_block_chain_drop_i8_mc / qbnn_stem_chain_drop_i8_mc): both convs, both
    dropouts, the Add and the ReLU per block; activations stay in LDS.  masks: the model's remaining injected masks in draw
    order (consumed here) or None.  stem = (layers.0, layers.3, patches [B, 1024, 32], input scale)."""
    S = _MC.samples
    dev = x.data.device if stem is None else stem[2].device
    keep = []
    n = len(blocks)
    descs, drops = (_lib.BlockDesc * n)(), (_lib.DropDesc * (2 * n))()
    d0 = _lib.DropDesc()
    if stem is not None:
        _fill_drop(d0, stem[1], masks.pop(0) if masks is not None else None, keep)
    from .layers import LAYOUT_MFMA32, LAYOUT_MFMA32_TAIL, w16_enabled
    layout = LAYOUT_MFMA32_TAIL if (stem is not None and n == 2 and w16_enabled()) else LAYOUT_MFMA32
    for k, blk in enumerate(blocks):
        assert len(blk.shortcut) == 0
        _fill_block(descs[k], blk, dev, layout)
        _fill_drop(drops[2 * k], blk.stem[3], masks.pop(0) if masks is not None else None, keep)
        _fill_drop(drops[2 * k + 1], blk.stem[6], masks.pop(0) if masks is not None else None, keep)
    a_hi = _a_hi(blocks[0].args)
    last = blocks[-1].add
    if stem is not None:
        l0, _, col, s_in = stem
        m0 = l0._tuned()
        pk0 = m0._ensure_packed(dev)
        B = col.shape[0]
        y = torch.empty((S, B, 32, 32, 24), dtype=torch.uint8, device=dev)
        with timed("stem + block_chain_drop_i8 x%d 32x32 c24" % n):
            _lib.check(_lib.lib().qbnn_stem_chain_drop_i8_mc(_lib.ptr(col), B, _lib.ptr(pk0["mu"]), 0, _lib.ptr(pk0["bias"]), s_in,
                                                             m0.add_weight.scale, m0.add_weight.zero_point, l0.scale, l0.zero_point, a_hi,
                                                             C.byref(d0), descs, drops, n, _lib.ptr(y), y[0].numel(), S, _MC.seed, _MC.sample_begin,
                                                             _lib.current_stream()))
        return MCQTensor(y, last.scale, last.zero_point)
    _, B, H, W, Cc = x.data.shape
    y = torch.empty((S, B, H, W, Cc), dtype=torch.uint8, device=dev)
    with timed("block_chain_drop_i8 x%d %dx%d c%d" % (n, H, W, Cc)):
        _lib.check(_lib.lib().qbnn_block_chain_drop_i8_mc(_lib.ptr(x.data), x.sample_stride(), x.scale, x.zero_point, B, H, Cc, a_hi, descs, drops, n,
                                                          _lib.ptr(y), y[0].numel(), S, _MC.seed, _MC.sample_begin, _lib.current_stream()))
    return MCQTensor(y, last.scale, last.zero_point)


def run_down_block_drop(blk, x, masks):
    """A down-sampling MC-Dropout block (shortcut conv + stem + three dropouts + Add + ReLU) in ONE fused kernel."""
    S = _MC.samples
    dev = x.data.device
    keep = []
    d = _lib.DownDesc()
    _fill_block(d.blk, blk, dev)
    cs = blk.shortcut[0]._tuned()
    ps = cs._ensure_packed(dev)
    d.w_s, d.w_s_sample_stride, d.bias_s = ps["mu"].data_ptr(), 0, (ps["bias"].data_ptr() if ps["bias"] is not None else None)
    d.s_ws, d.z_ws, d.s_s, d.z_s = cs.add_weight.scale, cs.add_weight.zero_point, blk.shortcut[0].scale, blk.shortcut[0].zero_point
    drops = (_lib.DropDesc * 3)()
    for i, dm in enumerate((blk.stem[3], blk.stem[6], blk.shortcut[2])):            # the reference's draw order
        _fill_drop(drops[i], dm, masks.pop(0) if masks is not None else None, keep)
    _, B, H, W, Cin = x.data.shape
    Cout = blk.stem[0].out_channels
    y = torch.empty((S, B, H // 2, W // 2, Cout), dtype=torch.uint8, device=dev)
    with timed("block_down_drop_i8 %dx%d %d->%d" % (H, W, Cin, Cout)):
        _lib.check(_lib.lib().qbnn_block_down_drop_i8_mc(_lib.ptr(x.data), x.sample_stride(), x.scale, x.zero_point, B, H, Cin, _a_hi(blk.args),
                                                         C.byref(d), drops, _lib.ptr(y), y[0].numel(), S, _MC.seed, _MC.sample_begin,
                                                         _lib.current_stream()))
    return MCQTensor(y, blk.add.scale, blk.add.zero_point)


class BasicBlock(nn.Module):
    """reference mcdropout/models_mc.py:116-160 after fuse_model + convert: stem = ConvReLU2d, -, -, Dropout, Conv2d, -, Dropout;
    shortcut = Conv2d(1x1, stride), -, Dropout where the shape changes; add; end ReLU."""
    expansion = 1

    def __init__(self, in_planes, planes, stride=1, q=True, args=None):
        super().__init__()
        self.args = args
        ident = nn.Identity
        self.stem = nn.ModuleList([QConvReLU2d(in_planes, planes, 3, stride, 1, bias=True, args=args), ident(), ident(), BernoulliDropout(args.p),
                                   QConv2d(planes, planes, 3, 1, 1, bias=True, args=args), ident(), BernoulliDropout(args.p)])
        self.shortcut = nn.ModuleList([])
        if stride != 1 or in_planes != planes:
            self.shortcut.append(QConv2d(in_planes, planes, 1, stride, 0, bias=True, args=args))
            self.shortcut.append(ident())
            self.shortcut.append(BernoulliDropout(args.p))
        self.add = QFunctional()

    def dropouts(self):
        return [self.stem[3], self.stem[6]] + ([self.shortcut[2]] if len(self.shortcut) else [])

    def load_reference_state(self, state, prefix):
        self.stem[0].load_reference_state(state, prefix + "stem.0.")
        self.stem[4].load_reference_state(state, prefix + "stem.4.")
        _load_dropout(self.stem[3], state, prefix + "stem.3.")
        _load_dropout(self.stem[6], state, prefix + "stem.6.")
        if len(self.shortcut):
            self.shortcut[0].load_reference_state(state, prefix + "shortcut.0.")
            _load_dropout(self.shortcut[2], state, prefix + "shortcut.2.")
        self.add = QFunctional(state[prefix + "add.add.scale"], state[prefix + "add.add.zero_point"])

    fuse_post = True       # dropouts and the Add in the convs' epilogues (False: one launch per op, the A/B and recording path)

    def forward(self, x, masks):
        convs = [self.stem[0], self.stem[4]] + ([self.shortcut[0]] if len(self.shortcut) else [])
        if (self.fuse_post and not x.shared and float(self.stem[3].p) > 0.0 and self.stem[0].is_tuned(x)
                and all(0 <= d.mul_mask.zero_point <= 127 for d in self.dropouts())
                and (not len(self.shortcut) or self.shortcut[0].is_tuned(x))):
            m_a = masks.pop(0) if masks is not None else None            # draw order of the reference: stem.3, stem.6, shortcut.2
            m_b = masks.pop(0) if masks is not None else None
            sc = x
            if len(self.shortcut):
                sc = self.shortcut[0].forward_post(x, self.shortcut[2], masks.pop(0) if masks is not None else None)
            out = self.stem[0].forward_post(x, self.stem[3], m_a)
            return self.stem[4].forward_post(out, self.stem[6], m_b, add=(sc, self.add.scale, self.add.zero_point))
        out = self.stem[3](self.stem[0](x), masks.pop(0) if masks is not None else None)
        out = self.stem[6](self.stem[4](out), masks.pop(0) if masks is not None else None)
        sc = x
        if len(self.shortcut):
            sc = self.shortcut[2](self.shortcut[0](x), masks.pop(0) if masks is not None else None)
        return add_relu_q(out, sc, self.add.scale, self.add.zero_point, _a_hi(self.args))


class ConvNetwork_ResNet(nn.Module):
    """reference mcdropout/models_mc.py:162-211 (`conv_resnet_mc`), converted int8 form: deterministic quantised convs with a
    channel dropout after every conv; every conv is its own launch (a dropout sits between each conv and the next)."""

    def __init__(self, input_size, output_size, q, args):
        super().__init__()
        if not q:
            raise NotImplementedError("this class is the converted int8 graph; ModelFactory routes q=False to models_mc_f32.ConvNetwork_ResNet")
        check_bits(args)
        self.args, self.q = args, q
        self.output_size = int(output_size)
        ident = nn.Identity
        self.layers = nn.ModuleList([QConvReLU2d(input_size[1], 24, 3, 1, 1, bias=True, args=args), ident(), ident(), BernoulliDropout(args.p)])
        inp = 24
        for planes, stride in ((24, 1), (48, 2), (96, 2), (192, 2)):
            blocks = []
            for st in (stride, 1):
                blocks.append(BasicBlock(inp, planes, st, q, args))
                inp = planes
            self.layers.append(nn.ModuleList(blocks))
        self.layers.append(ident())      # AvgPool2d(4)
        self.layers.append(ident())      # Flatten
        self.layers.append(QLinear(192, output_size, args=args))
        for i, m in enumerate(self.dropouts()):
            m.layer_id, m.args = i, args
        from .models import QuantStub
        self.quant = QuantStub()

    fuse_blocks = True        # False: one launch per conv (dropout / Add in its epilogue), the A/B and recording path

    def _can_fuse_blocks(self, x, record):
        drops = self.dropouts()
        return (self.fuse_blocks and record is None and tuple(x.shape[1:]) == (3, 32, 32) and float(self.layers[3].p) > 0.0
                and all(float(d.p) > 0.0 and 0 <= d.mul_mask.zero_point <= 127 for d in drops) and _a_hi(self.args) <= 127
                and len(self.layers[4][0].shortcut) == 0 and all(len(self.layers[li][0].shortcut) == 3 for li in (5, 6, 7)))

    def dropouts(self):
        out = [self.layers[3]]
        for li in (4, 5, 6, 7):
            for blk in self.layers[li]:
                out += blk.dropouts()
        return out

    def load_reference_state(self, state):
        self.layers[0].load_reference_state(state, "layers.0.")
        _load_dropout(self.layers[3], state, "layers.3.")
        for li in (4, 5, 6, 7):
            for bi, blk in enumerate(self.layers[li]):
                blk.load_reference_state(state, f"layers.{li}.{bi}.")
        self.layers[10].load_reference_state(state, "layers.10.")
        self.quant.scale = float(np.asarray(state["quant.scale"]).reshape(-1)[0])
        self.quant.zero_point = int(np.asarray(state["quant.zero_point"]).reshape(-1)[0])
        return self

    def forward_mc(self, x, record=None, masks=None):
        """All S samples -> softmax probabilities [S, B, classes].  masks: optional list of fp32 [S, B, C] in draw order."""
        if x.device.type != "cuda":
            raise RuntimeError("qbnn models run on an MI355X only (no CPU fallback)")
        S = _MC.samples
        x = x.to(torch.float32).contiguous()
        B, Cc, H, W = x.shape
        a_hi = _a_hi(self.args)
        xq = torch.empty((1, B, H, W, Cc), dtype=torch.uint8, device=x.device)
        _lib.check(_lib.lib().qbnn_quantize_input_nchw(_lib.ptr(x), B, Cc, H, W, self.quant.scale, self.quant.zero_point, a_hi, _lib.ptr(xq),
                                                       _lib.current_stream()))
        masks = list(masks) if masks is not None else None
        h = MCQTensor(xq, self.quant.scale, self.quant.zero_point, shared=True)
        if self._can_fuse_blocks(x, record):
            # whole BasicBlocks per launch (csrc/qbnn_blocks.hip, DROP kernels): dropouts, Add and ReLU in the convs' epilogues,
            # activations in LDS between a block's convs; layers.0 + layers.3 run inside the layer-1 kernel on the 27-tap patches
            col = torch.empty((B, H * W, 32), dtype=torch.int8, device=x.device)
            _lib.check(_lib.lib().qbnn_im2col3x3_c3(_lib.ptr(xq), B, H, W, self.quant.zero_point, _lib.ptr(col), _lib.current_stream()))
            h = run_identity_chain_drop(list(self.layers[4]), None, masks, stem=(self.layers[0], self.layers[3], col, self.quant.scale))
            for li in (5, 6, 7):
                h = run_down_block_drop(self.layers[li][0], h, masks)
                h = run_identity_chain_drop([self.layers[li][1]], h, masks)
        else:
            h = self.layers[3](self.layers[0](h), masks.pop(0) if masks is not None else None)
            if record is not None:
                record["layers.3.out"] = h.data
            for li in (4, 5, 6, 7):
                for bi, blk in enumerate(self.layers[li]):
                    h = blk(h, masks)
                    if record is not None:
                        record[f"layers.{li}.{bi}.out"] = h.data
        fc = self.layers[10]
        dev = fc._device_params(x.device, fc._weight.int_repr())
        probs = torch.empty((S, B, self.output_size), dtype=torch.float32, device=x.device)
        d = _lib.HeadDesc()
        d.B, d.k, d.C, d.N = B, h.data.shape[2], h.data.shape[4], self.output_size
        d.s_x, d.z_x = h.scale, h.zero_point
        d.s_w, d.z_w = fc._weight.q_scale(), fc._weight.q_zero_point()
        d.s_y, d.z_y = fc.scale, fc.zero_point
        d.a_hi, d.has_bias = a_hi, int(dev["bias"] is not None)
        with timed("head_i8"):
            _lib.check(_lib.lib().qbnn_head_i8_mc(_lib.ptr(h.data), h.sample_stride(), _lib.ptr(dev["w"]), 0, _lib.ptr(dev["bias"]),
                                                  _lib.ptr(probs), S, C.byref(d), _lib.current_stream()))
        return probs

    def forward(self, x):
        with mc_context(1, _MC.seed, _MC.sample_begin, _MC.eps):
            return self.forward_mc(x)[0]
