"""The small int8 Bayes-by-backprop graphs behind the reference's model API.

Mirror of reference src/models/stochastic/bbb/models_bbb.py: `ConvNetwork_LeNet` (:98-143) and `LinearNetwork` with
q=True (:32-95) after quant_utils.prepare_model -> convert.  Their channel counts (1/20/50, 13/100) do not fit the MFMA
tiling of the ResNet kernels; they run on the any-geometry kernels with per-sample row-major sampled weights.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib
import ctypes as C

from .layers import LAYOUT_MFMA32, LAYOUT_ROWMAJOR, Conv2d, Linear, LinearReLU, MCQTensor, _MC, mc_context, sample_all_weights, timed
from .models import QuantStub
from .models_mc import MaxPool2dQ
from .quant import UINT_BOUNDS, check_bits


def _quantize(x4, quant, a_hi):
    B, Cc, H, W = x4.shape
    out = torch.empty((1, B, H, W, Cc), dtype=torch.uint8, device=x4.device)
    _lib.check(_lib.lib().qbnn_quantize_input_nchw(_lib.ptr(x4), B, Cc, H, W, quant.scale, quant.zero_point, a_hi, _lib.ptr(out),
                                                   _lib.current_stream()))
    return MCQTensor(out, quant.scale, quant.zero_point, shared=True)


class _SmallBase(nn.Module):
    deterministic = False        # True: an SGHMC ensemble member (standard quantised layers, fixed weights; sgld/models_sgld.py:13-97)

    def _finish(self, args):
        for i, m in enumerate(self.stochastic_layers()):
            m.layer_id = i
            m.layout = LAYOUT_ROWMAJOR
            m.deterministic = self.deterministic
        self.quant = QuantStub()
        self.a_hi = UINT_BOUNDS[args.activation_precision][1]

    def load_reference_state(self, state):
        for n, m in zip(self.stochastic_layer_names(), self.stochastic_layers()):
            m.load_reference_state(state, n + ".")
        self.quant.scale = float(np.asarray(state["quant.scale"]).reshape(-1)[0])
        self.quant.zero_point = int(np.asarray(state["quant.zero_point"]).reshape(-1)[0])
        return self


class ConvNetwork_LeNet(_SmallBase):
    """reference models_bbb.ConvNetwork_LeNet (:98-143), int8: conv5x5 -> maxpool -> conv5x5 -> maxpool -> flatten ->
    LinearReLU(2450, 500) -> Linear(500, classes) -> dequant -> softmax (no ReLU after the convs)."""

    def __init__(self, input_size, output_size, q, args, deterministic=False):
        super().__init__()
        if not q:
            raise NotImplementedError("float conv BBB layers are not built yet")
        check_bits(args)
        self.args, self.q, self.output_size, self.deterministic = args, q, int(output_size), deterministic
        self.init_channels = input_size[0] if len(input_size) == 3 else input_size[1]       # [C,H,W] (bbb) or [1,C,H,W] (sgld)
        ident = nn.Identity
        self.layers = nn.ModuleList([Conv2d(self.init_channels, 20, (5, 5), stride=1, padding=2, bias=False, args=args), MaxPool2dQ(args),
                                     Conv2d(20, 50, (5, 5), stride=1, padding=2, bias=False, args=args), MaxPool2dQ(args),
                                     ident(),                                             # Flatten (NCHW order)
                                     LinearReLU(50 * 7 * 7, 500, bias_=False, args=args), ident(),
                                     Linear(500, output_size, bias_=False, args=args)])
        self._finish(args)

    def stochastic_layers(self):
        return [self.layers[0], self.layers[2], self.layers[5], self.layers[7]]

    def stochastic_layer_names(self):
        return ["layers.0", "layers.2", "layers.5", "layers.7"]

    fast_path = True        # False: every layer on the any-geometry kernel (the recording / A-B path)

    def _can_run_fast(self, x, record):
        l2 = self.layers[2]
        return (self.fast_path and record is None and not self.deterministic and tuple(x.shape[1:]) == (1, 28, 28) and self.a_hi <= 127
                and all(0 <= m.zero_point <= 127 for m in self.stochastic_layers()) and 0 <= self.quant.zero_point <= 127
                and all(m.bias_ is None for m in self.stochastic_layers()))

    def _set_layouts(self, fast):
        l0, l2, l5, l7 = self.stochastic_layers()
        l0.set_layout(LAYOUT_MFMA32 if fast else LAYOUT_ROWMAJOR)      # fast: one fragment tile (krow = k = 25) for qbnn_conv_c1_pool_i8_mc
        if fast:      # fragment layouts of qbnn_conv_pool_drop_i8_mc (krow = k * Cin) and qbnn_linear_i8_mc (krow = K)
            l2.set_layout(LAYOUT_MFMA32, 5 * 20); l5.set_layout(LAYOUT_MFMA32, 50 * 7 * 7); l7.set_layout(LAYOUT_MFMA32, 500)
        else:
            l2.set_layout(LAYOUT_ROWMAJOR); l5.set_layout(LAYOUT_ROWMAJOR); l7.set_layout(LAYOUT_ROWMAJOR)

    def _forward_fast(self, h, S, dev):
        """layers.2 .. softmax on the small networks' own kernels with per-sample (sampled) weights: conv 20 -> 50 + clamp + max-pool + Flatten
        (NHWC rows) in one launch, a pitched NHWC -> NCHW flatten (the stochastic LinearReLU's noise follows the reference's column order),
        two int8 GEMMs, softmax.  Same bits as the any-geometry kernels."""
        L = _lib.lib()
        l2, l5, l7 = self.layers[2], self.layers[5], self.layers[7]
        B = h.data.shape[1]

        def desc(m, x, Bn, H, Cin, Cout, ks, pad, relu):
            c = _lib.ConvDesc()
            c.B, c.H, c.W, c.Cin, c.Cout, c.ksize, c.stride, c.pad = Bn, H, H, Cin, Cout, ks, 1, pad
            c.s_x, c.z_x, c.s_w, c.z_w = x.scale, x.zero_point, m.add_weight.scale, m.add_weight.zero_point
            c.s_y, c.z_y, c.relu, c.a_hi, c.has_bias = m.scale, m.zero_point, int(relu), self.a_hi, 0
            return c

        w2, w5, w7 = l2.sample_weights(dev), l5.sample_weights(dev), l7.sample_weights(dev)
        ld = (7 * 7 * 50 + 15) // 16 * 16
        rows = torch.empty((S, B, ld), dtype=torch.uint8, device=dev)
        c2 = desc(l2, h, B, 14, 20, 50, 5, 2, False)
        with timed("conv_pool_drop_i8 20->50 k5 (sampled)"):
            _lib.check(L.qbnn_conv_pool_drop_i8_mc(_lib.ptr(h.data), h.sample_stride(), _lib.ptr(w2), w2.shape[1], None, _lib.ptr(rows), rows[0].numel(), ld, S,
                                                   C.byref(c2), 1, None, None, None, None, 0.0, 0, _MC.seed, _MC.sample_begin, _lib.current_stream()))
        flat = torch.empty((S, B, ld), dtype=torch.uint8, device=dev)
        with timed("flatten_nchw_rows"):
            _lib.check(L.qbnn_flatten_nchw_rows_mc(_lib.ptr(rows), rows[0].numel(), ld, B, 49, 50, _lib.ptr(flat), flat[0].numel(), ld, S, _lib.current_stream()))
        x5 = MCQTensor(flat, l2.scale, l2.zero_point)
        ld5 = (500 + 15) // 16 * 16
        y5 = torch.empty((S, B, ld5), dtype=torch.uint8, device=dev)
        c5 = desc(l5, x5, B, 1, 2450, 500, 1, 0, True)
        with timed("linear_i8 2450->500 (sampled)"):
            _lib.check(L.qbnn_linear_i8_mc(_lib.ptr(flat), flat[0].numel(), ld, _lib.ptr(w5), w5.shape[1], None, _lib.ptr(y5), y5[0].numel(), ld5, S, C.byref(c5),
                                           None, None, _MC.seed, _MC.sample_begin, _lib.current_stream()))
        x7 = MCQTensor(y5, l5.scale, l5.zero_point)
        y7 = torch.empty((S, B, self.output_size), dtype=torch.uint8, device=dev)
        c7 = desc(l7, x7, B, 1, 500, self.output_size, 1, 0, False)
        with timed("linear_i8 500->%d (sampled)" % self.output_size):
            _lib.check(L.qbnn_linear_i8_mc(_lib.ptr(y5), y5[0].numel(), ld5, _lib.ptr(w7), w7.shape[1], None, _lib.ptr(y7), y7[0].numel(), self.output_size, S,
                                           C.byref(c7), None, None, _MC.seed, _MC.sample_begin, _lib.current_stream()))
        probs = torch.empty((S, B, self.output_size), dtype=torch.float32, device=dev)
        _lib.check(L.qbnn_dequant_softmax_mc(_lib.ptr(y7), y7[0].numel(), B, self.output_size, l7.scale, l7.zero_point, _lib.ptr(probs), S, _lib.current_stream()))
        return probs

    def forward_mc(self, x, record=None):
        if x.device.type != "cuda":
            raise RuntimeError("qbnn models run on an MI355X only (no CPU fallback)")
        S, dev = _MC.samples, x.device
        rec = (lambda k, v: record.__setitem__(k, v)) if record is not None else (lambda k, v: None)
        fast = self._can_run_fast(x, record)
        if not self.deterministic:
            self._set_layouts(fast)
        h = _quantize(x.to(torch.float32).contiguous(), self.quant, self.a_hi); rec("quant.out", h.data)
        if not self.deterministic:
            sample_all_weights(self.stochastic_layers(), dev)
        if fast:
            # layers.0 + layers.1: patches of the shared input once, then one MFMA per 32 pixels and sample; only the pooled map is written
            l0, L = self.layers[0], _lib.lib()
            B = h.data.shape[1]
            col = torch.empty((B, 28 * 28, 32), dtype=torch.int8, device=dev)
            _lib.check(L.qbnn_im2col5x5_c1(_lib.ptr(h.data), B, 28, 28, h.zero_point, _lib.ptr(col), _lib.current_stream()))
            w0 = l0.sample_weights(dev)
            c0 = _lib.ConvDesc()
            c0.B, c0.H, c0.W, c0.Cin, c0.Cout, c0.ksize, c0.stride, c0.pad = B, 28, 28, 1, 20, 5, 1, 2
            c0.s_x, c0.z_x, c0.s_w, c0.z_w = h.scale, h.zero_point, l0.add_weight.scale, l0.add_weight.zero_point
            c0.s_y, c0.z_y, c0.relu, c0.a_hi, c0.has_bias = l0.scale, l0.zero_point, 0, self.a_hi, 0
            p1 = torch.empty((S, B, 14, 14, 20), dtype=torch.uint8, device=dev)
            with timed("conv_c1_pool_i8 1->20 k5 (sampled)"):
                _lib.check(L.qbnn_conv_c1_pool_i8_mc(_lib.ptr(col), 0, _lib.ptr(w0), w0.shape[1], None, _lib.ptr(p1), p1[0].numel(), S, C.byref(c0),
                                                     _lib.current_stream()))
            return self._forward_fast(MCQTensor(p1, l0.scale, l0.zero_point), S, dev)
        h = self.layers[0]._conv(h, self.layers[0].sample_weights(dev), S); rec("layers.0.out", h.data)
        h = self.layers[1](h); rec("layers.1.out", h.data)
        h = self.layers[2]._conv(h, self.layers[2].sample_weights(dev), S); rec("layers.2.out", h.data)
        h = self.layers[3](h); rec("layers.3.out", h.data)
        d = h.data
        _, B, H, W, Cc = d.shape
        flat = torch.empty((S, B, Cc * H * W), dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib().qbnn_flatten_nchw_mc(_lib.ptr(d), h.sample_stride(), B, H * W, Cc, _lib.ptr(flat), flat[0].numel(), S,
                                                   _lib.current_stream()))
        h = MCQTensor(flat, h.scale, h.zero_point)
        h = self.layers[5](h); rec("layers.5.out", h.data)
        h = self.layers[7](h); rec("layers.7.out", h.data)
        probs = torch.empty((S, B, self.output_size), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().qbnn_dequant_softmax_mc(_lib.ptr(h.data), h.sample_stride(), B, self.output_size, h.scale, h.zero_point,
                                                      _lib.ptr(probs), S, _lib.current_stream()))
        return probs

    def forward(self, x):
        with mc_context(1, _MC.seed, _MC.sample_begin, _MC.eps):
            return self.forward_mc(x)[0]


class LinearNetwork(_SmallBase):
    """reference models_bbb.LinearNetwork with q=True (:32-95): QuantStub -> 3 x LinearReLU(100) -> heads mu, log_var ->
    DeQuant -> (mu, exp(log_var))."""

    def __init__(self, input_size, output_size, q, args, deterministic=False):
        super().__init__()
        check_bits(args)
        self.args, self.q, self.deterministic = args, q, deterministic
        self.input_size = 1
        for i in input_size:
            self.input_size *= int(i)
        self.output_size = int(output_size)
        ident = nn.Identity
        self.layers = nn.ModuleList([LinearReLU(self.input_size, 100, bias_=True, args=args), ident(),
                                     LinearReLU(100, 100, bias_=True, args=args), ident(),
                                     LinearReLU(100, 100, bias_=True, args=args), ident()])
        self.mu = Linear(100, 1, bias_=True, args=args)
        self.log_var = Linear(100, 1, bias_=True, args=args)
        self._finish(args)

    def stochastic_layers(self):
        return [self.layers[0], self.layers[2], self.layers[4], self.mu, self.log_var]

    def stochastic_layer_names(self):
        return ["layers.0", "layers.2", "layers.4", "mu", "log_var"]

    def forward_mc(self, x, record=None):
        """-> (mu [S,B,1], var [S,B,1]) fp32."""
        if x.device.type != "cuda":
            raise RuntimeError("qbnn models run on an MI355X only (no CPU fallback)")
        S, dev = _MC.samples, x.device
        rec = (lambda k, v: record.__setitem__(k, v)) if record is not None else (lambda k, v: None)
        x2 = x.to(torch.float32).reshape(x.shape[0], -1)
        h = _quantize(x2.reshape(x2.shape[0], x2.shape[1], 1, 1).contiguous(), self.quant, self.a_hi)
        h = MCQTensor(h.data.reshape(1, x2.shape[0], x2.shape[1]), h.scale, h.zero_point, shared=True); rec("quant.out", h.data)
        if not self.deterministic:
            sample_all_weights(self.stochastic_layers(), dev)
        for i in (0, 2, 4):
            h = self.layers[i](h); rec(f"layers.{i}.out", h.data)
        qm, qv = self.mu(h), self.log_var(h)
        rec("mu.out", qm.data); rec("log_var.out", qv.data)
        return qm.dequantize(), torch.exp(qv.dequantize())

    def forward(self, x):
        with mc_context(1, _MC.seed, _MC.sample_begin, _MC.eps):
            mu, var = self.forward_mc(x)
        return mu[0], var[0]
