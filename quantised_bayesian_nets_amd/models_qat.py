"""QAT fake-quant evaluation with LIVE observers behind the reference's API.

Mirror of reference src/models/stochastic/bbb/quantized/conv_qat.py (`Conv2d._forward` :26-49, `ConvBn2d._forward`
:139-167, `ConvBnReLU2d` / `ConvReLU2d` :228-251) and linear_qat.py (`Linear._forward` :18-41, `LinearReLU` :78-79), as
assembled by quant_utils.prepare_model (:109-147): every tensor of the eval-mode forward passes a FakeQuantize whose
MovingAverageMinMaxObserver keeps updating.  The S Monte-Carlo samples are still evaluated together, layer by layer:
the observer recurrence links sample s to the earlier ones only through (min, max), so each FakeQuantize is
    per-sample min/max (parallel)  ->  EMA + qparams over the S samples in order (one small kernel)  ->
    fake-quantise every sample with ITS (scale, zero point)                           (qbnn_observe_f32_mc / qbnn_fake_quant_f32_mc)
and the result equals S sequential reference forwards.  Activations are fp32 NHWC [S, B, H, W, C].

State-dict names follow the reference's prepared model (`<layer>.weight_fake_quant.activation_post_process.min_val`, ...).
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .layers import _MC, bump_state_epoch, mc_context, timed
from .models_f32 import (affine_f32, conv2d_f32, flatten_f32, nchw_to_mc_nhwc, pool2d_f32, sample_conv_weights_f32, softmax_f32)
from .quant import INT_BOUNDS, UINT_BOUNDS, check_bits

OBS_BLOCKS = 512            # QBNN_OBSERVER_BLOCKS (include/qbnn.h)
AVG_CONST = 0.01            # MovingAverageMinMaxObserver default, never overridden by the reference


class FakeQuantize(nn.Module):
    """torch FakeQuantize(observer=MovingAverageMinMaxObserver, per_tensor_affine) with observer and fake-quant both
    enabled (the reference never disables either, so they run in eval).  `state` = (min, max, seen) lives on the device."""

    def __init__(self, qmin, qmax):
        super().__init__()
        self.qmin, self.qmax = int(qmin), int(qmax)
        self.register_buffer("state", torch.zeros(3, dtype=torch.float32))
        self.last_scale = None
        self.last_zero_point = None

    def load(self, st, prefix):
        self._key = prefix                      # where this observer lives in the reference's prepared state dict
        mn, mx = float(np.asarray(st[prefix + ".activation_post_process.min_val"])), float(np.asarray(st[prefix + ".activation_post_process.max_val"]))
        seen = np.isfinite(mn) and np.isfinite(mx)
        self.state = torch.tensor([mn if seen else 0.0, mx if seen else 0.0, 1.0 if seen else 0.0], dtype=torch.float32)

    def min_max(self):
        s = self.state.detach().cpu().numpy()
        return (float(s[0]), float(s[1])) if s[2] else (float("inf"), float("-inf"))

    def forward(self, x, partials=None, relu=False, f32_out=True):
        """x [S or 1, ...] fp32 on the GPU -> [S, ...]: sample s is quantised with the qparams the observer holds after
        having seen samples 0..s (a shared input is observed S times, as S reference forwards would).
        partials = (buffer, n_blocks): per-workgroup (min, max) the producing conv already wrote -- the min/max pass is skipped.
        relu: the ReLU that follows this FakeQuantize in the graph, applied in the same pass (a ReLU of grid values stays on the grid).
        f32_out = False (round 6): every consumer of the result takes its grid integers (`_q8`, with `_grid` = the per-sample scale) -- a conv on
        the int8 pipe, add_q8 -- so the fp32 tensor is allocated for its shape but NEVER WRITTEN (`_no_f32`); honoured only where `_q8` exists."""
        if x.device.type != "cuda":
            raise RuntimeError("qbnn layers run on an MI355X only (no CPU fallback)")
        S = _MC.samples
        x = x.contiguous()
        n = x[0].numel()
        xs = 0 if x.shape[0] == 1 else n
        if self.state.device != x.device:
            self.state = self.state.to(x.device)
        scale = torch.empty(S, dtype=torch.float32, device=x.device)
        zp = torch.empty(S, dtype=torch.int32, device=x.device)
        L = _lib.lib()
        with timed("observe_f32"):
            if partials is not None and x.shape[0] == S:
                _lib.check(L.qbnn_observe_partials_f32_mc(_lib.ptr(partials[0]), partials[1], S, _lib.ptr(self.state), AVG_CONST, self.qmin,
                                                          self.qmax, _lib.ptr(scale), _lib.ptr(zp), _lib.current_stream()))
            else:
                ws = torch.empty(S * OBS_BLOCKS * 2, dtype=torch.float32, device=x.device)
                _lib.check(L.qbnn_observe_f32_mc(_lib.ptr(x), xs, n, S, _lib.ptr(self.state), AVG_CONST, self.qmin, self.qmax, _lib.ptr(ws),
                                                 _lib.ptr(scale), _lib.ptr(zp), _lib.current_stream()))
        y = torch.empty((S,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
        q8 = None
        # q - z spans +-(qmax - qmin): an int8 holds it for grids of at most 128 steps (the 2- to 7-bit activation grids, quant_utils.py:120)
        int8_grid = self.qmax - self.qmin <= 127
        if qat_i8_enabled() and int8_grid:      # an activation grid: leave the integers q - z for a consumer conv on the int8 pipe
            q8 = torch.empty((S, n), dtype=torch.int8, device=x.device)
        skip_f32 = q8 is not None and not f32_out
        with timed("fake_quant_f32"):
            if q8 is not None or relu:
                _lib.check(L.qbnn_fake_quant_ex_f32_mc(_lib.ptr(x), xs, None if skip_f32 else _lib.ptr(y), n, n, _lib.ptr(scale), _lib.ptr(zp), self.qmin,
                                                       self.qmax, int(relu), _lib.ptr(q8), S, _lib.current_stream()))
            else:
                _lib.check(L.qbnn_fake_quant_f32_mc(_lib.ptr(x), xs, _lib.ptr(y), n, n, _lib.ptr(scale), _lib.ptr(zp), 1, self.qmin, self.qmax, S,
                                                    _lib.current_stream()))
        self.last_scale, self.last_zero_point = scale, zp
        # y[s] holds integers times scale[s]: what conv2d_q8 (the int8 matrix pipe) needs to know about its ACTIVATION operand -- only where
        # those integers fit an int8 (a wider grid sends the consumer conv down the fp64 path; the weight operand goes by weight_grid())
        y._grid = scale if int8_grid else None
        y._q8 = q8                   # ... and the integers themselves, [S, n] int8 in y's own element order
        y._no_f32 = skip_f32         # y's own storage holds nothing: fp32 consumers must not touch it (need_f32)
        return y

    def forward_add(self, a, b, relu=False, f32_out=True):
        """FakeQuantize(a + b) for two grid tensors (BasicBlock's Add, src/utils.py:49-55) WITHOUT the fp32 sum tensor (round 6): one pass over the
        operands' integers for the observer's (min, max) (add_q8, sums not stored), the scan, then qbnn_fake_quant_add_q8_mc recomputes each sum --
        the same two fp32 products and one fp32 add -- and quantises it.  Bit-identical to forward(add_q8(a, b)[0], ...)."""
        S, n = a.shape[0], a[0].numel()
        dev = a.device
        if self.state.device != dev:
            self.state = self.state.to(dev)
        L = _lib.lib()
        _, (partials, nblk) = add_q8(a, b, want_sum=False)
        scale = torch.empty(S, dtype=torch.float32, device=dev)
        zp = torch.empty(S, dtype=torch.int32, device=dev)
        with timed("observe_f32"):
            _lib.check(L.qbnn_observe_partials_f32_mc(_lib.ptr(partials), nblk, S, _lib.ptr(self.state), AVG_CONST, self.qmin, self.qmax, _lib.ptr(scale),
                                                      _lib.ptr(zp), _lib.current_stream()))
        y = torch.empty(tuple(a.shape), dtype=torch.float32, device=dev)
        q8 = torch.empty((S, n), dtype=torch.int8, device=dev)
        with timed("fake_quant_add_q8"):
            _lib.check(L.qbnn_fake_quant_add_q8_mc(_lib.ptr(a._q8), n, _lib.ptr(a._grid), _lib.ptr(b._q8), n, _lib.ptr(b._grid), _lib.ptr(y) if f32_out else None, n, n,
                                                   _lib.ptr(scale), _lib.ptr(zp), self.qmin, self.qmax, int(relu), _lib.ptr(q8), S, _lib.current_stream()))
        self.last_scale, self.last_zero_point = scale, zp
        y._grid, y._q8, y._no_f32 = scale, q8, not f32_out
        return y


def _bounds(args):
    check_bits(args)                 # quant_utils.py:120-121: 2..7-bit activations, 2..8-bit weights (every QAT model constructor comes through here)
    return UINT_BOUNDS[args.activation_precision], INT_BOUNDS[args.weight_precision]


def prepared_state(model):
    """The model's state in the reference's prepared-model vocabulary (the dict it was loaded from), with every observer's
    (min, max) as it stands NOW -- after the live-observer evaluations run so far.  This is what `convert.convert_model_state`
    (the reference's quant_utils.convert, :62-99) turns into the int8 model."""
    st = dict(model._prepared)
    for m in model.modules():
        if isinstance(m, FakeQuantize) and getattr(m, "_key", None) is not None:
            mn, mx = m.min_max()
            st[m._key + ".activation_post_process.min_val"] = np.float32(mn)
            st[m._key + ".activation_post_process.max_val"] = np.float32(mx)
    return st


def need_f32(x):
    """Guard of every fp32 consumer: a FakeQuantize output produced with f32_out = False has no fp32 values."""
    if getattr(x, "_no_f32", False):
        raise RuntimeError("qbnn QAT: this tensor was produced as grid integers only (f32_out=False); its fp32 storage was never written")
    return x


def add_q8(a, b, want_sum=True):
    """out + shortcut of two grid tensors from their integers: (fp32 sum [S, ...], (min / max partials, workgroups)) -- qbnn_add_q8_f32_mc.
    want_sum = False: only the partials (the sum tensor is None; FakeQuantize.forward_add recomputes the sums)."""
    S, n = a.shape[0], a[0].numel()
    L = _lib.lib()
    y = torch.empty(tuple(a.shape), dtype=torch.float32, device=a.device) if want_sum else None
    nblk = int(L.qbnn_add_q8_blocks(n))
    partials = torch.empty(S * nblk * 2, dtype=torch.float32, device=a.device)
    with timed("add_q8"):
        _lib.check(L.qbnn_add_q8_f32_mc(_lib.ptr(a._q8), n, _lib.ptr(a._grid), _lib.ptr(b._q8), n, _lib.ptr(b._grid), _lib.ptr(y), n, n, S,
                                        _lib.ptr(partials), _lib.current_stream()))
    return y, (partials, nblk)


def _grid_pair(a, b):
    """Both operands carry their grid integers for all S samples in the same element order."""
    return (getattr(a, "_q8", None) is not None and getattr(b, "_q8", None) is not None and getattr(a, "_grid", None) is not None
            and getattr(b, "_grid", None) is not None and a.shape == b.shape and a._q8.shape == b._q8.shape)


def block_add(fq, out, sc, f32_out=True, from_integers=True):
    """BasicBlock's Add -> FakeQuantize -> ReLU (`end`; the ReLU in the fake-quantiser's pass) for the block's two branches.
    from_integers = False: the operands' fp32 values are not (integer * `_grid`) in ONE rounding -- behind a BernoulliDropout they are
    fl(fl(m s) gain), while `_grid` holds fl(s gain) -- so the Add must read them."""
    if from_integers and _grid_pair(out, sc) and fq.qmax - fq.qmin <= 127:
        if os.environ.get("QBNN_QAT_ADDFQ", "1") != "0":      # Add -> observer -> FakeQuantize from the integers: the fp32 sum is never stored
            return fq.forward_add(out, sc, relu=True, f32_out=f32_out)
        z, mm = add_q8(out, sc)                               # (A/B: the sum as a tensor, its (min, max) for the observer in the same pass)
        return fq(z, partials=mm, relu=True, f32_out=f32_out)
    return fq(affine_f32(need_f32(out), res=need_f32(sc)), relu=True, f32_out=f32_out)


def qat_i8_enabled():
    """QAT convs / linears on the int8 matrix pipe where both operands are fake-quantised tensors (QBNN_QAT_I8=0: the fp64 sums)."""
    return os.environ.get("QBNN_QAT_I8", "1") != "0"


def keep_grid(dst, src, gain=None):
    """`dst` holds the same grid integers as `src` (ReLU, max-pooling, Flatten, a reshape), times `gain` if given (the dropout's 1 / (1 - p))."""
    g = getattr(src, "_grid", None)
    if g is not None:
        dst._grid = g if gain is None else g * float(gain)
        if dst.shape == src.shape and gain is not None:      # an elementwise gain: the same integers in the same order
            dst._q8 = getattr(src, "_q8", None)
    return dst


def weights_to_i8(W, s_w, z_w):
    """The raw integers q_w = rne(W / s_w) + z_w of a fake-quantised weight tensor W [S, n] (per-sample qparams), int8 [S, n]."""
    S = s_w.shape[0]
    n_w = W.shape[1]
    wq = torch.empty((S, n_w), dtype=torch.int8, device=W.device)
    with timed("grid_to_i8"):
        _lib.check(_lib.lib().qbnn_grid_to_i8_mc(_lib.ptr(W), 0 if W.shape[0] == 1 else n_w, n_w, _lib.ptr(s_w), _lib.ptr(z_w), _lib.ptr(wq), S,
                                                 _lib.current_stream()))
    return wq


def conv2d_q8(x, s_x, W, s_w, z_w, cin, cout, k, stride, pad, relu, bias=None, div=None, bn=None, x_q8=None):
    """conv of two fake-quantised tensors as an exact integer sum (qbnn_grid_to_i8_mc x 2 + qbnn_conv2d_q8_f32_mc): x fp32 [S,B,H,W,Cin] on the grid
    s_x[S], W fp32 [S, Cout*k*k*Cin] (OHWI) on the grid (s_w[S], z_w[S]).  Returns (y fp32 [S,B,Ho,Wo,Cout], (min/max partials, workgroups))."""
    S, B, H, Wd = x.shape[0], x.shape[1], x.shape[2], x.shape[3]
    L = _lib.lib()
    st = _lib.current_stream()
    x = x.contiguous()
    n_x, n_w = x[0].numel(), W.shape[1]
    xq = x_q8                              # left by the producing FakeQuantize (same element order as x)
    if xq is None:
        xq = torch.empty((S, n_x), dtype=torch.int8, device=x.device)
        with timed("grid_to_i8"):
            _lib.check(L.qbnn_grid_to_i8_mc(_lib.ptr(x), n_x, n_x, _lib.ptr(s_x), None, _lib.ptr(xq), S, st))
    for t in (s_x, s_w, z_w):            # (the weight side's qparams may come from a side stream's pipeline: keep them alive for this stream's kernel)
        t.record_stream(torch.cuda.current_stream())
    wq = getattr(W, "_q8", None)          # left by the weight pipeline (weights_to_i8: on its side stream when the weights were presampled)
    if wq is None:
        wq = weights_to_i8(W, s_w, z_w)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (Wd + 2 * pad - k) // stride + 1
    y = torch.empty((S, B, Ho, Wo, cout), dtype=torch.float32, device=x.device)
    nblk = int(L.qbnn_conv2d_q8_blocks(B, H, Wd, cin, cout, k, stride, pad))
    partials = torch.empty(S * nblk * 2, dtype=torch.float32, device=x.device)
    alpha, beta = bn if bn is not None else (None, None)
    with timed("conv2d_q8"):
        _lib.check(L.qbnn_conv2d_q8_f32_mc(_lib.ptr(xq), n_x, _lib.ptr(wq), n_w, _lib.ptr(s_x), _lib.ptr(s_w), _lib.ptr(z_w), _lib.ptr(div), _lib.ptr(bias),
                                           _lib.ptr(alpha), _lib.ptr(beta), _lib.ptr(y), y[0].numel(), B, H, Wd, cin, cout, k, stride, pad, int(relu), S,
                                           _lib.ptr(partials), st))
    return y, (partials, nblk)


class _QATBBB(nn.Module):
    """Shared weight pipeline of conv_qat.Conv2d / linear_qat.Linear in eval:
       w = FQ_w(mu * c), s = FQ_std(softplus(rho) * c), t = FQ_mul(eps * s), W = FQ_add(w + t)."""

    def _init_fq(self, args):
        (alo, ahi), (wlo, whi) = _bounds(args)
        self.weight_fake_quant = FakeQuantize(wlo, whi)
        self.std_fake_quant = FakeQuantize(wlo, whi)
        self.mul_noise = FakeQuantize(wlo, whi)          # FloatFunctional.activation_post_process
        self.add_weight = FakeQuantize(wlo, whi)
        self.activation_post_process = FakeQuantize(alo, ahi)
        self.layer_id = 0
        self._folded = None

    def _folded_params(self, dev):
        """(mu * c, softplus(rho) * c) flat, computed once with the reference's own torch ops."""
        if self._folded is None or self._folded[0].device != dev:
            mu, sg = self.weight.detach().float().cpu(), F.softplus(self.std.detach().float().cpu())
            c = self.scale_factor()
            if c is not None:
                shape = [-1] + [1] * (mu.dim() - 1)
                mu, sg = mu * c.reshape(shape), sg * c.reshape(shape)
            if mu.dim() == 4:      # conv: the mean also in [Cout][k][k][Cin] order (the layout the conv kernel streams)
                mu = mu.permute(0, 2, 3, 1)
            self._folded = (mu.reshape(1, -1).contiguous().to(dev), sg.reshape(1, -1).contiguous().to(dev))
        return self._folded

    def scale_factor(self):
        return None

    def weight_grid(self):
        """The FakeQuantize whose output the conv's weight operand is (its last per-sample qparams describe W's integer grid)."""
        return self.add_weight

    def sampled_weights(self, dev, eps=None):
        pre = getattr(self, "_presampled", None)
        self._presampled = None
        # produced ahead of the activation path on a side stream (presample_weights) -- for THIS MC context on THIS device: an entry left
        # behind by a forward that raised, or drawn under another (samples, seed, first sample index), is dropped, and nothing recorded
        # outside a capture is waited on inside one
        if pre is not None and eps is None and pre[2] == _presample_key(dev) and pre[3] == torch.cuda.is_current_stream_capturing():
            W, ev = pre[0], pre[1]
            if ev is None:          # batch_weights: drawn on this stream, in a static buffer
                return W
            torch.cuda.current_stream().wait_event(ev)
            W.record_stream(torch.cuda.current_stream())
            if getattr(W, "_q8", None) is not None:
                W._q8.record_stream(torch.cuda.current_stream())
            return W
        S = _MC.samples
        mu0, sg0 = self._folded_params(dev)
        w = self.weight_fake_quant(mu0)                                  # [S, n]
        s = self.std_fake_quant(sg0)
        n = w.shape[1]
        if self.weight.dim() == 4:
            # conv: w (fake-quantised mean) is already [Cout][k][k][Cin]; the noise term is drawn on the reference's element
            # order and written in that layout too, so everything downstream is elementwise
            t_pre = sample_conv_weights_f32(None, s, self.out_channels, self.in_channels, self.k, self.layer_id, eps, sigma_ss=n)
        else:
            t_pre = torch.empty((S, n), dtype=torch.float32, device=dev)
            if eps is not None:
                eps = eps.to(device=dev, dtype=torch.float32).contiguous()
            with timed("sample_weights_f32"):
                _lib.check(_lib.lib().qbnn_sample_weights_f32_strided(None, 0, _lib.ptr(s), n, n, _MC.seed, self.layer_id, _MC.sample_begin, S,
                                                                      _lib.ptr(eps), _lib.ptr(t_pre), _lib.current_stream()))
        t = self.mul_noise(t_pre)
        W = self.add_weight(affine_f32(w.unsqueeze(-1), res=t.unsqueeze(-1)).squeeze(-1))
        if qat_i8_enabled() and self.add_weight.qmin >= -128 and self.add_weight.qmax <= 127:
            W._q8 = weights_to_i8(W, self.add_weight.last_scale, self.add_weight.last_zero_point)      # the conv's weight operand on the int8 pipe
        return W

    def _load_common(self, st, name):
        self.weight.data = torch.from_numpy(np.asarray(st[name + ".weight"], np.float32).copy()).reshape(self.weight.shape)
        self.std.data = torch.from_numpy(np.asarray(st[name + ".std"], np.float32).copy()).reshape(self.std.shape)
        if self.bias is not None:
            self.bias.data = torch.from_numpy(np.asarray(st[name + ".bias"], np.float32).copy())
        self.weight_fake_quant.load(st, name + ".weight_fake_quant")
        self.std_fake_quant.load(st, name + ".std_fake_quant")
        self.mul_noise.load(st, name + ".mul_noise.activation_post_process")
        self.add_weight.load(st, name + ".add_weight.activation_post_process")
        self.activation_post_process.load(st, name + ".activation_post_process")
        self._folded = None


_SIDE_STREAMS = {}          # device index -> side streams


def _presample_in_capture():
    """Inside a stream capture the weight pipelines run IN LINE by default: measured on the QAT ResNet (B = 256, S = 10) the captured pass replays in
    4.36 ms with them in line and 5.02 ms with them forked onto the side streams (QBNN_QAT_PRESAMPLE_CAPTURE=1: every wait / record then happens
    between streams of the same capture) -- a replayed graph launches the ~300 small kernels back to back, and the fork / join edges cost more than the
    overlap gives.  Eager launches keep the side streams (there the host's launch rate is the bound)."""
    return os.environ.get("QBNN_QAT_PRESAMPLE_CAPTURE", "0") == "1"


def _presample_key(dev):
    dev = torch.device(dev)
    return (_MC.samples, _MC.seed, _MC.sample_begin, dev.index if dev.index is not None else torch.cuda.current_device())


def _wbatch_enabled():
    return os.environ.get("QBNN_QAT_WBATCH", "1") != "0"


class _WeightBatch:
    """Static state of qbnn_qat_weights_mc for one (layer list, device, S): the layer descriptors in device memory, the observers' constant
    (min, max) of mu c / sigma c, the workspaces and the output buffers (overwritten by every forward; consumed by the same forward's convs on the
    same stream)."""

    def __init__(self, layers, dev, S):
        import ctypes as C
        self.S, self.n_layers = S, len(layers)
        arr = (_lib.QatWLayer * len(layers))()
        self.keep, self.out = [], []
        blk0 = 0
        for i, m in enumerate(layers):
            mu0, sg0 = m._folded_params(dev)
            fqs = (m.weight_fake_quant, m.std_fake_quant, m.mul_noise, m.add_weight)
            for fq in fqs:
                if fq.state.device != mu0.device:
                    fq.state = fq.state.to(mu0.device)
            n = mu0.numel()
            d = arr[i]
            d.mu, d.sg = mu0.data_ptr(), sg0.data_ptr()
            d.st_w, d.st_s, d.st_m, d.st_a = (fq.state.data_ptr() for fq in fqs)
            d.cmm[0], d.cmm[1], d.cmm[2], d.cmm[3] = float(mu0.min()), float(mu0.max()), float(sg0.min()), float(sg0.max())
            conv = m.weight.dim() == 4
            d.n, d.Cout, d.Cin, d.KS = n, m.weight.shape[0], m.weight.shape[1], (m.k if conv else 0)
            d.layer_id, d.qmin, d.qmax = m.layer_id, m.add_weight.qmin, m.add_weight.qmax
            d.nblk = max(1, min(64, ((n + 3) // 4 + 511) // 512))
            d.blk0 = blk0
            blk0 += d.nblk
            pm = torch.empty(S * d.nblk * 2, dtype=torch.float32, device=dev)
            pa = torch.empty(S * d.nblk * 2, dtype=torch.float32, device=dev)
            W = torch.empty((S, n), dtype=torch.float32, device=dev)
            q8 = torch.empty((S, n), dtype=torch.int8, device=dev)
            sc = torch.empty(S, dtype=torch.float32, device=dev)
            zp = torch.empty(S, dtype=torch.int32, device=dev)
            d.pm, d.pa, d.W, d.q8, d.scale, d.zp = pm.data_ptr(), pa.data_ptr(), W.data_ptr(), q8.data_ptr(), sc.data_ptr(), zp.data_ptr()
            self.keep += [mu0, sg0, pm, pa] + [fq.state for fq in fqs]
            self.out.append((W, q8, sc, zp))
        self.total_blocks = blk0
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.desc = host.to(dev)
        self.ptrs = self.pointers(layers, dev)

    @staticmethod
    def pointers(layers, dev):
        """What the descriptors point at that a reload / a device move would replace."""
        out = []
        for m in layers:
            f = m._folded
            out.append((None if f is None else f[0].data_ptr(), None if f is None else f[1].data_ptr(), m.weight_fake_quant.state.data_ptr(), m.std_fake_quant.state.data_ptr(),
                        m.mul_noise.state.data_ptr(), m.add_weight.state.data_ptr(), m.layer_id))
        return out



def batch_weights(layers, dev):
    """All layers' weight pipelines in four launches (qbnn_qat_weights_mc): leaves each layer's sampled weights where `sampled_weights` picks them up.
    False where the batched form does not apply (injected eps, QBNN_QAT_WBATCH=0, more than 64 samples, a layer of another kind)."""
    S = _MC.samples
    if not _wbatch_enabled() or _MC.eps is not None or S > 64 or not all(isinstance(m, _QATBBB) and hasattr(m, "add_weight") and hasattr(m, "mul_noise") for m in layers):
        return False
    dev = torch.device(dev)
    # the batch lives on the first layer (it dies with the model): (ids of the layers, device index, S) -> _WeightBatch
    cache = layers[0].__dict__.setdefault("_wbatch_cache", {})
    key = (tuple(id(m) for m in layers), dev.index if dev.index is not None else torch.cuda.current_device(), S)
    wb = cache.get(key)
    if wb is None or wb.ptrs != _WeightBatch.pointers(layers, dev):
        if torch.cuda.is_current_stream_capturing():
            return False          # (descriptors go up with a host copy: built by the eager pass that precedes every capture)
        if len(cache) > 4:
            cache.clear()
        wb = cache[key] = _WeightBatch(layers, dev, S)
    with timed("qat_weights"):
        _lib.check(_lib.lib().qbnn_qat_weights_mc(_lib.ptr(wb.desc), wb.n_layers, wb.total_blocks, AVG_CONST, _MC.seed, _MC.sample_begin, S, _lib.current_stream()))
    pkey, capturing = _presample_key(dev), torch.cuda.is_current_stream_capturing()
    for m, (W, q8, sc, zp) in zip(layers, wb.out):
        W._q8 = q8 if (qat_i8_enabled() and m.add_weight.qmin >= -128 and m.add_weight.qmax <= 127) else None
        m.add_weight.last_scale, m.add_weight.last_zero_point = sc, zp
        m._presampled = (W, None, pkey, capturing)
    return True


def presample_weights(layers, dev, n_streams=4):
    """The weight pipelines of all stochastic layers (4 fake-quantisers each: 12 launches of a few microseconds per layer, none of
    which depends on an activation) up front on side streams, so that they run beside the activation path's convs instead of in
    front of each of them.  Each layer's own observers are only touched by its own pipeline: the order across layers is free."""
    from . import layers as _layers
    if batch_weights(layers, dev):
        return
    capturing = torch.cuda.is_current_stream_capturing()
    if (os.environ.get("QBNN_QAT_PRESAMPLE", "1") == "0" or _MC.eps is not None or (capturing and not _presample_in_capture())
            or _layers.PROFILE is not None):      # (profiling pairs events on ONE stream: keep everything in line)
        for m in layers:
            m._presampled = None
        return
    key = _presample_key(dev)
    main = torch.cuda.current_stream()
    streams = _SIDE_STREAMS.setdefault(key[3], [])
    while len(streams) < n_streams:
        streams.append(torch.cuda.Stream(device=key[3]))
    for st in streams[:n_streams]:
        st.wait_stream(main)
    for i, m in enumerate(layers):
        st = streams[i % n_streams]
        with torch.cuda.stream(st):
            m._presampled = None
            W = m.sampled_weights(dev)
            ev = torch.cuda.Event()
            ev.record(st)
            m._presampled = (W, ev, key, capturing)      # (an entry recorded outside a capture is never waited on inside one, and vice versa)


class Conv2d(_QATBBB):
    """conv_qat.Conv2d / ConvReLU2d / ConvBn2d / ConvBnReLU2d (`bn`, `relu` flags).  Constructor follows
    conv_qat.Conv2d(in_channels, out_channels, kernel_size, stride=1, padding=0, ..., bias=False, qconfig=None, args=None)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=False,
                 padding_mode="zeros", qconfig=None, args=None, bn=False, relu=False, eps=1e-5):
        super().__init__()
        k = kernel_size[0] if isinstance(kernel_size, (tuple, list)) else kernel_size
        self.in_channels, self.out_channels, self.k, self.stride, self.padding = in_channels, out_channels, int(k), int(stride), int(padding)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, self.k, self.k).uniform_(-0.01, 0.01), requires_grad=False)
        self.std = nn.Parameter(torch.full((out_channels, in_channels, self.k, self.k), -10.0), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(out_channels), requires_grad=False) if bias else None
        self.args, self.relu = args, relu
        self.bn = None
        if bn:
            from .models_f32 import BatchNorm2d
            self.bn = BatchNorm2d(out_channels, eps)
        self._cb = None
        self._init_fq(args)

    def scale_factor(self):
        if self.bn is None:
            return None
        running_std = torch.sqrt(self.bn.running_var.float().cpu() + self.bn.eps)          # conv_qat.py:140-141
        return self.bn.weight.detach().float().cpu() / running_std

    def forward(self, x, eps=None, f32_out=True):
        dev = x.device
        W = self.sampled_weights(dev, eps)
        gs, wg = getattr(x, "_grid", None), self.weight_grid()
        if qat_i8_enabled() and gs is not None and x.shape[0] == _MC.samples and wg.last_scale is not None and wg.qmin >= -128 and wg.qmax <= 127:
            # both operands are fake-quantised tensors: the conv is an exact integer sum on the int8 matrix pipe (csrc/qbnn_f32.hip, conv2d_q8_kernel)
            if self.bn is None:
                b = None if self.bias is None else self.bias.detach().to(dev).contiguous()
                z, mm = conv2d_q8(x, gs, W, wg.last_scale, wg.last_zero_point, self.in_channels, self.out_channels, self.k, self.stride, self.padding,
                                  self.relu, bias=b, x_q8=getattr(x, "_q8", None))
            else:
                if self._cb is None or self._cb[0].device != dev:
                    self._cb = (self.scale_factor().to(dev).contiguous(), None if self.bias is None else self.bias.detach().to(dev).contiguous())
                z, mm = conv2d_q8(x, gs, W, wg.last_scale, wg.last_zero_point, self.in_channels, self.out_channels, self.k, self.stride, self.padding,
                                  self.relu, bias=self._cb[1], div=self._cb[0], bn=self.bn.coefficients(dev), x_q8=getattr(x, "_q8", None))
            return self.activation_post_process(z, partials=mm, f32_out=f32_out)
        need_f32(x)
        if self.bn is None:
            b = None if self.bias is None else self.bias.detach().to(dev).contiguous()
            z, mm = conv2d_f32(x, W, b, self.in_channels, self.out_channels, self.k, self.stride, self.padding, self.relu, acc64=True,
                               ohwi=True, minmax=True)
        else:
            # conv, Z / scale_factor (+ bias) (:159-161), bn, ReLU: one kernel, each step rounded as the reference rounds it
            if self._cb is None or self._cb[0].device != dev:
                self._cb = (self.scale_factor().to(dev).contiguous(), None if self.bias is None else self.bias.detach().to(dev).contiguous())
            z, mm = conv2d_f32(x, W, self._cb[1], self.in_channels, self.out_channels, self.k, self.stride, self.padding, self.relu, acc64=True,
                               ohwi=True, div=self._cb[0], bn=self.bn.coefficients(dev), minmax=True)
        return self.activation_post_process(z, partials=mm, f32_out=f32_out)

    def load(self, st, name):
        self._load_common(st, name)
        if self.bn is not None:
            from .models_f32 import _load_bn
            _load_bn(self.bn, st, name + ".bn")
        self._cb = None
        return self


class Linear(_QATBBB):
    """linear_qat.Linear / LinearReLU."""

    def __init__(self, in_features, out_features, bias=False, qconfig=None, args=None, relu=False):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features).uniform_(-0.01, 0.01), requires_grad=False)
        self.std = nn.Parameter(torch.full((out_features, in_features), -3.0), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_features).uniform_(-0.01, 0.01), requires_grad=False) if bias else None
        self.args, self.relu = args, relu
        self._init_fq(args)

    def forward(self, x, eps=None, act=None):
        dev = x.device
        W = self.sampled_weights(dev, eps)
        S, B = _MC.samples, x.shape[1]
        y = torch.empty((S, B, self.out_features), dtype=torch.float32, device=dev)
        b = None if self.bias is None else self.bias.detach().to(dev).contiguous()
        a = (1 if self.relu else 0) if act is None else act
        x = x.contiguous()
        gs, wg = getattr(x, "_grid", None), self.weight_grid()
        if a in (0, 1) and qat_i8_enabled() and gs is not None and x.shape[0] == S and wg.last_scale is not None and wg.qmin >= -128 and wg.qmax <= 127:
            y5, mm = conv2d_q8(x.reshape(S, B, 1, 1, self.in_features), gs, W, wg.last_scale, wg.last_zero_point, self.in_features, self.out_features,
                               1, 1, 0, bool(a), bias=b, x_q8=getattr(x, "_q8", None))
            return self.activation_post_process(y5.reshape(S, B, self.out_features), partials=mm)
        if a in (0, 1):      # the 1x1 case of the conv kernel: its workgroups leave the output's (min, max) for the observer
            y5, mm = conv2d_f32(x.reshape(x.shape[0], B, 1, 1, self.in_features), W, b, self.in_features, self.out_features, 1, 1, 0, bool(a),
                                ohwi=True, minmax=True)
            return self.activation_post_process(y5.reshape(S, B, self.out_features), partials=mm)
        with timed("linear_f32"):
            _lib.check(_lib.lib().qbnn_linear_f32_mc(_lib.ptr(x), 0 if x.shape[0] == 1 else x[0].numel(), _lib.ptr(W), W.shape[1], _lib.ptr(b),
                                                     _lib.ptr(y), y[0].numel(), B, self.in_features, self.out_features, a, S, _lib.current_stream()))
        return self.activation_post_process(y)

    def load(self, st, name):
        self._load_common(st, name)
        return self


class QuantStub(nn.Module):
    """torch QuantStub of a prepared model: identity + activation FakeQuantize (forward hook)."""

    def __init__(self, args):
        super().__init__()
        (alo, ahi), _ = _bounds(args)
        self.activation_post_process = FakeQuantize(alo, ahi)

    def forward(self, x, f32_out=True):
        return self.activation_post_process(x, f32_out=f32_out)


class ConvNetwork_LeNet(nn.Module):
    """Prepared (QAT) `conv_lenet_bbb`, eval: quant - conv - maxpool - conv - maxpool - flatten - fc500+relu - fc - softmax."""
    sequential_samples = True      # live EMA observers: sample s depends on samples < s (mc.py refuses to shard these over ranks)

    def __init__(self, input_size, output_size, q, args):
        super().__init__()
        self.args, self.q = args, q
        c0 = input_size[0] if len(input_size) == 3 else input_size[1]
        self.quant = QuantStub(args)
        self.layers = nn.ModuleList([Conv2d(c0, 20, 5, 1, 2, args=args), nn.Identity(), Conv2d(20, 50, 5, 1, 2, args=args), nn.Identity(),
                                     nn.Identity(), Linear(50 * 7 * 7, 500, args=args, relu=True), nn.Identity(), Linear(500, output_size, args=args)])
        for i, m in enumerate(self.stochastic_layers()):
            m.layer_id = i

    def stochastic_layer_names(self):
        return ["layers.0", "layers.2", "layers.5", "layers.7"]

    def stochastic_layers(self):
        return [self.layers[0], self.layers[2], self.layers[5], self.layers[7]]

    def fake_quantizers(self):
        out = [("quant.activation_post_process", self.quant.activation_post_process)]
        for n, m in zip(self.stochastic_layer_names(), self.stochastic_layers()):
            out += [(n + ".weight_fake_quant", m.weight_fake_quant), (n + ".std_fake_quant", m.std_fake_quant),
                    (n + ".mul_noise.activation_post_process", m.mul_noise), (n + ".add_weight.activation_post_process", m.add_weight),
                    (n + ".activation_post_process", m.activation_post_process)]
        return out

    def prepared_state(self):
        return prepared_state(self)

    def load_reference_state(self, st):
        bump_state_epoch()
        self._prepared = st
        self.quant.activation_post_process.load(st, "quant.activation_post_process")
        for n, m in zip(self.stochastic_layer_names(), self.stochastic_layers()):
            m.load(st, n)
        return self

    def forward_mc(self, x):
        batch_weights(self.stochastic_layers(), x.device)         # all four weight pipelines in four launches
        h = self.quant(nchw_to_mc_nhwc(x))
        c = self.layers[0](h)
        h = keep_grid(pool2d_f32(c, 2, avg=False), c)            # max-pooling picks grid values
        c = self.layers[2](h)
        h = keep_grid(pool2d_f32(c, 2, avg=False), c)
        h = keep_grid(flatten_f32(h), h)
        h = self.layers[5](h)
        return softmax_f32(self.layers[7](h))

    def forward(self, x):
        with mc_context(1, _MC.seed, _MC.sample_begin, _MC.eps):
            return self.forward_mc(x)[0]


class LinearNetwork(nn.Module):
    """Prepared (QAT) `linear_bbb`: quant - 3 x (fc100 + relu) - heads mu / log_var -> (mu, exp(log_var))."""
    sequential_samples = True      # live EMA observers: sample s depends on samples < s (mc.py refuses to shard these over ranks)

    def __init__(self, input_size, output_size, q, args):
        super().__init__()
        self.args, self.q = args, q
        self.input_size = 1
        for i in input_size:
            self.input_size *= int(i)
        self.quant = QuantStub(args)
        self.layers = nn.ModuleList([])
        prev = self.input_size
        for _ in range(3):
            self.layers.append(Linear(prev, 100, bias=True, args=args, relu=True))
            self.layers.append(nn.Identity())
            prev = 100
        self.mu = Linear(prev, 1, bias=True, args=args)
        self.log_var = Linear(prev, 1, bias=True, args=args)
        for i, m in enumerate(self.stochastic_layers()):
            m.layer_id = i

    def stochastic_layer_names(self):
        return ["layers.0", "layers.2", "layers.4", "mu", "log_var"]

    def stochastic_layers(self):
        return [self.layers[0], self.layers[2], self.layers[4], self.mu, self.log_var]

    def prepared_state(self):
        return prepared_state(self)

    def load_reference_state(self, st):
        bump_state_epoch()
        self._prepared = st
        self.quant.activation_post_process.load(st, "quant.activation_post_process")
        for n, m in zip(self.stochastic_layer_names(), self.stochastic_layers()):
            m.load(st, n)
        return self

    def forward_mc(self, x):
        batch_weights(self.stochastic_layers(), x.device)         # all five weight pipelines in four launches
        h = self.quant(x.to(torch.float32).reshape(1, x.shape[0], -1))
        for m in (self.layers[0], self.layers[2], self.layers[4]):
            h = m(h)
        mu = self.mu(h)
        lv = self.log_var(h)
        return mu, torch.exp(lv)          # DeQuantStub is the identity on fp32; models_bbb.py:78

    def forward(self, x):
        with mc_context(1, _MC.seed, _MC.sample_begin, _MC.eps):
            mu, var = self.forward_mc(x)
        return mu[0], var[0]


class BasicBlock(nn.Module):
    """Prepared BasicBlock: stem.0 ConvBnReLU2d, stem.3 ConvBn2d, shortcut.0 ConvBn2d (where the shape changes), `add.add`
    FloatFunctional with its own FakeQuantize, `end` ReLU."""

    def __init__(self, in_planes, planes, stride, args):
        super().__init__()
        (alo, ahi), _ = _bounds(args)
        self.stem = nn.ModuleList([Conv2d(in_planes, planes, 3, stride, 1, args=args, bn=True, relu=True), nn.Identity(), nn.Identity(),
                                   Conv2d(planes, planes, 3, 1, 1, args=args, bn=True), nn.Identity()])
        self.shortcut = nn.ModuleList([])
        if stride != 1 or in_planes != planes:
            self.shortcut.append(Conv2d(in_planes, planes, 1, stride, 0, args=args, bn=True))
            self.shortcut.append(nn.Identity())
        self.add = FakeQuantize(alo, ahi)          # add.add.activation_post_process

    def forward(self, x, f32_out=True):
        """f32_out = False: the block's consumers (the next block's convs and Add) take its grid integers."""
        grid_in = getattr(x, "_q8", None) is not None            # then every tensor inside the block travels as grid integers + scale
        out = self.stem[3](self.stem[0](x, f32_out=not grid_in), f32_out=not grid_in)
        sc = self.shortcut[0](x, f32_out=not grid_in) if len(self.shortcut) else x
        return block_add(self.add, out, sc, f32_out)


class ConvNetwork_ResNet(nn.Module):
    """Prepared (QAT) `conv_resnet_bbb`, eval."""
    sequential_samples = True      # live EMA observers: sample s depends on samples < s (mc.py refuses to shard these over ranks)

    def __init__(self, input_size, output_size, q, args):
        super().__init__()
        self.args, self.q = args, q
        self.quant = QuantStub(args)
        self.layers = nn.ModuleList([Conv2d(input_size[1], 24, 3, 1, 1, args=args, bn=True, relu=True), nn.Identity(), nn.Identity()])
        inp = 24
        for planes, stride in ((24, 1), (48, 2), (96, 2), (192, 2)):
            blocks = []
            for st in (stride, 1):
                blocks.append(BasicBlock(inp, planes, st, args))
                inp = planes
            self.layers.append(nn.ModuleList(blocks))
        self.layers.append(nn.Identity())
        self.layers.append(nn.Identity())
        self.layers.append(Linear(192, output_size, args=args))
        for i, (_, m) in enumerate(self.stochastic_named()):
            m.layer_id = i

    def stochastic_named(self):
        out = [("layers.0", self.layers[0])]
        for li in (3, 4, 5, 6):
            for bi, blk in enumerate(self.layers[li]):
                out.append((f"layers.{li}.{bi}.stem.0", blk.stem[0]))
                out.append((f"layers.{li}.{bi}.stem.3", blk.stem[3]))
                if len(blk.shortcut):
                    out.append((f"layers.{li}.{bi}.shortcut.0", blk.shortcut[0]))
        out.append(("layers.9", self.layers[9]))
        return out

    def prepared_state(self):
        return prepared_state(self)

    def load_reference_state(self, st):
        bump_state_epoch()
        self._prepared = st
        self.quant.activation_post_process.load(st, "quant.activation_post_process")
        for n, m in self.stochastic_named():
            m.load(st, n)
        for li in (3, 4, 5, 6):
            for bi, blk in enumerate(self.layers[li]):
                blk.add.load(st, f"layers.{li}.{bi}.add.add.activation_post_process")
        return self

    def forward_mc(self, x):
        presample_weights([m for _, m in self.stochastic_named()], x.device)
        # every tensor between the input FakeQuantize and the last block's Add is consumed as grid integers (convs on the int8 pipe, add_q8):
        # their fp32 forms are never written (f32_out=False; FakeQuantize honours it only where the integers exist, i.e. not with QBNN_QAT_I8=0)
        h = self.layers[0](self.quant(nchw_to_mc_nhwc(x), f32_out=False), f32_out=False)
        for li in (3, 4, 5, 6):
            for bi, blk in enumerate(self.layers[li]):
                h = blk(h, f32_out=(li == 6 and bi == 1))        # the average pool reads fp32
        h = flatten_f32(pool2d_f32(need_f32(h), 4, avg=True))
        return softmax_f32(self.layers[9](h))

    def forward(self, x):
        with mc_context(1, _MC.seed, _MC.sample_begin, _MC.eps):
            return self.forward_mc(x)[0]
