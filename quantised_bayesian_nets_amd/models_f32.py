"""fp32 Bayes-by-backprop layers / MLP behind the reference's API (BASELINE config 0).

Mirror of reference src/models/stochastic/bbb/linear.py (`Linear`, :8-50, eval branch) and models_bbb.py
(`LinearNetwork`, :32-90).  Parameters keep the reference names: `weight` (mu), `std` (rho, pre-softplus), `bias`.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .layers import _MC, bump_state_epoch, mc_context, timed


class Linear(nn.Module):
    """reference bbb.linear.Linear(in_features, out_features, bias, sigma_prior=1.0, args=None).
    Eval-mode forward for S MC samples: sigma = softplus(rho); W_s = mu + eps_s * sigma; y_s = x_s @ W_s^T + b."""

    def __init__(self, in_features, out_features, bias, sigma_prior=1.0, args=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features).uniform_(-0.01, 0.01), requires_grad=False)
        self.std = nn.Parameter(torch.full((out_features, in_features), -3.0), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_features).uniform_(-0.01, 0.01), requires_grad=False) if bias else None
        self.std_prior = nn.Parameter(torch.ones((1,)) * sigma_prior, requires_grad=False)
        self.args = args
        self.layer_id = 0
        self._sigma = None

    def get_kl_divergence(self):
        """reference utils_bbb.kl_divergence (bbb/utils_bbb.py:3-5) against N(0, std_prior)."""
        sigma, mu, sp = F.softplus(self.std), self.weight, self.std_prior
        return 0.5 * (2 * torch.log(sp / sigma) - 1 + (sigma / sp).pow(2) + ((0 - mu) / sp).pow(2)).sum()

    def forward(self, x, act=0, eps=None):
        """x fp32 [S or 1, B, in_features] -> [S, B, out_features]."""
        if x.device.type != "cuda":
            raise RuntimeError("qbnn layers run on an MI355X only (no CPU fallback)")
        S = _MC.samples
        if self._sigma is None or self._sigma.device != x.device:
            self._sigma = F.softplus(self.std.detach().cpu().float()).to(x.device).contiguous()   # one-time, same op as the reference
            self._mu_dev = self.weight.detach().to(x.device).contiguous()                        # device copies made once (they were
            self._bias_dev = None if self.bias is None else self.bias.detach().to(x.device).contiguous()   # re-uploaded on every call)
        mu = self._mu_dev
        n = mu.numel()
        w = _take_presampled(self, x.device) if eps is None else None
        if w is None:
            w = torch.empty((S, n), dtype=torch.float32, device=x.device)
            if eps is not None:
                eps = eps.to(device=x.device, dtype=torch.float32).contiguous()
            with timed("sample_weights_f32"):
                _lib.check(_lib.lib().qbnn_sample_weights_f32(_lib.ptr(mu), _lib.ptr(self._sigma), n, _MC.seed, self.layer_id,
                                                              _MC.sample_begin, S, _lib.ptr(eps), _lib.ptr(w), _lib.current_stream()))
        B = x.shape[1]
        y = torch.empty((S, B, self.out_features), dtype=torch.float32, device=x.device)
        xs = 0 if x.shape[0] == 1 else x[0].numel()
        b = self._bias_dev
        with timed("linear_f32"):
            _lib.check(_lib.lib().qbnn_linear_f32_mc(_lib.ptr(x.contiguous()), xs, _lib.ptr(w), n, _lib.ptr(b), _lib.ptr(y), y[0].numel(),
                                                     B, self.in_features, self.out_features, act, S, _lib.current_stream()))
        return y


class LinearNetwork(nn.Module):
    """reference models_bbb.LinearNetwork (float, q=False): 3 x (Linear(100) + ReLU), heads `mu` and `log_var`;
    forward -> (mu, exp(log_var))."""

    def __init__(self, input_size, output_size, q, args):
        super().__init__()
        if q:
            raise NotImplementedError("the quantised MLP is not built yet")
        self.args = args
        self.input_size = 1
        for i in input_size:
            self.input_size *= int(i)
        self.output_size = int(output_size)
        sp = getattr(args, "sigma_prior", 1.0)
        widths = [100, 100, 100]
        self.layers = nn.ModuleList([])
        prev = self.input_size
        for wd in widths:
            self.layers.append(Linear(prev, wd, bias=True, sigma_prior=sp, args=args))
            self.layers.append(nn.ReLU())
            prev = wd
        self.mu = Linear(prev, 1, bias=True, sigma_prior=sp, args=args)
        self.log_var = Linear(prev, 1, bias=True, sigma_prior=sp, args=args)
        for i, m in enumerate(self.stochastic_layers()):
            m.layer_id = i
        self.q = q

    def stochastic_layers(self):
        return [self.layers[0], self.layers[2], self.layers[4], self.mu, self.log_var]

    def stochastic_layer_names(self):
        return ["layers.0", "layers.2", "layers.4", "mu", "log_var"]

    def load_reference_state(self, state):
        for n, m in zip(self.stochastic_layer_names(), self.stochastic_layers()):
            m.weight.data = torch.from_numpy(np.asarray(state[n + ".weight"], np.float32).copy())
            m.std.data = torch.from_numpy(np.asarray(state[n + ".std"], np.float32).copy())
            m.bias.data = torch.from_numpy(np.asarray(state[n + ".bias"], np.float32).copy())
            m._sigma = None
        bump_state_epoch()
        return self

    def get_kl_divergence(self):
        return sum(m.get_kl_divergence() for m in self.stochastic_layers())

    def forward_mc(self, x):
        """All S samples -> (mu [S,B,1], var [S,B,1]).  One sampler launch + one network launch (qbnn_mlp_bbb_f32_mc: weights and
        activations in LDS); QBNN_MLP_LAYERWISE=1 runs a sampler and a GEMM launch per layer instead (same values up to the fp32
        summation order of the dot products)."""
        import os
        if x.device.type != "cuda":
            raise RuntimeError("qbnn models run on an MI355X only (no CPU fallback)")
        x2 = x.to(torch.float32).reshape(x.shape[0], -1).contiguous()
        widths_ok = all(m.in_features <= 128 and m.out_features <= 128 for m in self.stochastic_layers())
        if widths_ok and os.environ.get("QBNN_MLP_LAYERWISE", "0") != "1":
            import ctypes as C
            S, B, dev = _MC.samples, x2.shape[0], x2.device
            arr = (_lib.MlpLayer * 5)()
            keep = []
            for d, m in zip(arr, self.stochastic_layers()):
                if m._sigma is None or m._sigma.device != dev:
                    m._sigma = F.softplus(m.std.detach().cpu().float()).to(dev).contiguous()      # one-time, same op as the reference
                    m._mu_dev = m.weight.detach().to(dev).contiguous()
                    m._bias_dev = None if m.bias is None else m.bias.detach().to(dev).contiguous()
                d.mu, d.sigma, d.bias = m._mu_dev.data_ptr(), m._sigma.data_ptr(), (None if m._bias_dev is None else m._bias_dev.data_ptr())
                d.out_features, d.in_features, d.layer_id = m.out_features, m.in_features, m.layer_id
                keep.append(m)
            L = _lib.lib()
            ws = torch.empty((S, int(L.qbnn_mlp_bbb_f32_workspace_floats(arr))), dtype=torch.float32, device=dev)
            mu = torch.empty((S, B, 1), dtype=torch.float32, device=dev)
            var = torch.empty((S, B, 1), dtype=torch.float32, device=dev)
            with timed("mlp_bbb_f32"):
                _lib.check(L.qbnn_mlp_bbb_f32_mc(_lib.ptr(x2), B, arr, _MC.seed, _MC.sample_begin, S, _lib.ptr(ws), _lib.ptr(mu), _lib.ptr(var),
                                                 _lib.current_stream()))
            return mu, var
        h = x2.reshape(1, x2.shape[0], -1)
        for m in (self.layers[0], self.layers[2], self.layers[4]):
            h = m(h, act=1)
        return self.mu(h, act=0), self.log_var(h, act=2)

    def forward(self, x):
        with mc_context(1, _MC.seed, _MC.sample_begin, _MC.eps):
            mu, var = self.forward_mc(x)
        return mu[0], var[0]


# =====================================================================================================================
# fp32 convolutional Bayes-by-backprop graphs: activations fp32 NHWC [S, B, H, W, C]
# =====================================================================================================================
def _f32(t, dev):
    return None if t is None else t.detach().to(device=dev, dtype=torch.float32).contiguous()


def conv2d_f32(x, w, bias, cin, cout, k, stride, pad, relu, acc64=False, ohwi=False, bn=None, res=None, div=None, minmax=False):
    """x [S|1,B,H,W,Cin], w [S|1, Cout*Cin*k*k] (reference order, or [Cout,k,k,Cin] when `ohwi`) -> [S,B,Ho,Wo,Cout]
    through qbnn_conv2d_f32_mc."""
    S = max(x.shape[0], w.shape[0])
    B, H, W = x.shape[1], x.shape[2], x.shape[3]
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty((S, B, Ho, Wo, cout), dtype=torch.float32, device=x.device)
    alpha, beta = bn if bn is not None else (None, None)      # BatchNorm eval coefficients: fused tail of the conv kernel
    partials, nblk = None, 0
    if minmax:       # QAT: every workgroup leaves the (min, max) of its outputs for the observer that follows the conv
        nblk = int(_lib.lib().qbnn_conv2d_f32_blocks(B, H, W, cout, k, stride, pad))
        partials = torch.empty(S * nblk * 2, dtype=torch.float32, device=x.device)
    with timed("conv2d_f32"):
        _lib.check(_lib.lib().qbnn_conv2d_f32_fused_mc(_lib.ptr(x), 0 if x.shape[0] == 1 else x[0].numel(), _lib.ptr(w),
                                                       0 if w.shape[0] == 1 else w[0].numel(), _lib.ptr(div), _lib.ptr(bias), _lib.ptr(alpha), _lib.ptr(beta),
                                                       _lib.ptr(res), 0 if res is None or res.shape[0] == 1 else res[0].numel(), _lib.ptr(y),
                                                       y[0].numel(), B, H, W, cin, cout, k, stride, pad,
                                                       int(relu) | (2 if acc64 else 0) | (4 if ohwi else 0), S, _lib.ptr(partials),
                                                       _lib.current_stream()))
    if minmax:
        return y, (partials, nblk)
    return y


def affine_f32(x, p0=None, p1=None, res=None, relu=False, mode=0):
    """Per-channel x*p0+p1 (mode 0) or x/p0+p1 (mode 1), + res, ReLU: BatchNorm eval / conv-bn unfolding / Add / ReLU."""
    S, C = x.shape[0], x.shape[-1]
    y = torch.empty_like(x)
    n = x[0].numel()
    with timed("affine_f32"):
        _lib.check(_lib.lib().qbnn_affine_f32_mc(_lib.ptr(x), n, _lib.ptr(res), 0 if res is None or res.shape[0] == 1 else n, _lib.ptr(p0),
                                                 _lib.ptr(p1), _lib.ptr(y), n, n, C, mode, int(relu), S, _lib.current_stream()))
    return y


def pool2d_f32(x, k, avg):
    S, B, H, W, C = x.shape
    y = torch.empty((S, B, H // k, W // k, C), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().qbnn_pool2d_f32_mc(_lib.ptr(x), x[0].numel(), _lib.ptr(y), y[0].numel(), B, H, W, C, k, int(avg), S,
                                             _lib.current_stream()))
    return y


def flatten_f32(x):
    """Reference Flatten on the NCHW view: [S,B,H,W,C] -> [S,B,C*H*W]."""
    S, B, H, W, C = x.shape
    y = torch.empty((S, B, C * H * W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().qbnn_flatten_nchw_f32_mc(_lib.ptr(x), x[0].numel(), B, H * W, C, _lib.ptr(y), y[0].numel(), S, _lib.current_stream()))
    return y


def softmax_f32(x):
    S, B, N = x.shape
    p = torch.empty((S, B, N), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().qbnn_softmax_f32_mc(_lib.ptr(x.contiguous()), B * N, B, N, _lib.ptr(p), S, _lib.current_stream()))
    return p


def sample_weights_f32(mu, sigma, layer_id, eps=None):
    """W[s] = mu + eps_s * sigma for the S samples of the active mc_context (flat, reference element order)."""
    S, n = _MC.samples, mu.numel()
    w = torch.empty((S, n), dtype=torch.float32, device=mu.device)
    if eps is not None:
        eps = eps.to(device=mu.device, dtype=torch.float32).contiguous()
    with timed("sample_weights_f32"):
        _lib.check(_lib.lib().qbnn_sample_weights_f32(_lib.ptr(mu), _lib.ptr(sigma), n, _MC.seed, layer_id, _MC.sample_begin, S,
                                                      _lib.ptr(eps), _lib.ptr(w), _lib.current_stream()))
    return w


def sample_conv_weights_f32(mu, sigma, cout, cin, k, layer_id, eps=None, mu_ss=0, sigma_ss=0):
    """As sample_weights_f32 for a conv weight (noise indexed in the reference's [Cout][Cin][k][k] order), written
    [Cout][k][k][Cin]; mu / sigma may be per-sample ([S, n], stride n) or shared (stride 0); mu None = noise term only."""
    S, n = _MC.samples, cout * cin * k * k
    w = torch.empty((S, n), dtype=torch.float32, device=sigma.device)
    if eps is not None:
        eps = eps.to(device=sigma.device, dtype=torch.float32).contiguous()
    with timed("sample_weights_f32"):
        _lib.check(_lib.lib().qbnn_sample_weights_f32_ohwi(_lib.ptr(mu), mu_ss, _lib.ptr(sigma), sigma_ss, cout, cin, k, _MC.seed, layer_id,
                                                           _MC.sample_begin, S, _lib.ptr(eps), _lib.ptr(w), _lib.current_stream()))
    return w


def nchw_to_mc_nhwc(x):
    """[B,C,H,W] fp32 input -> [1,B,H,W,C] (shared by all samples)."""
    return x.to(torch.float32).permute(0, 2, 3, 1).contiguous().unsqueeze(0)


class _WeightBatchF32:
    """Static state of qbnn_sample_weights_f32_batch for one (layer list, device, S): device copies of mu / softplus(rho), the layer descriptors in
    device memory and the [S, n] output buffers (overwritten by every forward, consumed by the same forward's convs on the same stream)."""

    def __init__(self, layers, dev, S):
        arr = (_lib.F32WLayer * len(layers))()
        self.keep, self.out, self.n_layers = [], [], len(layers)
        blk0 = 0
        for i, m in enumerate(layers):
            mu = m.weight.detach().to(device=dev, dtype=torch.float32).contiguous().reshape(-1)
            sg = F.softplus(m.std.detach().cpu().float()).to(dev).contiguous().reshape(-1)          # the reference's own op, once
            n = mu.numel()
            W = torch.empty((S, n), dtype=torch.float32, device=dev)
            d = arr[i]
            d.mu, d.sigma, d.w = mu.data_ptr(), sg.data_ptr(), W.data_ptr()
            conv = m.weight.dim() == 4
            d.n, d.Cout, d.Cin, d.KS = n, m.weight.shape[0], m.weight.shape[1], (m.k if conv else 0)
            d.layer_id = m.layer_id
            d.nblk = max(1, min(128, ((n + 3) // 4 + 255) // 256))
            d.blk0 = blk0
            blk0 += d.nblk
            self.keep += [mu, sg]
            self.out.append(W)
        self.total_blocks = blk0
        self.desc = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.versions = self.param_versions(layers)

    @staticmethod
    def param_versions(layers):
        return [(m.weight.data_ptr(), m.weight._version, m.std.data_ptr(), m.std._version, m.layer_id) for m in layers]


def batch_weights_f32(layers, dev):
    """Every layer's W_s = mu + eps_s * softplus(rho) for the S samples of the active mc_context in ONE launch; each layer's forward then picks its
    weights up.  False where it does not apply (injected eps, QBNN_F32_WBATCH=0)."""
    if os.environ.get("QBNN_F32_WBATCH", "1") == "0" or _MC.eps is not None or not layers:
        return False
    dev = torch.device(dev)
    S = _MC.samples
    cache = layers[0].__dict__.setdefault("_wbatch_cache", {})
    key = (tuple(id(m) for m in layers), dev.index if dev.index is not None else torch.cuda.current_device(), S)
    wb = cache.get(key)
    if wb is None or wb.versions != _WeightBatchF32.param_versions(layers):
        if torch.cuda.is_current_stream_capturing():
            return False          # (descriptors go up with a host copy: built by the eager pass that precedes every capture)
        if len(cache) > 4:
            cache.clear()
        wb = cache[key] = _WeightBatchF32(layers, dev, S)
    with timed("sample_weights_f32"):
        _lib.check(_lib.lib().qbnn_sample_weights_f32_batch(_lib.ptr(wb.desc), wb.n_layers, wb.total_blocks, _MC.seed, _MC.sample_begin, S, _lib.current_stream()))
    tag = (_MC.samples, _MC.seed, _MC.sample_begin, key[1])
    for m, W in zip(layers, wb.out):
        m._presampled = (W, tag)
    return True


def _take_presampled(m, dev):
    """The weights batch_weights_f32 drew for THIS MC context on this device, or None."""
    pre = m.__dict__.pop("_presampled", None)
    dev = torch.device(dev)
    if pre is not None and pre[1] == (_MC.samples, _MC.seed, _MC.sample_begin, dev.index if dev.index is not None else torch.cuda.current_device()) and _MC.eps is None:
        return pre[0]
    return None


class Conv2d(nn.Module):
    """reference bbb.conv.Conv2d(in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
    bias=False, padding_mode='zeros', sigma_prior=-2, args=None), eval branch (conv.py:33-39)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=False,
                 padding_mode="zeros", sigma_prior=-2, args=None):
        super().__init__()
        k = kernel_size[0] if isinstance(kernel_size, (tuple, list)) else kernel_size
        if dilation != 1 or groups != 1 or padding_mode != "zeros":
            raise NotImplementedError("dilation / groups / padding modes other than the reference's defaults")
        self.in_channels, self.out_channels, self.k, self.stride, self.padding = in_channels, out_channels, int(k), int(stride), int(padding)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, self.k, self.k).uniform_(-0.01, 0.01), requires_grad=False)
        self.std = nn.Parameter(torch.full((out_channels, in_channels, self.k, self.k), -10.0), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(out_channels), requires_grad=False) if bias else None
        self.std_prior = nn.Parameter(torch.ones((1,)) * sigma_prior, requires_grad=False)
        self.args = args
        self.layer_id = 0
        self._sigma = None

    def get_kl_divergence(self):
        sigma, mu, sp = F.softplus(self.std), self.weight, self.std_prior
        return 0.5 * (2 * torch.log(sp / sigma) - 1 + (sigma / sp).pow(2) + ((0 - mu) / sp).pow(2)).sum()

    def forward(self, x, relu=False, eps=None, bn=None, res=None):
        """bn: a BatchNorm2d (eval) applied to the conv output, res: a tensor added after it, relu: final ReLU -- all in the
        conv kernel's epilogue, rounded step by step like the separate modules of the reference graph."""
        if x.device.type != "cuda":
            raise RuntimeError("qbnn layers run on an MI355X only (no CPU fallback)")
        w = _take_presampled(self, x.device) if eps is None else None
        if w is None:
            if self._sigma is None or self._sigma.device != x.device:
                self._sigma = F.softplus(self.std.detach().cpu().float()).to(x.device).contiguous().reshape(-1)
            w = sample_conv_weights_f32(_f32(self.weight, x.device).reshape(-1), self._sigma, self.out_channels, self.in_channels, self.k,
                                        self.layer_id, eps)
        return conv2d_f32(x, w, _f32(self.bias, x.device), self.in_channels, self.out_channels, self.k, self.stride, self.padding, relu,
                          ohwi=True, bn=None if bn is None else bn.coefficients(x.device), res=res)


class BatchNorm2d(nn.Module):
    """nn.BatchNorm2d in eval mode: y = x * alpha + beta with alpha = weight / sqrt(running_var + eps),
    beta = bias - running_mean * alpha (the two-rounding form ATen's CPU kernel evaluates)."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.weight = nn.Parameter(torch.ones(num_features), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(num_features), requires_grad=False)
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self._ab = None

    def coefficients(self, dev):
        if self._ab is None or self._ab[0].device != dev:
            invstd = 1.0 / torch.sqrt(self.running_var.float().cpu() + self.eps)
            alpha = self.weight.detach().float().cpu() * invstd
            beta = self.bias.detach().float().cpu() - self.running_mean.float().cpu() * alpha
            self._ab = (alpha.to(dev).contiguous(), beta.to(dev).contiguous())
        return self._ab

    def forward(self, x, relu=False, res=None):
        a, b = self.coefficients(x.device)
        return affine_f32(x, a, b, res=res, relu=relu)


def _load_bbb(m, state, name):
    m.weight.data = torch.from_numpy(np.asarray(state[name + ".weight"], np.float32).copy()).reshape(m.weight.shape)
    m.std.data = torch.from_numpy(np.asarray(state[name + ".std"], np.float32).copy()).reshape(m.std.shape)
    if getattr(m, "bias", None) is not None and name + ".bias" in state:
        m.bias.data = torch.from_numpy(np.asarray(state[name + ".bias"], np.float32).copy())
    m._sigma = None
    bump_state_epoch()


def _load_bn(m, state, name):
    for k in ("weight", "bias"):
        getattr(m, k).data = torch.from_numpy(np.asarray(state[f"{name}.{k}"], np.float32).copy())
    m.running_mean = torch.from_numpy(np.asarray(state[name + ".running_mean"], np.float32).copy())
    m.running_var = torch.from_numpy(np.asarray(state[name + ".running_var"], np.float32).copy())
    m._ab = None
    bump_state_epoch()


class ConvNetwork_LeNet(nn.Module):
    """reference models_bbb.ConvNetwork_LeNet with q=False (models_bbb.py:100-133): conv5x5(p2) - maxpool2 - conv5x5(p2) -
    maxpool2 - flatten - fc500 - relu - fc - softmax; no bias anywhere, no ReLU after the convs."""

    def __init__(self, input_size, output_size, q, args):
        super().__init__()
        self.args, self.q = args, q
        sp = getattr(args, "sigma_prior", -2)
        c0 = input_size[0] if len(input_size) == 3 else input_size[1]
        self.layers = nn.ModuleList([Conv2d(c0, 20, 5, 1, 2, sigma_prior=sp, args=args), nn.Identity(),
                                     Conv2d(20, 50, 5, 1, 2, sigma_prior=sp, args=args), nn.Identity(), nn.Identity(),
                                     Linear(50 * 7 * 7, 500, bias=False, sigma_prior=sp, args=args), nn.ReLU(),
                                     Linear(500, output_size, bias=False, sigma_prior=sp, args=args)])
        for i, m in enumerate(self.stochastic_layers()):
            m.layer_id = i

    def stochastic_layer_names(self):
        return ["layers.0", "layers.2", "layers.5", "layers.7"]

    def stochastic_layers(self):
        return [self.layers[0], self.layers[2], self.layers[5], self.layers[7]]

    def load_reference_state(self, state):
        for n, m in zip(self.stochastic_layer_names(), self.stochastic_layers()):
            _load_bbb(m, state, n)
        return self

    def get_kl_divergence(self):
        return sum(m.get_kl_divergence() for m in self.stochastic_layers())

    def forward_mc(self, x):
        batch_weights_f32(self.stochastic_layers(), x.device)      # the four layers' weight draws in one launch
        h = nchw_to_mc_nhwc(x)
        h = pool2d_f32(self.layers[0](h), 2, avg=False)
        h = pool2d_f32(self.layers[2](h), 2, avg=False)
        h = flatten_f32(h)
        h = self.layers[5](h, act=1)
        return softmax_f32(self.layers[7](h, act=0))

    def forward(self, x):
        with mc_context(1, _MC.seed, _MC.sample_begin, _MC.eps):
            return self.forward_mc(x)[0]


class BasicBlock(nn.Module):
    """reference models_bbb.BasicBlock (float): stem conv-bn-relu-conv-bn, optional 1x1/s conv-bn shortcut, Add, ReLU."""
    expansion = 1

    def __init__(self, in_planes, planes, stride=1, q=False, args=None):
        super().__init__()
        sp = getattr(args, "sigma_prior", -2)
        self.stem = nn.ModuleList([Conv2d(in_planes, planes, 3, stride, 1, sigma_prior=sp, args=args), BatchNorm2d(planes), nn.ReLU(),
                                   Conv2d(planes, planes, 3, 1, 1, sigma_prior=sp, args=args), BatchNorm2d(planes)])
        self.shortcut = nn.ModuleList([])
        if stride != 1 or in_planes != planes:
            self.shortcut.append(Conv2d(in_planes, planes, 1, stride, 0, sigma_prior=sp, args=args))
            self.shortcut.append(BatchNorm2d(planes))

    def forward(self, x):
        out = self.stem[0](x, relu=True, bn=self.stem[1])                              # conv - bn - relu
        sc = self.shortcut[0](x, bn=self.shortcut[1]) if len(self.shortcut) else x       # conv - bn
        return self.stem[3](out, relu=True, bn=self.stem[4], res=sc)                     # conv - bn - Add - ReLU


class ConvNetwork_ResNet(nn.Module):
    """reference models_bbb.ConvNetwork_ResNet with q=False (models_bbb.py:191-245), eval mode."""

    def __init__(self, input_size, output_size, q, args):
        super().__init__()
        self.args, self.q = args, q
        sp = getattr(args, "sigma_prior", -2)
        self.in_planes = 24
        self.layers = nn.ModuleList([Conv2d(input_size[1], 24, 3, 1, 1, sigma_prior=sp, args=args), BatchNorm2d(24), nn.ReLU()])
        for planes, stride in ((24, 1), (48, 2), (96, 2), (192, 2)):
            blocks = []
            for st in (stride, 1):
                blocks.append(BasicBlock(self.in_planes, planes, st, q, args))
                self.in_planes = planes
            self.layers.append(nn.ModuleList(blocks))
        self.layers.append(nn.Identity())     # AvgPool2d(4)
        self.layers.append(nn.Identity())     # Flatten
        self.layers.append(Linear(192, output_size, bias=False, sigma_prior=sp, args=args))
        for i, (_, m) in enumerate(self.stochastic_named()):
            m.layer_id = i

    def stochastic_named(self):
        """(reference name, module) in execution = noise-draw order (stem.0, stem.3, then the shortcut conv)."""
        out = [("layers.0", self.layers[0])]
        for li in (3, 4, 5, 6):
            for bi, blk in enumerate(self.layers[li]):
                out.append((f"layers.{li}.{bi}.stem.0", blk.stem[0]))
                out.append((f"layers.{li}.{bi}.stem.3", blk.stem[3]))
                if len(blk.shortcut):
                    out.append((f"layers.{li}.{bi}.shortcut.0", blk.shortcut[0]))
        out.append(("layers.9", self.layers[9]))
        return out

    def load_reference_state(self, state):
        for n, m in self.stochastic_named():
            _load_bbb(m, state, n)
        _load_bn(self.layers[1], state, "layers.1")
        for li in (3, 4, 5, 6):
            for bi, blk in enumerate(self.layers[li]):
                _load_bn(blk.stem[1], state, f"layers.{li}.{bi}.stem.1")
                _load_bn(blk.stem[4], state, f"layers.{li}.{bi}.stem.4")
                if len(blk.shortcut):
                    _load_bn(blk.shortcut[1], state, f"layers.{li}.{bi}.shortcut.1")
        return self

    def get_kl_divergence(self):
        return sum(m.get_kl_divergence() for _, m in self.stochastic_named())

    def forward_mc(self, x):
        batch_weights_f32([m for _, m in self.stochastic_named()], x.device)      # all 21 layers' weight draws in one launch
        h = nchw_to_mc_nhwc(x)
        h = self.layers[0](h, relu=True, bn=self.layers[1])
        for li in (3, 4, 5, 6):
            for blk in self.layers[li]:
                h = blk(h)
        h = pool2d_f32(h, 4, avg=True)
        h = flatten_f32(h)
        return softmax_f32(self.layers[9](h, act=0))

    def forward(self, x):
        with mc_context(1, _MC.seed, _MC.sample_begin, _MC.eps):
            return self.forward_mc(x)[0]
