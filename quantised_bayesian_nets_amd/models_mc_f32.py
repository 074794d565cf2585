"""Float MC-Dropout graphs behind the reference's model API (SURVEY rows a6 / a7 with q=False).

Mirror of reference src/models/stochastic/mcdropout/dropout.py (`BernoulliDropout`, :6-46, un-prepared: `mul_mask` / `mul_scalar` are
FloatFunctional, i.e. plain fp32 `(x * mask) * multiplier`) and mcdropout/models_mc.py: `LinearNetwork` (:10-73, `linear_mc`),
`ConvNetwork_LeNet` (:75-115, `conv_lenet_mc`), `BasicBlock` / `ConvNetwork_ResNet` (:116-226, `conv_resnet_mc`) with q=False in eval mode.
Weights are deterministic (nn.Linear / nn.Conv2d / nn.BatchNorm2d state); the only noise is the always-on Bernoulli mask, drawn from
the build's Philox uniform stream (seed, dropout index in execution order, global sample index) -- the same stream the quantised dropout uses.
All S samples of the active mc_context are evaluated per call: activations fp32 NHWC [S, B, H, W, C] / [S, B, F].
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .layers import _MC, bump_state_epoch, mc_context, timed
from .models_f32 import BatchNorm2d, _f32, _load_bn, flatten_f32, nchw_to_mc_nhwc, pool2d_f32, softmax_f32


class BernoulliDropout(nn.Module):
    """reference mcdropout/dropout.py:6-46 on fp32 tensors.  Always stochastic (no `training` check); p <= 0 is the identity; a 4-D input
    drops whole channels (one draw per (sample, image, channel)), a 2-D one single elements; y = (x * mask) * (1 / (1 - p))."""

    def __init__(self, p=0.0):
        super().__init__()
        self.p = nn.Parameter(torch.ones((1,)) * p, requires_grad=False)
        self.multiplier = nn.Parameter(torch.ones((1,)) / (1.0 - self.p), requires_grad=False)
        self.layer_id = 0           # Philox tensor id: index of this dropout among the model's dropouts (execution order)

    def active(self):
        return float(self.p) > 0.0

    def mult(self):
        return float(np.float32(self.multiplier.item()))

    def mask(self, B, C, device, injected=None):
        """fp32 0 / 1 table [S, B, C] for the S samples of the active mc_context (Philox, or `injected` in parity mode)."""
        S = _MC.samples
        if injected is not None:
            m = injected.to(device=device, dtype=torch.float32).contiguous()
            assert m.numel() == S * B * C
            return m
        m = torch.empty((S, B * C), dtype=torch.float32, device=device)
        keep = float(np.float32(1.0) - np.float32(self.p.item()))
        with timed("dropout_mask_f32"):
            _lib.check(_lib.lib().qbnn_dropout_mask_f32_mc(B * C, keep, _MC.seed, self.layer_id, _MC.sample_begin, S, _lib.ptr(m),
                                                           _lib.current_stream()))
        return m

    def forward(self, x, injected=None, res=None, relu=False):
        """x [S|1, B, (H, W,) C] -> [S, ...]; optionally + res and ReLU behind the dropout (the Add + `end` of a BasicBlock)."""
        if x.device.type != "cuda":
            raise RuntimeError("qbnn layers run on an MI355X only (no CPU fallback)")
        if not self.active():
            assert res is None and not relu
            return x
        S = _MC.samples
        B, C = x.shape[1], x.shape[-1]
        HW = int(np.prod(x.shape[2:-1])) if x.dim() > 3 else 1
        m = self.mask(B, C, x.device, injected)
        x = x.contiguous()
        y = torch.empty((S,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
        n = y[0].numel()
        with timed("dropout_f32"):
            _lib.check(_lib.lib().qbnn_dropout_f32_mc(_lib.ptr(x), 0 if x.shape[0] == 1 else n, _lib.ptr(m), B, HW, C, self.mult(), _lib.ptr(res),
                                                      0 if res is None or res.shape[0] == 1 else n, int(relu), _lib.ptr(y), n, S,
                                                      _lib.current_stream()))
        return y

    def extra_repr(self):
        return 'p={}, quant={}'.format(self.p.item(), False)


class Linear(nn.Module):
    """nn.Linear evaluated for every MC sample: y_s = x_s @ W^T + b (one weight for all samples)."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features).uniform_(-0.01, 0.01), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(out_features), requires_grad=False) if bias else None
        self._dev = None

    def forward(self, x, act=0):
        """x fp32 [S|1, B, in_features] -> [S|1, B, out_features]; act: 0 none, 1 ReLU, 2 exp."""
        if x.device.type != "cuda":
            raise RuntimeError("qbnn layers run on an MI355X only (no CPU fallback)")
        if self._dev is None or self._dev[0].device != x.device:
            self._dev = (_f32(self.weight, x.device).reshape(-1), _f32(self.bias, x.device))
        w, b = self._dev
        S, B = x.shape[0], x.shape[1]
        x = x.contiguous()
        y = torch.empty((S, B, self.out_features), dtype=torch.float32, device=x.device)
        with timed("linear_f32"):
            _lib.check(_lib.lib().qbnn_linear_f32_mc(_lib.ptr(x), x[0].numel(), _lib.ptr(w), 0, _lib.ptr(b), _lib.ptr(y), y[0].numel(), B,
                                                     self.in_features, self.out_features, act, S, _lib.current_stream()))
        return y


class Conv2d(nn.Module):
    """nn.Conv2d (bias optional) evaluated for every MC sample, with the graph's tail -- BatchNorm (eval), BernoulliDropout, Add, ReLU --
    in the conv kernel's epilogue (qbnn_conv2d_f32_drop_mc), each step rounded to fp32 like the reference's separate modules."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False):
        super().__init__()
        self.in_channels, self.out_channels, self.k, self.stride, self.padding = in_channels, out_channels, int(kernel_size), int(stride), int(padding)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, self.k, self.k).uniform_(-0.01, 0.01), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(out_channels), requires_grad=False) if bias else None
        self._dev = None

    def forward(self, x, bn=None, drop=None, injected=None, res=None, relu=False):
        if x.device.type != "cuda":
            raise RuntimeError("qbnn layers run on an MI355X only (no CPU fallback)")
        if self._dev is None or self._dev[0].device != x.device:
            w = self.weight.detach().float().permute(0, 2, 3, 1).contiguous()          # [Cout][k][k][Cin]: K contiguous for the implicit GEMM
            self._dev = (w.to(x.device).reshape(-1), _f32(self.bias, x.device))
        w, b = self._dev
        alpha, beta = bn.coefficients(x.device) if bn is not None else (None, None)
        Sx, B, H, W = x.shape[0], x.shape[1], x.shape[2], x.shape[3]
        dropping = drop is not None and drop.active()
        S = _MC.samples if (dropping or (res is not None and res.shape[0] > 1)) else Sx
        mask = drop.mask(B, self.out_channels, x.device, injected) if dropping else None
        Ho, Wo = (H + 2 * self.padding - self.k) // self.stride + 1, (W + 2 * self.padding - self.k) // self.stride + 1
        x = x.contiguous()
        y = torch.empty((S, B, Ho, Wo, self.out_channels), dtype=torch.float32, device=x.device)
        with timed("conv2d_f32_drop"):
            _lib.check(_lib.lib().qbnn_conv2d_f32_drop_mc(_lib.ptr(x), 0 if Sx == 1 else x[0].numel(), _lib.ptr(w), 0, _lib.ptr(b), _lib.ptr(alpha),
                                                          _lib.ptr(beta), _lib.ptr(mask), drop.mult() if dropping else 1.0, _lib.ptr(res),
                                                          0 if res is None or res.shape[0] == 1 else res[0].numel(), _lib.ptr(y), y[0].numel(), B, H, W,
                                                          self.in_channels, self.out_channels, self.k, self.stride, self.padding,
                                                          int(relu) | 4, S, _lib.current_stream()))
        return y


def _load_wb(m, state, name):
    m.weight.data = torch.from_numpy(np.asarray(state[name + ".weight"], np.float32).copy()).reshape(m.weight.shape)
    if m.bias is not None:
        m.bias.data = torch.from_numpy(np.asarray(state[name + ".bias"], np.float32).copy())
    m._dev = None
    bump_state_epoch()


def _load_drop(d, state, name):
    d.p.data = torch.from_numpy(np.asarray(state[name + ".p"], np.float32).reshape(1).copy())
    d.multiplier.data = torch.from_numpy(np.asarray(state[name + ".multiplier"], np.float32).reshape(1).copy())


def _number_dropouts(drops):
    for i, d in enumerate(drops):
        d.layer_id = i


def _take(masks):
    return masks.pop(0) if masks is not None else None


class LinearNetwork(nn.Module):
    """reference mcdropout/models_mc.LinearNetwork (:10-73) with q=False: 3 x (Linear(100) + ReLU) with a dropout after the first two,
    heads `mu` / `log_var` = [BernoulliDropout, Linear(100, 1)]; forward -> (mu, exp(log_var))."""

    def __init__(self, input_size, output_size, q, args):
        super().__init__()
        self.args, self.q = args, q
        self.input_size = 1
        for i in input_size:
            self.input_size *= int(i)
        self.output_size = int(output_size)
        widths = [100, 100, 100]
        self.layers = nn.ModuleList([])
        prev = self.input_size
        for i, wd in enumerate(widths):
            self.layers.append(Linear(prev, wd, bias=True))
            self.layers.append(nn.ReLU())
            if i != len(widths) - 1:
                self.layers.append(BernoulliDropout(args.p))
            prev = wd
        self.mu = nn.ModuleList([BernoulliDropout(args.p), Linear(prev, 1, bias=True)])
        self.log_var = nn.ModuleList([BernoulliDropout(args.p), Linear(prev, 1, bias=True)])
        _number_dropouts(self.dropouts())

    def dropouts(self):
        """Execution (= mask draw) order: layers.2, layers.5, mu.0, log_var.0."""
        return [self.layers[2], self.layers[5], self.mu[0], self.log_var[0]]

    def load_reference_state(self, state):
        for n, m in (("layers.0", self.layers[0]), ("layers.3", self.layers[3]), ("layers.6", self.layers[6]), ("mu.1", self.mu[1]),
                     ("log_var.1", self.log_var[1])):
            _load_wb(m, state, n)
        for n, d in zip(("layers.2", "layers.5", "mu.0", "log_var.0"), self.dropouts()):
            _load_drop(d, state, n)
        return self

    def forward_mc(self, x, masks=None):
        """All S samples -> (mu [S,B,1], var [S,B,1]).  masks: optional list of fp32 [S, B, 100] in draw order (parity mode)."""
        if x.device.type != "cuda":
            raise RuntimeError("qbnn models run on an MI355X only (no CPU fallback)")
        masks = list(masks) if masks is not None else None
        h = x.to(torch.float32).reshape(1, x.shape[0], -1).contiguous()
        h = self.layers[2](self.layers[0](h, act=1), _take(masks))
        h = self.layers[5](self.layers[3](h, act=1), _take(masks))
        h = self.layers[6](h, act=1)
        mu = self.mu[1](self.mu[0](h, _take(masks)), act=0)
        var = self.log_var[1](self.log_var[0](h, _take(masks)), act=2)
        return mu, var

    def forward(self, x):
        with mc_context(1, _MC.seed, _MC.sample_begin, _MC.eps):
            mu, var = self.forward_mc(x)
        return mu[0], var[0]


class ConvNetwork_LeNet(nn.Module):
    """reference mcdropout/models_mc.ConvNetwork_LeNet (:75-115) with q=False: conv5x5(p2) - dropout - maxpool2 - conv5x5(p2) - dropout -
    maxpool2 - flatten - fc500 - relu - dropout - fc - softmax (no bias anywhere, no ReLU after the convs)."""

    def __init__(self, input_size, output_size, q, args):
        super().__init__()
        self.args, self.q, self.output_size = args, q, int(output_size)
        ident = nn.Identity
        self.layers = nn.ModuleList([Conv2d(input_size[0], 20, 5, 1, 2), BernoulliDropout(args.p), ident(),
                                     Conv2d(20, 50, 5, 1, 2), BernoulliDropout(args.p), ident(), ident(),
                                     Linear(50 * 7 * 7, 500, bias=False), nn.ReLU(), BernoulliDropout(args.p),
                                     Linear(500, output_size, bias=False)])
        _number_dropouts(self.dropouts())

    def dropouts(self):
        return [self.layers[1], self.layers[4], self.layers[9]]

    def load_reference_state(self, state):
        for i in (0, 3, 7, 10):
            _load_wb(self.layers[i], state, f"layers.{i}")
        for i in (1, 4, 9):
            _load_drop(self.layers[i], state, f"layers.{i}")
        return self

    def forward_mc(self, x, masks=None):
        """-> softmax probabilities [S, B, classes].  masks: optional list [S,B,20], [S,B,50], [S,B,500]."""
        if x.device.type != "cuda":
            raise RuntimeError("qbnn models run on an MI355X only (no CPU fallback)")
        masks = list(masks) if masks is not None else None
        h = nchw_to_mc_nhwc(x)
        # dropout -> max-pool == max-pool -> dropout bit for bit (one mask value per pooling window, factor >= 0), but the conv's fused
        # tail keeps the reference's order: conv -> dropout, then the pool
        h = pool2d_f32(self.layers[0](h, drop=self.layers[1], injected=_take(masks)), 2, avg=False)
        h = pool2d_f32(self.layers[3](h, drop=self.layers[4], injected=_take(masks)), 2, avg=False)
        h = flatten_f32(h)
        h = self.layers[9](self.layers[7](h, act=1), _take(masks))
        return softmax_f32(self.layers[10](h, act=0))

    def forward(self, x):
        with mc_context(1, _MC.seed, _MC.sample_begin, _MC.eps):
            return self.forward_mc(x)[0]


class BasicBlock(nn.Module):
    """reference mcdropout/models_mc.BasicBlock (:116-160), float: stem = conv, bn, relu, dropout, conv, bn, dropout; shortcut = conv1x1/s,
    bn, dropout where the shape changes; Add; ReLU.  Three launches: every conv carries its BatchNorm / dropout / Add / ReLU tail."""
    expansion = 1

    def __init__(self, in_planes, planes, stride=1, q=False, args=None):
        super().__init__()
        self.args = args
        self.stem = nn.ModuleList([Conv2d(in_planes, planes, 3, stride, 1), BatchNorm2d(planes), nn.ReLU(), BernoulliDropout(args.p),
                                   Conv2d(planes, planes, 3, 1, 1), BatchNorm2d(planes), BernoulliDropout(args.p)])
        self.shortcut = nn.ModuleList([])
        if stride != 1 or in_planes != planes:
            self.shortcut.append(Conv2d(in_planes, planes, 1, stride, 0))
            self.shortcut.append(BatchNorm2d(planes))
            self.shortcut.append(BernoulliDropout(args.p))

    def dropouts(self):
        return [self.stem[3], self.stem[6]] + ([self.shortcut[2]] if len(self.shortcut) else [])

    def load_reference_state(self, state, prefix):
        _load_wb(self.stem[0], state, prefix + "stem.0"); _load_bn(self.stem[1], state, prefix + "stem.1"); _load_drop(self.stem[3], state, prefix + "stem.3")
        _load_wb(self.stem[4], state, prefix + "stem.4"); _load_bn(self.stem[5], state, prefix + "stem.5"); _load_drop(self.stem[6], state, prefix + "stem.6")
        if len(self.shortcut):
            _load_wb(self.shortcut[0], state, prefix + "shortcut.0"); _load_bn(self.shortcut[1], state, prefix + "shortcut.1")
            _load_drop(self.shortcut[2], state, prefix + "shortcut.2")

    def forward(self, x, masks):
        m_a, m_b = _take(masks), _take(masks)                 # the reference's draw order: stem.3, stem.6, shortcut.2
        sc = x
        if len(self.shortcut):
            sc = self.shortcut[0](x, bn=self.shortcut[1], drop=self.shortcut[2], injected=_take(masks))
        out = self.stem[0](x, bn=self.stem[1], drop=self.stem[3], injected=m_a, relu=True)      # relu(.) * m == relu(. * m): m >= 0
        return self.stem[4](out, bn=self.stem[5], drop=self.stem[6], injected=m_b, res=sc, relu=True)


class ConvNetwork_ResNet(nn.Module):
    """reference mcdropout/models_mc.ConvNetwork_ResNet (:162-226) with q=False, eval mode."""

    def __init__(self, input_size, output_size, q, args):
        super().__init__()
        self.args, self.q, self.output_size = args, q, int(output_size)
        ident = nn.Identity
        self.layers = nn.ModuleList([Conv2d(input_size[1], 24, 3, 1, 1), BatchNorm2d(24), nn.ReLU(), BernoulliDropout(args.p)])
        inp = 24
        for planes, stride in ((24, 1), (48, 2), (96, 2), (192, 2)):
            blocks = []
            for st in (stride, 1):
                blocks.append(BasicBlock(inp, planes, st, q, args))
                inp = planes
            self.layers.append(nn.ModuleList(blocks))
        self.layers.append(ident())      # AvgPool2d(4)
        self.layers.append(ident())      # Flatten
        self.layers.append(Linear(192, output_size, bias=False))
        _number_dropouts(self.dropouts())

    def dropouts(self):
        out = [self.layers[3]]
        for li in (4, 5, 6, 7):
            for blk in self.layers[li]:
                out += blk.dropouts()
        return out

    def load_reference_state(self, state):
        _load_wb(self.layers[0], state, "layers.0"); _load_bn(self.layers[1], state, "layers.1"); _load_drop(self.layers[3], state, "layers.3")
        for li in (4, 5, 6, 7):
            for bi, blk in enumerate(self.layers[li]):
                blk.load_reference_state(state, f"layers.{li}.{bi}.")
        _load_wb(self.layers[10], state, "layers.10")
        return self

    def forward_mc(self, x, masks=None):
        """-> softmax probabilities [S, B, classes].  masks: optional list of fp32 [S, B, C] in draw order."""
        if x.device.type != "cuda":
            raise RuntimeError("qbnn models run on an MI355X only (no CPU fallback)")
        masks = list(masks) if masks is not None else None
        h = nchw_to_mc_nhwc(x)
        h = self.layers[0](h, bn=self.layers[1], drop=self.layers[3], injected=_take(masks), relu=True)
        for li in (4, 5, 6, 7):
            for blk in self.layers[li]:
                h = blk(h, masks)
        h = flatten_f32(pool2d_f32(h, 4, avg=True))
        return softmax_f32(self.layers[10](h, act=0))

    def forward(self, x):
        with mc_context(1, _MC.seed, _MC.sample_begin, _MC.eps):
            return self.forward_mc(x)[0]
