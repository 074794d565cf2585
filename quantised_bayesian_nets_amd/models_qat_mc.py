"""QAT fake-quant evaluation with LIVE observers of the NON-BBB graphs (SURVEY 8(f).3 widened; round 5).

quant_utils.prepare_model has two branches: `prepare_qat` for every model without `bbb` in its name (reference src/quant_utils.py:139-140) and
the BBB mapping (:141-147, models_qat.py).  This module restates the first for the graphs the reference ships: the MC-Dropout nets
(mcdropout/models_mc.py: `linear_mc` :10-73, `conv_lenet_mc` :75-115, `conv_resnet_mc` :116-226) and the SGHMC member templates
(sgld/models_sgld.py: `linear_sgld`, `conv_lenet_sgld`, `conv_resnet_sgld` with training_mode=True, state names under `main_net.`).
After `prepare_qat` their layers are torch.ao.nn.qat Linear / Conv2d and intrinsic.qat LinearReLU / ConvBn2d / ConvBnReLU2d:
    W = weight_fake_quant(weight * c)        c = gamma / sqrt(running_var + eps) for a fused conv-bn, else 1  (deterministic weights)
    Z = conv(X, W);  Z / c (+ bias);  bn;  (ReLU);  activation FakeQuantize
i.e. models_qat.Conv2d / Linear without the noise branch, and `BernoulliDropout`'s two FloatFunctionals carry FakeQuantize observers
(mcdropout/dropout.py:9-13): y = FQ_mul_mask(x * mask) * multiplier -- `mul_scalar` is not observed (torch FloatFunctional.mul_scalar).
Masks come from the build's Philox uniform stream (seed, dropout index in execution order, global sample index), as in models_mc_f32.py.
Every observer keeps updating in eval; the S samples are evaluated together with the observer recurrence resolved on the device
(models_qat.FakeQuantize)."""
import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .layers import _MC, bump_state_epoch, mc_context, timed
from .models_f32 import affine_f32, flatten_f32, nchw_to_mc_nhwc, pool2d_f32, softmax_f32
from .models_qat import Conv2d as _ConvBBB
from .models_qat import FakeQuantize, QuantStub, _bounds, block_add, keep_grid, need_f32, prepared_state, presample_weights
from .models_qat import Linear as _LinearBBB


class _Deterministic:
    """The weight pipeline of a torch.ao.nn.qat layer: one FakeQuantize on (weight * c); it is observed once per forward, so sample s is
    quantised with the qparams the observer holds after s + 1 looks at the same tensor."""

    def _init_fq(self, args):
        (alo, ahi), (wlo, whi) = _bounds(args)
        self.weight_fake_quant = FakeQuantize(wlo, whi)
        self.activation_post_process = FakeQuantize(alo, ahi)
        self.layer_id = 0
        self._folded = None

    def weight_grid(self):
        return self.weight_fake_quant

    def _folded_params(self, dev):
        if self._folded is None or self._folded.device != dev:
            mu = self.weight.detach().float().cpu()
            c = self.scale_factor()
            if c is not None:
                mu = mu * c.reshape([-1] + [1] * (mu.dim() - 1))
            if mu.dim() == 4:
                mu = mu.permute(0, 2, 3, 1)         # [Cout][k][k][Cin]: the layout the conv kernel streams
            self._folded = mu.reshape(1, -1).contiguous().to(dev)
        return self._folded

    def sampled_weights(self, dev, eps=None):
        pre = getattr(self, "_presampled", None)
        self._presampled = None
        from .models_qat import _presample_key
        if pre is not None and pre[2] == _presample_key(dev) and pre[3] == torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream().wait_event(pre[1])
            pre[0].record_stream(torch.cuda.current_stream())
            if getattr(pre[0], "_q8", None) is not None:
                pre[0]._q8.record_stream(torch.cuda.current_stream())
            return pre[0]
        W = self.weight_fake_quant(self._folded_params(dev))
        from .models_qat import qat_i8_enabled, weights_to_i8
        if qat_i8_enabled() and self.weight_fake_quant.qmin >= -128 and self.weight_fake_quant.qmax <= 127:
            W._q8 = weights_to_i8(W, self.weight_fake_quant.last_scale, self.weight_fake_quant.last_zero_point)
        return W

    def _load_common(self, st, name):
        self.weight.data = torch.from_numpy(np.asarray(st[name + ".weight"], np.float32).copy()).reshape(self.weight.shape)
        if self.bias is not None:
            self.bias.data = torch.from_numpy(np.asarray(st[name + ".bias"], np.float32).copy())
        self.weight_fake_quant.load(st, name + ".weight_fake_quant")
        self.activation_post_process.load(st, name + ".activation_post_process")
        self._folded = None


class Conv2d(_Deterministic, _ConvBBB):
    """torch.ao.nn.qat.Conv2d / intrinsic.qat ConvBn2d / ConvBnReLU2d (`bn`, `relu` flags) in eval."""


class Linear(_Deterministic, _LinearBBB):
    """torch.ao.nn.qat.Linear / intrinsic.qat.LinearReLU in eval."""


class BernoulliDropout(nn.Module):
    """mcdropout/dropout.py:6-46 after prepare_qat: FakeQuantize on the masked tensor, the 1 / (1 - p) gain behind it."""

    def __init__(self, p, args):
        super().__init__()
        self.p = nn.Parameter(torch.ones((1,)) * p, requires_grad=False)
        self.multiplier = nn.Parameter(torch.ones((1,)) / (1.0 - self.p), requires_grad=False)
        (alo, ahi), _ = _bounds(args)
        self.mul_mask = FakeQuantize(alo, ahi)           # mul_mask.activation_post_process
        self.layer_id = 0                                # index among the model's dropouts in execution order (Philox tensor id)
        self._gain = None

    def load(self, st, name):
        self.p.data = torch.from_numpy(np.asarray(st[name + ".p"], np.float32).reshape(1).copy())
        self.multiplier.data = torch.from_numpy(np.asarray(st[name + ".multiplier"], np.float32).reshape(1).copy())
        self.mul_mask.load(st, name + ".mul_mask.activation_post_process")
        self._gain = None
        return self

    def forward(self, x):
        if float(self.p) <= 0.0:
            return x
        S = _MC.samples
        B, C = x.shape[1], x.shape[-1]
        HW = int(np.prod(x.shape[2:-1])) if x.dim() > 3 else 1
        dev = x.device
        m = torch.empty((S, B * C), dtype=torch.float32, device=dev)
        keep = float(np.float32(1.0) - np.float32(self.p.item()))
        L = _lib.lib()
        with timed("dropout_mask_f32"):
            _lib.check(L.qbnn_dropout_mask_f32_mc(B * C, keep, _MC.seed, self.layer_id, _MC.sample_begin, S, _lib.ptr(m), _lib.current_stream()))
        x = x.contiguous()
        y = torch.empty((S,) + tuple(x.shape[1:]), dtype=torch.float32, device=dev)
        n = y[0].numel()
        with timed("dropout_f32"):          # x * mask (gain 1: exact)
            _lib.check(L.qbnn_dropout_f32_mc(_lib.ptr(x), 0 if x.shape[0] == 1 else n, _lib.ptr(m), B, HW, C, 1.0, None, 0, 0, _lib.ptr(y), n, S,
                                             _lib.current_stream()))
        y = self.mul_mask(y)
        if self._gain is None or self._gain.device != dev or self._gain.numel() != C:
            self._gain = torch.full((C,), float(np.float32(self.multiplier.item())), dtype=torch.float32, device=dev)
        return keep_grid(affine_f32(y, p0=self._gain), y, gain=np.float32(self.multiplier.item()))      # grid integers x (scale / (1 - p))


class _Net(nn.Module):
    sequential_samples = True      # live EMA observers: sample s depends on samples < s (mc.py refuses to shard these over ranks)
    prefix = ""

    def _finish(self):
        for i, (_, m) in enumerate(self.weighted()):
            m.layer_id = i
        for i, (_, d) in enumerate(self.dropouts()):
            d.layer_id = i

    def dropouts(self):
        return []

    def extra_observers(self):
        return []

    def prepared_state(self):
        return prepared_state(self)

    def load_reference_state(self, st):
        bump_state_epoch()
        self._prepared = st
        pre = self.prefix
        self.quant.activation_post_process.load(st, pre + "quant.activation_post_process")
        for n, m in self.weighted():
            m.load(st, pre + n)
        for n, d in self.dropouts():
            d.load(st, pre + n)
        for n, f in self.extra_observers():
            f.load(st, pre + n)
        return self

    def forward(self, x):
        with mc_context(1, _MC.seed, _MC.sample_begin, _MC.eps):
            y = self.forward_mc(x)
        return (y[0][0], y[1][0]) if isinstance(y, tuple) else y[0]


class LinearNetwork(_Net):
    """Prepared `linear_mc` (mcdropout/models_mc.py:10-73; p > 0) or the `linear_sgld` template (p = 0: no dropout modules, the BBB MLP's names)."""

    def __init__(self, input_size, output_size, q, args, p=None, prefix=""):
        super().__init__()
        self.args, self.q, self.prefix = args, q, prefix
        self.input_size = 1
        for i in input_size:
            self.input_size *= int(i)
        self.p = float(getattr(args, "p", 0.0)) if p is None else p
        self.quant = QuantStub(args)
        self.fc = nn.ModuleList([Linear(self.input_size if i == 0 else 100, 100, bias=True, args=args, relu=True) for i in range(3)])
        self.mu_fc, self.lv_fc = Linear(100, 1, bias=True, args=args), Linear(100, 1, bias=True, args=args)
        self.drop = nn.ModuleList([BernoulliDropout(self.p, args) for _ in range(4)]) if self.p > 0 else nn.ModuleList([])
        self._finish()

    def weighted(self):
        if self.p > 0:
            return list(zip(["layers.0", "layers.3", "layers.6", "mu.1", "log_var.1"], list(self.fc) + [self.mu_fc, self.lv_fc]))
        return list(zip(["layers.0", "layers.2", "layers.4", "mu", "log_var"], list(self.fc) + [self.mu_fc, self.lv_fc]))

    def dropouts(self):
        return list(zip(["layers.2", "layers.5", "mu.0", "log_var.0"], self.drop)) if self.p > 0 else []

    def forward_mc(self, x):
        h = self.quant(x.to(torch.float32).reshape(1, x.shape[0], -1))
        d = list(self.drop) if self.p > 0 else [lambda t: t] * 4
        h = d[0](self.fc[0](h))
        h = d[1](self.fc[1](h))
        h = self.fc[2](h)
        mu = self.mu_fc(d[2](h))
        lv = self.lv_fc(d[3](h))
        return mu, torch.exp(lv)


class ConvNetwork_LeNet(_Net):
    """Prepared `conv_lenet_mc` (models_mc.py:75-115) or the `conv_lenet_sgld` template."""

    def __init__(self, input_size, output_size, q, args, p=None, prefix=""):
        super().__init__()
        self.args, self.q, self.prefix = args, q, prefix
        c0 = input_size[0] if len(input_size) == 3 else input_size[1]
        self.p = float(getattr(args, "p", 0.0)) if p is None else p
        self.quant = QuantStub(args)
        self.c1, self.c2 = Conv2d(c0, 20, 5, 1, 2, args=args), Conv2d(20, 50, 5, 1, 2, args=args)
        self.f1, self.f2 = Linear(50 * 7 * 7, 500, args=args, relu=True), Linear(500, output_size, args=args)
        self.drop = nn.ModuleList([BernoulliDropout(self.p, args) for _ in range(3)]) if self.p > 0 else nn.ModuleList([])
        self._finish()

    def weighted(self):
        names = ["layers.0", "layers.3", "layers.7", "layers.10"] if self.p > 0 else ["layers.0", "layers.2", "layers.5", "layers.7"]
        return list(zip(names, [self.c1, self.c2, self.f1, self.f2]))

    def dropouts(self):
        return list(zip(["layers.1", "layers.4", "layers.9"], self.drop)) if self.p > 0 else []

    def forward_mc(self, x):
        d = list(self.drop) if self.p > 0 else [lambda t: t] * 3
        h = self.quant(nchw_to_mc_nhwc(x))
        c = d[0](self.c1(h))
        h = keep_grid(pool2d_f32(c, 2, avg=False), c)
        c = d[1](self.c2(h))
        h = keep_grid(pool2d_f32(c, 2, avg=False), c)
        h = d[2](self.f1(keep_grid(flatten_f32(h), h)))
        return softmax_f32(self.f2(h))


class _Block(nn.Module):
    def __init__(self, in_planes, planes, stride, args, p):
        super().__init__()
        (alo, ahi), _ = _bounds(args)
        self.a = Conv2d(in_planes, planes, 3, stride, 1, args=args, bn=True, relu=True)
        self.b = Conv2d(planes, planes, 3, 1, 1, args=args, bn=True)
        self.s = Conv2d(in_planes, planes, 1, stride, 0, args=args, bn=True) if (stride != 1 or in_planes != planes) else None
        self.da, self.db = (BernoulliDropout(p, args), BernoulliDropout(p, args)) if p > 0 else (None, None)
        self.ds = BernoulliDropout(p, args) if (p > 0 and self.s is not None) else None
        self.add = FakeQuantize(alo, ahi)          # add.add.activation_post_process

    def forward(self, x, f32_out=True):
        # Without dropouts (the SGHMC member template) every tensor inside the block is consumed by a conv on the int8 pipe or by the Add: it travels as
        # grid integers + scale, as in models_qat.BasicBlock (round 6).  A BernoulliDropout works on the fp32 values: those graphs keep them.
        grid = self.da is None and self.db is None and self.ds is None and getattr(x, "_q8", None) is not None
        out = self.a(x, f32_out=not grid)
        if self.da is not None:
            out = self.da(out)
        out = self.b(out, f32_out=not grid)
        if self.db is not None:
            out = self.db(out)
        sc = x
        if self.s is not None:
            sc = self.s(x, f32_out=not grid)
            if self.ds is not None:
                sc = self.ds(sc)
        return block_add(self.add, out, sc, f32_out, from_integers=grid)


class ConvNetwork_ResNet(_Net):
    """Prepared `conv_resnet_mc` (models_mc.py:116-226: a channel dropout behind every conv) or the `conv_resnet_sgld` template (no dropout)."""

    def __init__(self, input_size, output_size, q, args, p=None, prefix=""):
        super().__init__()
        self.args, self.q, self.prefix = args, q, prefix
        self.p = float(getattr(args, "p", 0.0)) if p is None else p
        self.quant = QuantStub(args)
        self.c0 = Conv2d(input_size[1], 24, 3, 1, 1, args=args, bn=True, relu=True)
        self.d0 = BernoulliDropout(self.p, args) if self.p > 0 else None
        blocks, inp = [], 24
        for planes, stride in ((24, 1), (48, 2), (96, 2), (192, 2)):
            for st in (stride, 1):
                blocks.append(_Block(inp, planes, st, args, self.p))
                inp = planes
        self.blocks = nn.ModuleList(blocks)
        self.fc = Linear(192, output_size, args=args)
        self._finish()

    def _block_names(self):
        first = 4 if self.p > 0 else 3          # the MC net has a dropout module at layers.3
        return [f"layers.{first + i // 2}.{i % 2}" for i in range(8)]

    def weighted(self):
        b_name, s_name = ("stem.4", "shortcut.0") if self.p > 0 else ("stem.3", "shortcut.0")
        out = [("layers.0", self.c0)]
        for n, blk in zip(self._block_names(), self.blocks):
            out += [(n + ".stem.0", blk.a), (n + "." + b_name, blk.b)]
            if blk.s is not None:
                out.append((n + "." + s_name, blk.s))
        out.append(("layers.10" if self.p > 0 else "layers.9", self.fc))
        return out

    def dropouts(self):
        if self.p <= 0:
            return []
        out = [("layers.3", self.d0)]
        for n, blk in zip(self._block_names(), self.blocks):
            out += [(n + ".stem.3", blk.da), (n + ".stem.6", blk.db)]
            if blk.ds is not None:
                out.append((n + ".shortcut.2", blk.ds))
        return out

    def extra_observers(self):
        return [(n + ".add.add.activation_post_process", blk.add) for n, blk in zip(self._block_names(), self.blocks)]

    def forward_mc(self, x):
        presample_weights([m for _, m in self.weighted()], x.device)
        nodrop = self.d0 is None                 # (the template: grid integers only between the input FakeQuantize and the last block's Add)
        h = self.c0(self.quant(nchw_to_mc_nhwc(x), f32_out=not nodrop), f32_out=not nodrop)
        if self.d0 is not None:
            h = self.d0(h)
        for i, blk in enumerate(self.blocks):
            h = blk(h, f32_out=(not nodrop) or i == len(self.blocks) - 1)
        return softmax_f32(self.fc(flatten_f32(pool2d_f32(need_f32(h), 4, avg=True))))


def get_model(model, input_size, output_size, q, args):
    """ModelFactory's hook: `args.qat_eval` on a non-BBB name."""
    if model.endswith("_sgld"):
        cls = {"linear_sgld": LinearNetwork, "conv_lenet_sgld": ConvNetwork_LeNet, "conv_resnet_sgld": ConvNetwork_ResNet}[model]
        return cls(input_size, output_size, q, args, p=0.0, prefix="main_net.")
    cls = {"linear_mc": LinearNetwork, "conv_lenet_mc": ConvNetwork_LeNet, "conv_resnet_mc": ConvNetwork_ResNet}[model]
    return cls(input_size, output_size, q, args)
