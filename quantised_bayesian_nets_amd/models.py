"""int8 conv_resnet_bbb behind the reference's model API.

Mirror of reference src/models/stochastic/bbb/models_bbb.py: BasicBlock (:146-188) and ConvNetwork_ResNet
(:191-256) *after* quant_utils.prepare_model -> convert, i.e. the graph the reference's MC evaluation executes
(SURVEY.md section 3.1).  Module names / state-dict keys are the reference's (`layers.4.0.stem.3.weight`, ...), so
a converted reference checkpoint loads by key.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .layers import LAYOUT_MFMA32, LAYOUT_MFMA32_N24, LAYOUT_MFMA32_TAIL, LAYOUT_MFMA32_N24_TAIL, w16_enabled, Conv2d, ConvReLU2d, Linear, MCQTensor, QFunctional, _MC, mc_context, timed, sample_all_weights
from .quant import UINT_BOUNDS, check_bits


GRAPH_FALLBACKS = []          # HIP-graph captures that were refused (the eager launch chain ran instead)


class Add(nn.Module):
    """reference src/utils.py:49-55 (`Add`, a FloatFunctional -> QFunctional add).  Fused into stem.3's epilogue."""

    def __init__(self):
        super().__init__()
        self.add = QFunctional()


class BasicBlock(nn.Module):
    """reference models_bbb.py:146-188 after fuse_model(): stem.0 = ConvReLU2d, stem.3 = Conv2d,
    shortcut.0 = Conv2d (1x1, stride 2) where the shape changes; add; end ReLU."""
    expansion = 1

    def __init__(self, in_planes, planes, stride=1, q=True, args=None):
        super().__init__()
        self.args, self.q = args, q
        ident = nn.Identity
        self.stem = nn.ModuleList([ConvReLU2d(in_planes, planes, 3, stride=stride, padding=1, bias=True, args=args), ident(), ident(),
                                   Conv2d(planes, planes, 3, stride=1, padding=1, bias=True, args=args), ident()])
        self.shortcut = nn.ModuleList([])
        if stride != 1 or in_planes != self.expansion * planes:
            self.shortcut.append(Conv2d(in_planes, self.expansion * planes, 1, stride=stride, bias=True, args=args))
            self.shortcut.append(ident())
        self.add = Add()
        self.end = nn.ReLU()

    def forward(self, x):
        S = _MC.samples
        dev = x.data.device
        w0 = self.stem[0].sample_weights(dev)
        w3 = self.stem[3].sample_weights(dev)
        out = self.stem[0]._conv(x, w0, S)
        if len(self.shortcut):
            ws = self.shortcut[0].sample_weights(dev)
            sc = self.shortcut[0]._conv(x, ws, S)
        else:
            sc = x
        return self.stem[3]._conv(out, w3, S, residual=sc, add_qparams=(self.add.add.scale, self.add.add.zero_point))


def _fill_block_desc(d, blk, dev, keep):
    ca, cb = blk.stem[0], blk.stem[3]
    wa, wb = ca.sample_weights(dev), cb.sample_weights(dev)
    pa, pb = ca._ensure_packed(dev), cb._ensure_packed(dev)
    keep += [wa, wb]
    d.w_a, d.w_a_sample_stride, d.bias_a = wa.data_ptr(), wa.shape[1], (pa["bias"].data_ptr() if pa["bias"] is not None else None)
    d.s_wa, d.z_wa, d.s_a, d.z_a = ca.add_weight.scale, ca.add_weight.zero_point, ca.scale, ca.zero_point
    d.w_b, d.w_b_sample_stride, d.bias_b = wb.data_ptr(), wb.shape[1], (pb["bias"].data_ptr() if pb["bias"] is not None else None)
    d.s_wb, d.z_wb, d.s_b, d.z_b = cb.add_weight.scale, cb.add_weight.zero_point, cb.scale, cb.zero_point
    d.s_o, d.z_o = blk.add.add.scale, blk.add.add.zero_point
    # (the 24 -> 48 down block's layout set: stem.0 as MFMA32_N24_TAIL beside MFMA32_N24 for stem.3 and the shortcut)
    assert ca.layout == cb.layout or (ca.layout, cb.layout) == (LAYOUT_MFMA32_N24_TAIL, LAYOUT_MFMA32_N24), "a block's convs share one packed layout set"
    d.w_layout = cb.layout


def run_down_block(blk, x):
    """A down-sampling BasicBlock (shortcut conv + stem + add) in ONE persistent fused kernel (qbnn_block_down_i8_mc)."""
    S = _MC.samples
    dev = x.data.device
    keep = []
    d = _lib.DownDesc()
    _fill_block_desc(d.blk, blk, dev, keep)
    cs = blk.shortcut[0]
    ws = cs.sample_weights(dev)
    ps = cs._ensure_packed(dev)
    keep.append(ws)
    d.w_s, d.w_s_sample_stride, d.bias_s = ws.data_ptr(), ws.shape[1], (ps["bias"].data_ptr() if ps["bias"] is not None else None)
    d.s_ws, d.z_ws, d.s_s, d.z_s = cs.add_weight.scale, cs.add_weight.zero_point, cs.scale, cs.zero_point
    _, B, H, W, Cin = x.data.shape
    Cout = blk.stem[0].out_channels
    y = torch.empty((S, B, H // 2, W // 2, Cout), dtype=torch.uint8, device=dev)
    a_hi = UINT_BOUNDS[blk.args.activation_precision][1]
    nw = lambda l: l._packed["cout"] * l._packed["k"]
    meta = dict(fused=True, convs=[(H, Cin, Cout, 3, 2, nw(blk.stem[0])), (H, Cin, Cout, 1, 2, nw(cs)), (H // 2, Cout, Cout, 3, 1, nw(blk.stem[3]))],
                res_convs=[2])          # stem.3 carries the Add + ReLU epilogue
    with timed("block_down_i8 %dx%d %d->%d" % (H, W, Cin, Cout), meta):
        _lib.check(_lib.lib().qbnn_block_down_i8_mc(_lib.ptr(x.data), x.sample_stride(), x.scale, x.zero_point, B, H, Cin, a_hi,
                                                    C.byref(d), _lib.ptr(y), y[0].numel(), S, _lib.current_stream()))
    return MCQTensor(y, blk.add.add.scale, blk.add.add.zero_point)


def pool_out_enabled():
    """The network's last block hands the head its AvgPool2d(4) instead of the 4 x 4 map (QBNN_BLOCK_POOL_OUT; QBNN_HEAD_POOL=0 for the A/B);
    the flag is served by the ring form of the 192-channel block only."""
    return os.environ.get("QBNN_HEAD_POOL", "1") != "0"


def run_identity_chain(blocks, x, stem=None, pool_out=False):
    """Identity BasicBlocks (no shortcut conv) in ONE persistent fused kernel per call (qbnn_block_chain_i8_mc; 1 or 2 blocks at
    24 / 48 channels, one block per launch at 96 / 192 -- longer lists are walked here):
    activations stay in LDS between stem.0, stem.3 and the residual add.  Same results as calling the blocks.
    `stem` = (layers.0 module, its sampled weights, im2col patches [B, 1024, 32], input scale): the network's first conv
    runs inside the same kernel (qbnn_stem_chain_i8_mc) and `x` only carries conv0's output qparams / shape."""
    S = _MC.samples
    if stem is None and (len(blocks) > 2 or (len(blocks) == 2 and x.data.shape[4] >= 96)):
        h = x
        step = 1 if x.data.shape[4] >= 96 else 2
        for i in range(0, len(blocks), step):
            h = run_identity_chain(blocks[i:i + step], h, pool_out=pool_out and i + step >= len(blocks))
        return h
    dev = x.data.device if stem is None else stem[2].device
    descs = (_lib.BlockDesc * len(blocks))()
    keep = []
    for d, blk in zip(descs, blocks):
        assert len(blk.shortcut) == 0
        ca, cb = blk.stem[0], blk.stem[3]
        wa, wb = ca.sample_weights(dev), cb.sample_weights(dev)
        pa, pb = ca._ensure_packed(dev), cb._ensure_packed(dev)
        keep += [wa, wb]
        d.w_a, d.w_a_sample_stride, d.bias_a = wa.data_ptr(), wa.shape[1], (pa["bias"].data_ptr() if pa["bias"] is not None else None)
        d.s_wa, d.z_wa, d.s_a, d.z_a = ca.add_weight.scale, ca.add_weight.zero_point, ca.scale, ca.zero_point
        d.w_b, d.w_b_sample_stride, d.bias_b = wb.data_ptr(), wb.shape[1], (pb["bias"].data_ptr() if pb["bias"] is not None else None)
        d.s_wb, d.z_wb, d.s_b, d.z_b = cb.add_weight.scale, cb.add_weight.zero_point, cb.scale, cb.zero_point
        d.s_o, d.z_o = blk.add.add.scale, blk.add.add.zero_point
        assert ca.layout == cb.layout, "a block's two convs share one packed layout"
        d.w_layout = ca.layout
    a_hi = UINT_BOUNDS[blocks[0].args.activation_precision][1]
    nw = lambda l: l._packed["cout"] * l._packed["k"]
    if stem is not None:
        l0, w0, col, s_in = stem
        B, H, W, Cc = col.shape[0], 32, 32, 24
        pk0 = l0._ensure_packed(dev)
        y = torch.empty((S, B, H, W, Cc), dtype=torch.uint8, device=dev)
        key = "stem + block_chain_i8 x%d %dx%d c%d" % (len(blocks), H, W, Cc)
        meta = dict(fused=True, convs=[(H, 3, Cc, 3, 1, nw(l0))] + [(H, Cc, Cc, 3, 1, nw(c)) for b in blocks for c in (b.stem[0], b.stem[3])],
                    res_convs=[2 + 2 * i for i in range(len(blocks))])
        with timed(key, meta):
            _lib.check(_lib.lib().qbnn_stem_chain_i8_mc(_lib.ptr(col), B, _lib.ptr(w0), w0.shape[1], _lib.ptr(pk0["bias"]), s_in,
                                                        l0.add_weight.scale, l0.add_weight.zero_point, l0.scale, l0.zero_point, a_hi,
                                                        descs, len(blocks), _lib.ptr(y), y[0].numel(), S, _lib.current_stream()))
        last = blocks[-1].add.add
        return MCQTensor(y, last.scale, last.zero_point)
    _, B, H, W, Cc = x.data.shape
    pool_out = pool_out and len(blocks) == 1 and (H, Cc) == (4, 192) and blocks[0].stem[0].layout == LAYOUT_MFMA32
    if pool_out:
        descs[0].flags = 1          # QBNN_BLOCK_POOL_OUT: y = AvgPool2d(4) of the block output, [S, B, 1, 1, C]
    y = torch.empty((S, B, 1, 1, Cc) if pool_out else (S, B, H, W, Cc), dtype=torch.uint8, device=dev)
    key = "block_chain_i8 x%d %dx%d c%d" % (len(blocks), H, W, Cc)
    meta = dict(fused=True, convs=[(H, Cc, Cc, 3, 1, nw(c)) for b in blocks for c in (b.stem[0], b.stem[3])],
                res_convs=[1 + 2 * i for i in range(len(blocks))])
    with timed(key, meta):
        _lib.check(_lib.lib().qbnn_block_chain_i8_mc(_lib.ptr(x.data), x.sample_stride(), x.scale, x.zero_point, B, H, Cc, a_hi,
                                                     descs, len(blocks), _lib.ptr(y), y[0].numel(), S, _lib.current_stream()))
    last = blocks[-1].add.add
    return MCQTensor(y, last.scale, last.zero_point)


class _QParamsOnly:
    """(scale, zero point) of a tensor that is never materialised (conv0's output inside the fused stem kernel)."""

    def __init__(self, scale, zero_point):
        self.scale, self.zero_point, self.data = scale, zero_point, None


class QuantStub(nn.Module):
    def __init__(self):
        super().__init__()
        self.scale, self.zero_point = 1.0, 0


class ConvNetwork_ResNet(nn.Module):
    """reference models_bbb.py:191-256 (narrow ResNet-18: 24/48/96/192), converted int8 form."""

    def __init__(self, input_size, output_size, q, args, deterministic=False):
        super().__init__()
        if not q:
            raise NotImplementedError("this class is the converted int8 graph; ModelFactory routes q=False to models_f32.ConvNetwork_ResNet")
        check_bits(args)
        self.args, self.q = args, q
        self.deterministic = deterministic       # True: an ensemble member (standard quantised layers, no weight noise)
        self.in_planes = 24
        self.init_channels = input_size[1]
        self.output_size = int(output_size)
        ident = nn.Identity
        self.layers = nn.ModuleList([])
        self.layers.append(ConvReLU2d(self.init_channels, 24, 3, stride=1, padding=1, bias=True, args=args))
        self.layers.append(ident())
        self.layers.append(ident())
        self.layers.append(self._make_layer(24, 2, 1))
        self.layers.append(self._make_layer(48, 2, 2))
        self.layers.append(self._make_layer(96, 2, 2))
        self.layers.append(self._make_layer(192, 2, 2))
        self.layers.append(nn.AvgPool2d(4))
        self.layers.append(ident())            # Flatten
        self.layers.append(Linear(192 * BasicBlock.expansion, output_size, bias_=False, args=args))
        self.quant = QuantStub()
        self.dequant = ident()
        self.fuse_blocks = True        # False: every conv as its own launch (layer-level C ABI), for A/B checks
        self.fuse_stem = os.environ.get("QBNN_NO_STEM_FUSION", "0") != "1"    # layers.0 inside the layer-1 chain kernel
        # Philox tensor ids = execution order of the stochastic layers
        for i, m in enumerate(self.stochastic_layers()):
            m.layer_id = i
            m.deterministic = deterministic

    def _make_layer(self, planes, num_blocks, stride):
        strides = [stride] + [1] * (num_blocks - 1)
        blocks = []
        for st in strides:
            blocks.append(BasicBlock(self.in_planes, planes, st, self.q, self.args))
            self.in_planes = planes * BasicBlock.expansion
        return nn.ModuleList(blocks)

    def _apply_layouts(self, fused, fused_stem=None):
        """Packed weight layouts per execution path.  The fused 16-wave kernel of the 48-channel identity block (csrc/qbnn_c48.hip) takes
        its two convs as (24 + 1)-row tile halves (QBNN_LAYOUT_MFMA32_N24), the 16-wave layer-1 kernel (csrc/qbnn_w16.hip, behind the fused
        stem) its four 24-channel convs with the kernel rows' tails gathered (QBNN_LAYOUT_MFMA32_TAIL: 7 k-steps instead of 9); the
        layer-level conv kernel -- the recording / un-fused path -- and every other fused kernel take MFMA32.  A switch re-packs mu / sigma
        once (layers.set_layout).  QBNN_C48=0 / QBNN_W16=0: MFMA32 there (the kernels of rounds 2 - 4, for A/B checks)."""
        n24 = fused and os.environ.get("QBNN_C48", "1") != "0"
        blk = self.layers[4][1]
        for c in (blk.stem[0], blk.stem[3]):
            c.set_layout(LAYOUT_MFMA32_N24 if n24 else LAYOUT_MFMA32)
        d24 = fused and os.environ.get("QBNN_D24", "1") != "0" and len(self.layers[4][0].shortcut) > 0      # the 24 -> 48 down block's 16-wave kernel
        blk = self.layers[4][0]
        blk.stem[0].set_layout(LAYOUT_MFMA32_N24_TAIL if d24 else LAYOUT_MFMA32)
        blk.stem[3].set_layout(LAYOUT_MFMA32_N24 if d24 else LAYOUT_MFMA32)
        if len(blk.shortcut):
            blk.shortcut[0].set_layout(LAYOUT_MFMA32_N24 if d24 else LAYOUT_MFMA32)
        if fused_stem is None:
            fused_stem = fused and self.fuse_stem and len(self.layers[3][0].shortcut) == 0
        tail = fused_stem and w16_enabled() and len(self.layers[3]) == 2
        for blk in self.layers[3]:
            for c in (blk.stem[0], blk.stem[3]):
                c.set_layout(LAYOUT_MFMA32_TAIL if tail else LAYOUT_MFMA32)

    def stochastic_layers(self):
        out = [self.layers[0]]
        for li in (3, 4, 5, 6):
            for blk in self.layers[li]:
                out += [blk.stem[0], blk.stem[3]] + ([blk.shortcut[0]] if len(blk.shortcut) else [])
        out.append(self.layers[9])
        return out

    def stochastic_layer_names(self):
        names = ["layers.0"]
        for li in (3, 4, 5, 6):
            for bi, blk in enumerate(self.layers[li]):
                names += [f"layers.{li}.{bi}.stem.0", f"layers.{li}.{bi}.stem.3"]
                if len(blk.shortcut):
                    names.append(f"layers.{li}.{bi}.shortcut.0")
        names.append("layers.9")
        return names

    # ---- reference-format checkpoint ingestion (flat numpy dict; see tests/golden/make_golden.py:flat_state)
    def load_reference_state(self, state):
        for name, m in zip(self.stochastic_layer_names(), self.stochastic_layers()):
            m.load_reference_state(state, name + ".")
        for li in (3, 4, 5, 6):
            for bi, blk in enumerate(self.layers[li]):
                p = f"layers.{li}.{bi}.add.add."
                blk.add.add = QFunctional(state[p + "scale"], state[p + "zero_point"])
        self.quant.scale = float(np.asarray(state["quant.scale"]).reshape(-1)[0])
        self.quant.zero_point = int(np.asarray(state["quant.zero_point"]).reshape(-1)[0])
        return self

    def n_weights(self):
        return sum(int(np.prod(m.weight.shape)) for m in self.stochastic_layers())

    # ---- forward
    def quantize_input(self, x):
        """QuantStub + clamp_activation (models_bbb.py:227-229): fp32 NCHW -> MCQTensor shared by all samples."""
        if x.device.type != "cuda":
            raise RuntimeError("qbnn models run on an MI355X only (no CPU fallback)")
        x = x.to(torch.float32).contiguous()
        B, Cc, H, W = x.shape
        out = torch.empty((1, B, H, W, Cc), dtype=torch.uint8, device=x.device)
        a_hi = UINT_BOUNDS[self.args.activation_precision][1]
        with timed("quantize_input"):
            _lib.check(_lib.lib().qbnn_quantize_input_nchw(_lib.ptr(x), B, Cc, H, W, self.quant.scale, self.quant.zero_point, a_hi,
                                                           _lib.ptr(out), _lib.current_stream()))
        return MCQTensor(out, self.quant.scale, self.quant.zero_point, shared=True)

    def forward_mc(self, x, record=None):
        """All S samples of the current mc_context: returns per-sample softmax probabilities [S, B, classes]."""
        S = _MC.samples
        L = _lib.lib()
        dev = x.device
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, 32, 32):
            raise NotImplementedError("conv_resnet_bbb expects 3x32x32 inputs")
        l0 = self.layers[0]
        fuse_stem = self.fuse_blocks and self.fuse_stem and record is None and len(self.layers[3][0].shortcut) == 0
        B, Cc, H, W = x.shape
        col = torch.empty((B, H * W, 32), dtype=torch.int8, device=dev)
        self._apply_layouts(self.fuse_blocks and record is None, fuse_stem)
        if fuse_stem and 0 <= self.quant.zero_point <= 127:
            # QuantStub + clamp + the layer-0 patch gather in ONE pass over the fp32 input (the quantised image itself is not needed:
            # layers.0 runs on the patches inside the layer-1 kernel)
            if x.device.type != "cuda":
                raise RuntimeError("qbnn models run on an MI355X only (no CPU fallback)")
            xf = x.to(torch.float32).contiguous()
            xq = _QParamsOnly(self.quant.scale, self.quant.zero_point)
            a_hi = UINT_BOUNDS[self.args.activation_precision][1]
            with timed("quantize + im2col3x3_c3"):
                _lib.check(L.qbnn_quantize_im2col3x3_c3_multi(_lib.ptr(xf), B, H, W, (C.c_float * 1)(self.quant.scale), (C.c_int32 * 1)(self.quant.zero_point), 1, a_hi,
                                                              _lib.ptr(col), col.numel(), _lib.current_stream()))
        else:
            xq = self.quantize_input(x)
            if record is not None:
                record["quant.out"] = xq.data[0]
            with timed("im2col3x3_c3"):
                _lib.check(L.qbnn_im2col3x3_c3(_lib.ptr(xq.data), B, H, W, xq.zero_point, _lib.ptr(col), _lib.current_stream()))
        if not self.deterministic:
            sample_all_weights(self.stochastic_layers(), dev)      # one launch for the 21 layers of this MC batch
        if fuse_stem:
            # layers.0 runs inside the layer-1 chain kernel: its output (the largest activation of the net) stays on chip
            h = run_identity_chain(list(self.layers[3]), _QParamsOnly(l0.scale, l0.zero_point), stem=(l0, l0.sample_weights(dev), col, xq.scale))
        else:
            h = l0._conv(xq, l0.sample_weights(dev), S, im2col=col)
        if record is not None:
            record["layers.0.out"] = h.data
        for li in (3, 4, 5, 6):
            if fuse_stem and li == 3:
                continue
            blocks = list(self.layers[li])
            if self.fuse_blocks and record is None:
                # identity blocks run as fused persistent chains; a down-sampling block 0 runs layer by layer
                if len(blocks[0].shortcut) == 0:
                    h = run_identity_chain(blocks, h)
                else:
                    h = run_down_block(blocks[0], h)
                    h = run_identity_chain(blocks[1:], h, pool_out=(li == 6 and pool_out_enabled()))      # the head pools what the last block leaves
                continue
            for bi, blk in enumerate(blocks):
                if not self.fuse_blocks:
                    h = blk(h)
                else:
                    h = run_identity_chain([blk], h) if len(blk.shortcut) == 0 else run_down_block(blk, h)
                if record is not None:
                    record[f"layers.{li}.{bi}.out"] = h.data
        fc = self.layers[9]
        wfc = fc.sample_weights(dev)
        pk = fc._ensure_packed(dev)
        probs = torch.empty((S, B, self.output_size), dtype=torch.float32, device=dev)
        d = _lib.HeadDesc()
        d.B, d.k, d.C, d.N = B, h.data.shape[2], h.data.shape[4], self.output_size
        d.s_x, d.z_x = h.scale, h.zero_point
        d.s_w, d.z_w = fc.add_weight.scale, fc.add_weight.zero_point
        d.s_y, d.z_y = fc.scale, fc.zero_point
        d.a_hi, d.has_bias = UINT_BOUNDS[self.args.activation_precision][1], int(pk["bias"] is not None)
        with timed("head_i8"):
            _lib.check(L.qbnn_head_i8_mc(_lib.ptr(h.data), h.sample_stride(), _lib.ptr(wfc), wfc.shape[1], _lib.ptr(pk["bias"]),
                                         _lib.ptr(probs), S, C.byref(d), _lib.current_stream()))
        return probs

    def forward(self, x):
        """Reference call contract: one stochastic forward -> probs [B, classes] (models_bbb.py:226-245).
        Under mc_context(samples=S) it evaluates sample `sample_begin` only; use forward_mc for all S."""
        with mc_context(1, _MC.seed, _MC.sample_begin, _MC.eps):
            return self.forward_mc(x)[0]


class Network(nn.Module):
    """reference sgld.Network (src/models/stochastic/sgld/models_sgld.py:214-288) in evaluation form
    (training_mode=False): `args.samples` deterministic int8 members; the "MC samples" are the members.
    forward() round-robins the members through `self.counter` exactly like the reference; forward_mc() evaluates the
    members [sample_begin, sample_begin + S) of the active mc_context (so mc_predict shards members over GPUs)."""

    def __init__(self, input_size, output_size, q, args, training_mode=True):
        super().__init__()
        if training_mode:
            raise NotImplementedError("training is out of scope (inference hot path only)")
        if args.model == "conv_resnet_sgld":
            template = ConvNetwork_ResNet
        elif args.model == "conv_lenet_sgld":
            from .models_small import ConvNetwork_LeNet as template
        elif args.model == "linear_sgld":
            from .models_small import LinearNetwork as template
        else:
            raise NotImplementedError("Other templates not implemented!")
        self.args, self.q, self.training_mode = args, q, training_mode
        self.output_size = int(output_size)
        self.ensemble = nn.ModuleList([template(input_size, output_size, q, args, deterministic=True) for _ in range(args.samples)])
        self.regression = args.model == "linear_sgld"        # members return (mu, var); no softmax (models_sgld.py:286-287)
        self.counter = 0
        # A member's forward is ~12 launches on one sample's worth of work: launch-bound.  Members are deterministic (no
        # seed, no sample offset), so each member's launch chain is captured once per input shape into a HIP graph and
        # replayed (QBNN_NO_GRAPHS=1: always launch eagerly).
        self.use_graphs = os.environ.get("QBNN_NO_GRAPHS", "0") != "1"
        self._graphs = {}
        self._streams = []
        # members side by side in fused multi-call launches (the ResNet template's default); False: one launch chain per member
        self.fused_members = args.model == "conv_resnet_sgld"
        self.prepared_launches = os.environ.get("QBNN_ENSEMBLE_BY_VALUE", "0") != "1"     # argument blocks in device memory: one launch per stage for all members
        self._plans = {}

    def load_reference_state(self, member_states):
        assert len(member_states) == len(self.ensemble)
        for m, st in zip(self.ensemble, member_states):
            m.load_reference_state(st)
        self._graphs = {}
        self._plans = {}
        return self

    def load_ensemble(self, args, path=None, special_info=""):
        """reference models_sgld.py:245-261: one checkpoint file per member (`weights_<special><n>.pt` under `path`, natural
        order, the last `args.samples`), each read like `utils.load_model` (prefixes `module.` / `main_net.` stripped)."""
        from .checkpoint import ensemble_files, load_state
        path = args.save if path is None else path
        self.sample_names = ensemble_files(path, args.samples, special_info)
        if len(self.sample_names) < len(self.ensemble):
            raise RuntimeError(f"found {len(self.sample_names)} member checkpoints under {path}, need {len(self.ensemble)}")
        return self.load_reference_state([load_state(os.path.join(path, n)) for n in self.sample_names])

    @staticmethod
    def _graph_key(idx, x):
        """(the epoch: a layout switch or a reloaded state drops the packed weights a captured chain points at)"""
        from .layers import state_epoch
        return (idx, tuple(x.shape), x.dtype, x.device.index, state_epoch())

    def _member_forward(self, idx, x, record=None):
        """[1, B, C] probabilities of member `idx`; replays the member's captured launch chain when there is one."""
        from . import layers as _layers
        member = self.ensemble[idx]
        if record is not None or not self.use_graphs or self.regression or _layers.PROFILE is not None or x.device.type != "cuda":
            with mc_context(1, 0, 0):
                return member.forward_mc(x, record=record)
        ent = self._graphs.get(self._graph_key(idx, x))
        if ent is None:
            with mc_context(1, 0, 0):
                member.forward_mc(x)                        # eager once: packs the weights, fills every cache the chain reads
                try:
                    static_x = x.clone()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        static_y = member.forward_mc(static_x)
                    ent = (graph, static_x, static_y)
                except Exception as e:                      # capture refused: keep launching eagerly (same kernels), but say so
                    import warnings
                    warnings.warn(f"qbnn ensemble: HIP graph capture of member {idx} failed ({e!r}); launching eagerly")
                    GRAPH_FALLBACKS.append(f"member {idx}: {e!r}")     # a benchmark must not time the fallback silently: bench.py exits non-zero
                    torch.cuda.synchronize()
                    ent = False
            # the key AFTER the warm-up forward (it may switch layouts and bump the epoch itself: a key taken before it would be dead on arrival
            # and the chain captured twice); entries of older epochs go -- each pins a captured graph with its static input / output
            key = self._graph_key(idx, x)
            self._graphs = {k: v for k, v in self._graphs.items() if k[-1] == key[-1]}
            self._graphs[key] = ent
        if ent is False:
            with mc_context(1, 0, 0):
                return member.forward_mc(x)
        graph, static_x, static_y = ent
        static_x.copy_(x)
        graph.replay()
        return static_y.clone()

    # ---- members side by side: one fused multi-call launch per layer group (qbnn_*_multi) instead of ~12 launches per member
    def _member_plan(self, idx, B, dev):
        """Static buffers + the call arrays of the fused launches for members `idx` at batch B (built once, replayed)."""
        from .layers import state_epoch
        mem = [self.ensemble[j] for j in idx]
        for m in mem:
            m._apply_layouts(True)
        key = (tuple(idx), B, dev.index, state_epoch())      # (the epoch: see _member_forward)
        plan = self._plans.get(key)
        if plan is not None:
            return plan
        M, L = len(idx), _lib.lib()
        a_hi = UINT_BOUNDS[self.args.activation_precision][1]
        keep = []
        u8 = lambda *shape: torch.empty(shape, dtype=torch.uint8, device=dev)
        # layer-0 patches: one tensor per DISTINCT input quantisation.  The QuantStub observes the network input, not the member's weights,
        # so members calibrated on the same data share (scale, zero point) -- and then the patches (8 MB at B = 256, L2-resident) too.
        qin = sorted({(float(m.quant.scale), int(m.quant.zero_point)) for m in mem})
        col_of = [qin.index((float(m.quant.scale), int(m.quant.zero_point))) for m in mem]
        col = torch.empty((len(qin), B, 1024, 32), dtype=torch.int8, device=dev)
        acts = {3: u8(M, B, 32, 32, 24), "4d": u8(M, B, 16, 16, 48), 4: u8(M, B, 16, 16, 48), "5d": u8(M, B, 8, 8, 96), 5: u8(M, B, 8, 8, 96),
                "6d": u8(M, B, 4, 4, 192), 6: u8(M, B, 4, 4, 192)}
        probs = torch.empty((M, B, self.output_size), dtype=torch.float32, device=dev)
        scales = (C.c_float * len(qin))(*[q[0] for q in qin])
        zps = (C.c_int32 * len(qin))(*[q[1] for q in qin])
        with mc_context(1, 0, 0):
            # layers.0 + layers.3 (two identity blocks) behind the fused stem
            stem_calls = (_lib.ChainCall * M)()
            for i, m in enumerate(mem):
                l0 = m.layers[0]
                pk0 = l0._ensure_packed(dev)
                descs = (_lib.BlockDesc * 2)()
                for d, blk in zip(descs, m.layers[3]):
                    _fill_block_desc(d, blk, dev, keep)
                keep.append(descs)
                c = stem_calls[i]
                c.blocks, c.y, c.y_sample_stride, c.n_samples = descs, acts[3][i].data_ptr(), acts[3][i].numel(), 1
                c.im2col, c.w0_packed, c.w0_sample_stride = col[col_of[i]].data_ptr(), pk0["mu"].data_ptr(), 0
                c.bias0 = pk0["bias"].data_ptr() if pk0["bias"] is not None else None
                c.s_in, c.s_w0, c.z_w0, c.s_y0, c.z_y0 = m.quant.scale, l0.add_weight.scale, l0.add_weight.zero_point, l0.scale, l0.zero_point
            steps = [("stem", stem_calls)]
            prev = 3
            for li, (H, Cin) in ((4, (32, 24)), (5, (16, 48)), (6, (8, 96))):
                dcalls, ccalls = (_lib.DownCall * M)(), (_lib.ChainCall * M)()
                for i, m in enumerate(mem):
                    b0, b1 = m.layers[li][0], m.layers[li][1]
                    prev_add = m.layers[prev][-1].add.add
                    dd = _lib.DownDesc()
                    _fill_block_desc(dd.blk, b0, dev, keep)
                    cs = b0.shortcut[0]
                    ps = cs._ensure_packed(dev)
                    dd.w_s, dd.w_s_sample_stride, dd.bias_s = ps["mu"].data_ptr(), 0, (ps["bias"].data_ptr() if ps["bias"] is not None else None)
                    dd.s_ws, dd.z_ws, dd.s_s, dd.z_s = cs.add_weight.scale, cs.add_weight.zero_point, cs.scale, cs.zero_point
                    keep.append(dd)
                    k = dcalls[i]
                    xin = acts[prev][i]
                    k.x, k.x_sample_stride, k.s_x, k.z_x = xin.data_ptr(), xin.numel(), prev_add.scale, prev_add.zero_point
                    k.desc, k.y, k.y_sample_stride, k.n_samples = C.pointer(dd), acts[f"{li}d"][i].data_ptr(), acts[f"{li}d"][i].numel(), 1
                    d1 = (_lib.BlockDesc * 1)()
                    _fill_block_desc(d1[0], b1, dev, keep)
                    keep.append(d1)
                    q = ccalls[i]
                    q.x, q.x_sample_stride, q.s_x, q.z_x = acts[f"{li}d"][i].data_ptr(), acts[f"{li}d"][i].numel(), b0.add.add.scale, b0.add.add.zero_point
                    q.blocks, q.y, q.y_sample_stride, q.n_samples = d1, acts[li][i].data_ptr(), acts[li][i].numel(), 1
                steps += [("down", dcalls, H, Cin), ("chain", ccalls, H // 2, 2 * Cin)]
                prev = li
            hcalls = (_lib.HeadCall * M)()
            for i, m in enumerate(mem):
                fc, last = m.layers[9], m.layers[6][-1].add.add
                pk = fc._ensure_packed(dev)
                hd = _lib.HeadDesc()
                hd.B, hd.k, hd.C, hd.N = B, 4, 192, self.output_size
                hd.s_x, hd.z_x, hd.s_w, hd.z_w = last.scale, last.zero_point, fc.add_weight.scale, fc.add_weight.zero_point
                hd.s_y, hd.z_y, hd.a_hi, hd.has_bias = fc.scale, fc.zero_point, a_hi, int(pk["bias"] is not None)
                keep.append(hd)
                h = hcalls[i]
                h.x, h.x_sample_stride, h.w, h.w_sample_stride = acts[6][i].data_ptr(), acts[6][i].numel(), pk["mu"].data_ptr(), 0
                h.bias, h.probs, h.n_samples, h.desc = (pk["bias"].data_ptr() if pk["bias"] is not None else None), probs[i].data_ptr(), 1, C.pointer(hd)
        # the argument blocks of every stage go to device memory once (qbnn_*_multi_prepare): a stage is then ONE launch for all M members
        # (by value, 4 KiB of kernel arguments hold 4 - 8 of them).  Not under stream capture: the plan is built by the first eager call.
        dev_steps = []
        if self.prepared_launches and not torch.cuda.is_current_stream_capturing():
            for step in steps:
                if step[0] == "down":
                    buf = torch.empty(int(L.qbnn_down_multi_args_bytes(M)), dtype=torch.uint8, device=dev)
                    _lib.check(L.qbnn_block_down_i8_multi_prepare(step[1], M, B, a_hi, _lib.ptr(buf), _lib.current_stream()))
                else:
                    nb = 2 if step[0] == "stem" else 1
                    buf = torch.empty(int(L.qbnn_chain_multi_args_bytes(M, nb)), dtype=torch.uint8, device=dev)
                    _lib.check(L.qbnn_block_chain_i8_multi_prepare(step[1], M, int(step[0] == "stem"), B, a_hi, nb, _lib.ptr(buf), _lib.current_stream()))
                dev_steps.append(buf)
        plan = dict(M=M, col=col, acts=acts, probs=probs, scales=scales, zps=zps, steps=steps, dev_steps=dev_steps, head=hcalls, keep=keep, a_hi=a_hi)
        self._plans = {k: v for k, v in self._plans.items() if k[-1] == key[-1]}      # plans of older epochs pin activation buffers and device argument blocks
        self._plans[key] = plan
        return plan

    def _forward_members_fused(self, idx, x):
        """[len(idx), B, C] probabilities of members `idx`: 10 launches for ALL of them (input quantisation + patches, stem + layer 1,
        three down blocks, three identity blocks, head), each member with its own weights and quantisation parameters."""
        if x.device.type != "cuda":
            raise RuntimeError("qbnn models run on an MI355X only (no CPU fallback)")
        x = x.to(torch.float32).contiguous()
        B, Cc, H, W = x.shape
        if Cc != 3 or H != 32 or W != 32:
            raise NotImplementedError("conv_resnet_sgld expects 3x32x32 inputs")
        p = self._member_plan(idx, B, x.device)
        L, st, M = _lib.lib(), _lib.current_stream(), p["M"]
        with timed("ensemble quantize + im2col"):
            _lib.check(L.qbnn_quantize_im2col3x3_c3_multi(_lib.ptr(x), B, H, W, p["scales"], p["zps"], p["col"].shape[0], p["a_hi"], _lib.ptr(p["col"]),
                                                          p["col"][0].numel(), st))
        for si, step in enumerate(p["steps"]):
            dargs = _lib.ptr(p["dev_steps"][si]) if p["dev_steps"] else None
            if step[0] == "stem":
                with timed("ensemble stem + layer 1"):
                    if dargs:
                        _lib.check(L.qbnn_block_chain_i8_multi_launch(dargs, M, 1, B, 32, 24, p["a_hi"], step[1][0].blocks[0].w_layout, 2, 1, st))
                    else:
                        _lib.check(L.qbnn_block_chain_i8_multi(step[1], M, 1, B, 32, 24, p["a_hi"], 2, st))
            elif step[0] == "down":
                with timed("ensemble down %d" % step[3]):
                    if dargs:
                        _lib.check(L.qbnn_block_down_i8_multi_launch(dargs, M, B, step[2], step[3], step[1][0].desc.contents.blk.w_layout, 1, st))
                    else:
                        _lib.check(L.qbnn_block_down_i8_multi(step[1], M, B, step[2], step[3], p["a_hi"], st))
            else:
                with timed("ensemble chain %d" % step[3]):
                    if dargs:
                        _lib.check(L.qbnn_block_chain_i8_multi_launch(dargs, M, 0, B, step[2], step[3], p["a_hi"], step[1][0].blocks[0].w_layout, 1, 1, st))
                    else:
                        _lib.check(L.qbnn_block_chain_i8_multi(step[1], M, 0, B, step[2], step[3], p["a_hi"], 1, st))
        with timed("ensemble head"):
            _lib.check(L.qbnn_head_i8_multi(p["head"], M, st))
        return p["probs"].clone()

    def forward_mc(self, x, record=None):
        n = len(self.ensemble)
        idx = [(_MC.sample_begin + i) % n for i in range(_MC.samples)]
        if record is None and self.fused_members and x.device.type == "cuda" and len(set(idx)) == len(idx):
            return self._forward_members_fused(idx, x)
        graphed = (record is None and self.use_graphs and not self.regression and x.device.type == "cuda" and len(idx) > 1 and
                   all(self._graphs.get(self._graph_key(j, x)) for j in idx))
        if not graphed:
            outs = [self._member_forward(j, x, record=record if i == 0 else None) for i, j in enumerate(idx)]
            if self.regression:
                return torch.cat([o[0] for o in outs], 0), torch.cat([o[1] for o in outs], 0)
            return torch.cat(outs, 0)
        # every member has a captured chain: replay them round-robin on side streams (8; QBNN_ENSEMBLE_STREAMS) -- one member's late layers
        # fill 16-64 of the 256 CUs, so independent members overlap
        main = torch.cuda.current_stream()
        if not self._streams:
            self._streams = [torch.cuda.Stream() for _ in range(int(os.environ.get("QBNN_ENSEMBLE_STREAMS", "8")))]
        outs = [None] * len(idx)
        for st in self._streams:
            st.wait_stream(main)
        for i, j in enumerate(idx):
            st = self._streams[j % len(self._streams)]      # by MEMBER: two replays of one member's graph (samples > members) share its
                                                            # static buffers and must not run concurrently
            with torch.cuda.stream(st):
                outs[i] = self._member_forward(j, x)
                outs[i].record_stream(main)
        for st in self._streams:
            main.wait_stream(st)
        return torch.cat(outs, 0)

    def forward(self, x):
        with mc_context(1, 0, 0):
            y = self._member_forward(self.counter, x)
        y = (y[0][0], y[1][0]) if self.regression else y[0]
        self.counter += 1
        if self.counter >= self.args.samples:
            self.counter = 0
        return y


class ModelFactory:
    """reference src/models/__init__.py:12-40 (names kept)."""

    @staticmethod
    def get_model(model, input_size, output_size, q, args, training_mode=True):
        if q and getattr(args, "qat_eval", False) and model in ("conv_resnet_bbb", "conv_lenet_bbb", "linear_bbb"):
            # the prepared-but-not-converted model (quant_utils.prepare_model): fake-quant evaluation with live observers
            from . import models_qat
            cls = {"conv_resnet_bbb": models_qat.ConvNetwork_ResNet, "conv_lenet_bbb": models_qat.ConvNetwork_LeNet,
                   "linear_bbb": models_qat.LinearNetwork}[model]
            return cls(input_size, output_size, q, args)
        if q and getattr(args, "qat_eval", False) and model in ("linear_mc", "conv_lenet_mc", "conv_resnet_mc", "linear_sgld", "conv_lenet_sgld", "conv_resnet_sgld"):
            # quant_utils.prepare_model's other branch (`prepare_qat`, :139-140): the MC-Dropout graphs and the SGHMC member template
            from . import models_qat_mc
            return models_qat_mc.get_model(model, input_size, output_size, q, args)
        if model == "conv_resnet_bbb":
            if not q:
                from .models_f32 import ConvNetwork_ResNet as ConvNetwork_ResNetF32
                return ConvNetwork_ResNetF32(input_size, output_size, q, args)
            return ConvNetwork_ResNet(input_size, output_size, q, args)
        if "sgld" in model:
            return Network(input_size, output_size, q, args, training_mode)
        if model == "linear_bbb":
            if q:
                from .models_small import LinearNetwork as LinearNetworkBBBQ
                return LinearNetworkBBBQ(input_size, output_size, q, args)
            from .models_f32 import LinearNetwork as LinearNetworkBBB
            return LinearNetworkBBB(input_size, output_size, q, args)
        if model == "conv_lenet_bbb":
            if not q:
                from .models_f32 import ConvNetwork_LeNet as ConvNetwork_LeNetF32
                return ConvNetwork_LeNetF32(input_size, output_size, q, args)
            from .models_small import ConvNetwork_LeNet as ConvNetwork_LeNetBBB
            return ConvNetwork_LeNetBBB(input_size, output_size, q, args)
        if model in ("linear_mc", "conv_lenet_mc", "conv_resnet_mc"):
            # q=True: the converted int8 graph (models_mc); q=False: the float graph with the FloatFunctional dropout (models_mc_f32)
            from . import models_mc, models_mc_f32
            mod = models_mc if q else models_mc_f32
            cls = {"linear_mc": mod.LinearNetwork, "conv_lenet_mc": mod.ConvNetwork_LeNet, "conv_resnet_mc": mod.ConvNetwork_ResNet}[model]
            return cls(input_size, output_size, q, args)
        raise NotImplementedError("Other models not implemented")
