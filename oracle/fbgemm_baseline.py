"""CPU baseline of kind "torch-fbgemm": the converted int8 conv_resnet_bbb driven through PyTorch's own quantised CPU
operators (ATen + FBGEMM), i.e. the same third-party arithmetic the reference executes on its CPU path.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (like everything under oracle/): imported by tests/ and by bench.py's
`cpu_baseline` leg, never by the product.  It does not import the reference; it issues the op sequence the reference's
layers issue, from a flat reference-format state dict:

  per stochastic layer (reference conv_q.py:107-125 / :189-209, linear_q.py:80-94):
      eps   = FloatTensor(shape).normal_(0, 1)                          (or an injected eps: parity mode)
      eps_q = torch.quantize_per_tensor(eps, 3/127, 0, qint8)
      W     = quantized.add(weight, quantized.mul(std, eps_q, s_m, z_m), s_a, z_a)
      W     = clamp to INT_BOUNDS[w_bits]                               (src/utils.py:32-37)
      y     = quantized.conv2d(_relu)(x, conv2d_prepack(W, bias, ...), scale, zero_point)   -- re-packed every call
      y     = clamp to UINT_BOUNDS[a_bits]                              (src/utils.py:25-30, after every module)
  graph (models_bbb.py:170-183, :226-245): quant -> conv-relu -> 8 BasicBlocks (stem.0, stem.3, [shortcut.0], Add, ReLU)
      -> AvgPool2d(4) -> flatten -> linear -> dequant -> softmax.

With the oracle's Philox eps injected it reproduces the golden probabilities recorded from the reference
(tests/test_oracle_golden.py::test_fbgemm_harness_matches_golden), so what bench.py times IS the reference's arithmetic.
"""
import numpy as np
import torch

NOISE_SCALE, NOISE_ZERO_POINT = 0.02362204724, 0           # reference bbb/quantized/__init__.py:1-2
UINT_BOUNDS = {8: [0, 255], 7: [0, 127], 6: [0, 63], 5: [0, 31], 4: [0, 15], 3: [0, 7], 2: [0, 3]}
INT_BOUNDS = {8: [-128, 127], 7: [-64, 63], 6: [-32, 31], 5: [-16, 15], 4: [-8, 7], 3: [-4, 3], 2: [-2, 1]}


def _layer_table():
    t = [("layers.0.", 1, 1, True)]
    for li, first_stride in ((3, 1), (4, 2), (5, 2), (6, 2)):
        for bi in (0, 1):
            st = first_stride if bi == 0 else 1
            t.append((f"layers.{li}.{bi}.stem.0.", st, 1, True))
            t.append((f"layers.{li}.{bi}.stem.3.", 1, 1, False))
            if bi == 0 and li != 3:
                t.append((f"layers.{li}.{bi}.shortcut.0.", st, 0, False))
    t.append(("layers.9.", 1, 0, False))
    return t


class _Layer:
    def __init__(self, state, p, stride, pad, relu):
        g = lambda k: state[p + k]
        mk = lambda a, s, z: torch._make_per_tensor_quantized_tensor(torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.int8))), float(s), int(z))
        self.weight = mk(g("weight"), g("weight.q_scale"), g("weight.q_zero_point"))
        self.std = mk(g("std"), g("std.q_scale"), g("std.q_zero_point"))
        b = state.get(p + "bias_", None)
        self.bias = None if b is None or np.asarray(b).size == 0 else torch.from_numpy(np.asarray(b, np.float32).copy())
        self.s_m, self.z_m = float(g("mul_noise.scale")), int(g("mul_noise.zero_point"))
        self.s_a, self.z_a = float(g("add_weight.scale")), int(g("add_weight.zero_point"))
        self.scale, self.zero_point = float(g("scale")), int(g("zero_point"))
        self.stride, self.pad, self.relu = stride, pad, relu


def _clamp_q(x, lo, hi):
    """torch.clamp on a quantised tensor with float bounds (q_bound - z) * s: an integer clamp of the stored values."""
    z, s = x.q_zero_point(), x.q_scale()
    return torch.clamp(x, (lo - z) * s, (hi - z) * s)


class FbgemmResNetBBB:
    def __init__(self, state, a_bits=7, w_bits=8):
        torch.backends.quantized.engine = "fbgemm"
        self.state = state
        self.a_lo, self.a_hi = UINT_BOUNDS[a_bits]
        self.w_lo, self.w_hi = INT_BOUNDS[w_bits]
        self.layers = {p: _Layer(state, p, st, pd, rl) for p, st, pd, rl in _layer_table()}
        self.s_in = float(np.asarray(state["quant.scale"]).reshape(-1)[0])
        self.z_in = int(np.asarray(state["quant.zero_point"]).reshape(-1)[0])

    def _sample(self, L, eps):
        if eps is None:
            eps = torch.empty(L.std.shape, dtype=torch.float32).normal_(0, 1)
        eps_q = torch.quantize_per_tensor(eps, NOISE_SCALE, NOISE_ZERO_POINT, torch.qint8)
        t = torch.ops.quantized.mul(L.std, eps_q, L.s_m, L.z_m)
        w = torch.ops.quantized.add(L.weight, t, L.s_a, L.z_a)
        return _clamp_q(w, self.w_lo, self.w_hi)

    def _run(self, p, x, eps):
        L = self.layers[p]
        e = None
        if eps is not None:                       # oracle eps are OHWI numpy arrays; torch weights are OIHW
            e = np.asarray(eps[p], np.float32)
            shape = (e.shape[0], e.shape[3], e.shape[1], e.shape[2]) if e.ndim == 4 else e.shape
            # through a flat copy: canonical strides (a [48,24,1,1] view of the transposed array has ambiguous strides, which
            # ATen's quantized::mul reads as a different memory format from `std`'s and then walks out of bounds)
            flat = np.array(e.transpose(0, 3, 1, 2) if e.ndim == 4 else e, order="C", copy=True).reshape(-1)
            e = torch.from_numpy(flat).view(shape)
        w = self._sample(L, e)
        if w.dim() == 2:
            packed = torch.ops.quantized.linear_prepack(w, L.bias)
            y = (torch.ops.quantized.linear_relu if L.relu else torch.ops.quantized.linear)(x, packed, L.scale, L.zero_point)
        else:
            packed = torch.ops.quantized.conv2d_prepack(w, L.bias, [L.stride] * 2, [L.pad] * 2, [1, 1], 1)
            y = (torch.ops.quantized.conv2d_relu if L.relu else torch.ops.quantized.conv2d)(x, packed, L.scale, L.zero_point)
        return _clamp_q(y, self.a_lo, self.a_hi)

    @torch.no_grad()
    def forward(self, x_nchw, eps=None, record=None):
        """One stochastic forward: fp32 NCHW numpy / tensor -> softmax probabilities [B, 10] (numpy).
        record: optional dict filled with the NHWC integer representation of every block's output (and its two addends)."""
        nhwc = lambda t: t.int_repr().permute(0, 2, 3, 1).contiguous().numpy()
        x = torch.as_tensor(x_nchw, dtype=torch.float32)
        c = lambda t: _clamp_q(t, self.a_lo, self.a_hi)
        x = c(torch.quantize_per_tensor(x, self.s_in, self.z_in, torch.quint8))
        x = self._run("layers.0.", x, eps)
        for li in (3, 4, 5, 6):
            for bi in (0, 1):
                p = f"layers.{li}.{bi}."
                o = self._run(p + "stem.0.", x, eps)
                o = self._run(p + "stem.3.", o, eps)
                sc = self._run(p + "shortcut.0.", x, eps) if (p + "shortcut.0.") in self.layers else x
                sa, za = float(self.state[p + "add.add.scale"]), int(self.state[p + "add.add.zero_point"])
                added = c(torch.ops.quantized.add(o, sc, sa, za))
                x = c(torch.relu(added))
                if record is not None:
                    record[p + "stem.3.out"], record[p + "res"], record[p + "add"], record[p + "out"] = nhwc(o), nhwc(sc), nhwc(added), nhwc(x)
        x = c(torch.nn.functional.avg_pool2d(x, 4))
        x = c(x.reshape(x.size(0), -1))
        x = self._run("layers.9.", x, eps)
        return torch.softmax(x.dequantize(), dim=-1).numpy()
