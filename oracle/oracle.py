"""ctypes front-end of the CPU oracle (oracle/qbnn_oracle.c) plus the model-level
restatement of the reference's int8 forward graph.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by quantised_bayesian_nets_amd/.

Model graph follows /root/reference/src/models/stochastic/bbb/models_bbb.py:
  ConvNetwork_ResNet.forward :226-245, BasicBlock.forward :170-183,
and the MC loop of /root/reference/experiments/utils.py:342-355.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NOISE_SCALE = float(0.02362204724)   # reference: bbb/quantized/__init__.py:1
NOISE_ZERO_POINT = 0                 # :2
UINT_BOUNDS = {8: (0, 255), 7: (0, 127), 6: (0, 63), 5: (0, 31), 4: (0, 15), 3: (0, 7), 2: (0, 3)}   # src/utils.py:18
INT_BOUNDS = {8: (-128, 127), 7: (-64, 63), 6: (-32, 31), 5: (-16, 15), 4: (-8, 7), 3: (-4, 3), 2: (-2, 1)}  # :19-20


class SampleParams(C.Structure):
    _fields_ = [("inv_noise_scale", C.c_float), ("mul_multiplier", C.c_float),
                ("z_sigma", C.c_int32), ("z_mul", C.c_int32),
                ("s_w", C.c_float), ("nzs_w", C.c_float),
                ("s_mul", C.c_float), ("nzs_mul", C.c_float),
                ("inv_s_add", C.c_float), ("z_add", C.c_int32),
                ("w_lo", C.c_int32), ("w_hi", C.c_int32)]


class ConvParams(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
                ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("z_x", C.c_int32), ("z_w", C.c_int32),
                ("s_x", C.c_float), ("s_w", C.c_float), ("s_y", C.c_float),
                ("z_y", C.c_int32), ("relu", C.c_int32), ("a_hi", C.c_int32)]


def build(force=False):
    so = os.path.join(_HERE, "libqbnn_oracle.so")
    srcs = [os.path.join(_HERE, "qbnn_oracle.c"), os.path.join(_HERE, "qbnn_eps_table.h")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.qbo_num_threads.restype = C.c_int
    return _LIB


def _p(a, t=None):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def f32(x):
    return np.float32(x)


# ---------------------------------------------------------------- RNG ------
def philox(ctr, key):
    out = np.zeros(4, np.uint32)
    lib().qbo_philox4x32_10(_p(np.asarray(ctr, np.uint32)), _p(np.asarray(key, np.uint32)), _p(out))
    return out


def fill_normal(n, seed, layer, sample):
    eps = np.empty(n, np.float32)
    lib().qbo_fill_normal(_p(eps), C.c_int64(n), C.c_uint64(seed), C.c_uint32(layer), C.c_uint32(sample))
    return eps


def fill_eps_q(n, seed, layer, sample):
    """The int8 mode's weight noise: eps_q in [-128, 127] drawn directly from Philox words through the alias table of
    clamp(rne(N(0,1) / s_n)) (qbnn_oracle.c: qbo_fill_eps_q)."""
    e = np.empty(n, np.int8)
    lib().qbo_fill_eps_q(_p(e), C.c_int64(n), C.c_uint64(seed), C.c_uint32(layer), C.c_uint32(sample))
    return e


def eps_from_eps_q(eps_q):
    """The fp32 eps to inject into the reference so that ITS quantize_per_tensor(eps, s_n, 0, qint8) returns eps_q."""
    return eps_q.astype(np.float32) * f32(NOISE_SCALE)


def fill_eps_i8(n, seed, layer, sample):
    """fp32 eps of the int8 noise stream (what the fixture generators inject into the reference's normal_)."""
    return eps_from_eps_q(fill_eps_q(n, seed, layer, sample))


def fill_uniform(n, seed, layer, sample):
    u = np.empty(n, np.float32)
    lib().qbo_fill_uniform(_p(u), C.c_int64(n), C.c_uint64(seed), C.c_uint32(layer), C.c_uint32(sample))
    return u


# ------------------------------------------------------------ int8 ops -----
def sample_params(s_w, z_w, s_sigma, z_sigma, s_mul, z_mul, s_add, z_add, w_bits):
    p = SampleParams()
    p.inv_noise_scale = f32(1.0) / f32(NOISE_SCALE)
    p.mul_multiplier = f32(np.float64(s_sigma) * np.float64(NOISE_SCALE) / np.float64(s_mul))
    p.z_sigma = int(z_sigma)
    p.z_mul = int(z_mul)
    p.s_w = f32(s_w)
    p.nzs_w = f32(-int(z_w)) * f32(s_w)
    p.s_mul = f32(s_mul)
    p.nzs_mul = f32(-int(z_mul)) * f32(s_mul)
    p.inv_s_add = f32(1.0) / f32(s_add)
    p.z_add = int(z_add)
    p.w_lo, p.w_hi = INT_BOUNDS[w_bits]
    return p


def quantize_eps(eps):
    eps = np.ascontiguousarray(eps, np.float32)
    out = np.empty(eps.shape, np.int8)
    lib().qbo_quantize_eps(_p(eps), C.c_int64(eps.size), C.c_float(f32(1.0) / f32(NOISE_SCALE)), _p(out))
    return out


def sample_weights_i8(mu_q, sigma_q, eps, p, want_t=False):
    mu_q = np.ascontiguousarray(mu_q, np.int8)
    sigma_q = np.ascontiguousarray(sigma_q, np.int8)
    eps = np.ascontiguousarray(eps, np.float32)
    w = np.empty(mu_q.shape, np.int8)
    t = np.empty(mu_q.shape, np.int8) if want_t else None
    lib().qbo_sample_weights_i8(_p(mu_q), _p(sigma_q), _p(eps), C.c_int64(mu_q.size), C.byref(p), _p(t), _p(w))
    return (t, w) if want_t else w


def sample_weights_i8_philox(mu_q, sigma_q, p, seed, layer, sample):
    mu_q = np.ascontiguousarray(mu_q, np.int8)
    sigma_q = np.ascontiguousarray(sigma_q, np.int8)
    w = np.empty(mu_q.shape, np.int8)
    lib().qbo_sample_weights_i8_philox(_p(mu_q), _p(sigma_q), C.c_int64(mu_q.size), C.byref(p),
                                       C.c_uint64(seed), C.c_uint32(layer), C.c_uint32(sample), _p(w))
    return w


def conv2d_i8(x, w_ohwi, bias, stride, pad, s_x, z_x, s_w, z_w, s_y, z_y, relu, a_hi):
    """x [B,H,W,Cin] uint8; w [Cout,KH,KW,Cin] int8; returns [B,Ho,Wo,Cout] uint8."""
    x = np.ascontiguousarray(x, np.uint8)
    w = np.ascontiguousarray(w_ohwi, np.int8)
    B, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    p = ConvParams(B, H, W, Cin, Cout, KH, KW, stride, pad, int(z_x), int(z_w),
                   f32(s_x), f32(s_w), f32(s_y), int(z_y), int(bool(relu)), int(a_hi))
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    y = np.empty((B, Ho, Wo, Cout), np.uint8)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    lib().qbo_conv2d_i8(_p(x), _p(w), _p(b), C.byref(p), _p(y))
    return y


def linear_i8(x, w, bias, s_x, z_x, s_w, z_w, s_y, z_y, relu, a_hi):
    B, K = x.shape
    N = w.shape[0]
    y = conv2d_i8(x.reshape(B, 1, 1, K), w.reshape(N, 1, 1, K), bias, 1, 0, s_x, z_x, s_w, z_w, s_y, z_y, relu, a_hi)
    return y.reshape(B, N)


def qadd_relu(a, s_a, z_a, b, s_b, z_b, s_o, z_o, relu, a_hi):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    out = np.empty(a.shape, np.uint8)
    lib().qbo_qadd_relu(_p(a), C.c_float(f32(s_a)), C.c_int32(int(z_a)), _p(b), C.c_float(f32(s_b)), C.c_int32(int(z_b)),
                        C.c_float(f32(s_o)), C.c_int32(int(z_o)), C.c_int32(int(bool(relu))), C.c_int32(int(a_hi)),
                        C.c_int64(a.size), _p(out))
    return out


def quantize_input_nchw(x, s, z, a_hi):
    x = np.ascontiguousarray(x, np.float32)
    B, Cc, H, W = x.shape
    out = np.empty((B, H, W, Cc), np.uint8)
    lib().qbo_quantize_input_nchw(_p(x), B, Cc, H, W, C.c_float(f32(s)), C.c_int32(int(z)), C.c_int32(int(a_hi)), _p(out))
    return out


def avgpool_q(x, k, z, a_hi):
    x = np.ascontiguousarray(x, np.uint8)
    B, H, W, Cc = x.shape
    out = np.empty((B, H // k, W // k, Cc), np.uint8)
    lib().qbo_avgpool_q(_p(x), B, H, W, Cc, k, C.c_int32(int(z)), C.c_int32(int(a_hi)), _p(out))
    return out


def maxpool2_q(x):
    x = np.ascontiguousarray(x, np.uint8)
    B, H, W, Cc = x.shape
    out = np.empty((B, H // 2, W // 2, Cc), np.uint8)
    lib().qbo_maxpool2_q(_p(x), B, H, W, Cc, _p(out))
    return out


def dequant_softmax(q, s, z):
    q = np.ascontiguousarray(q, np.uint8)
    B, Cc = q.shape
    out = np.empty((B, Cc), np.float32)
    lib().qbo_dequant_softmax(_p(q), B, Cc, C.c_float(f32(s)), C.c_int32(int(z)), _p(out))
    return out


def dropout_q(x, mask, z_x, s_x, s_m, z_m, a_hi):
    """x [B,HW,C] or [B,C] uint8; mask [B,C] float 0/1.  Returns ints; the output qparams are (s_m / (1-p), z_m)."""
    x = np.ascontiguousarray(x, np.uint8)
    shp = x.shape
    B, Cc = shp[0], shp[-1]
    HW = int(np.prod(shp[1:-1])) if x.ndim > 2 else 1
    mask = np.ascontiguousarray(mask, np.float32)
    mult = f32(np.float64(s_x) * np.float64(s_m) / np.float64(s_m))
    out = np.empty(shp, np.uint8)
    lib().qbo_dropout_q(_p(x), C.c_int64(B), C.c_int64(HW), C.c_int64(Cc), _p(mask), C.c_int32(int(z_x)), C.c_float(f32(s_m)),
                        C.c_int32(int(z_m)), C.c_float(mult), C.c_int32(int(a_hi)), _p(out))
    return out


# ------------------------------------------------------------- fp32 ops ----
def softplus(rho):
    rho = np.ascontiguousarray(rho, np.float32)
    out = np.empty(rho.shape, np.float32)
    lib().qbo_softplus(_p(rho), C.c_int64(rho.size), _p(out))
    return out


def sample_weights_f32(mu, sigma, eps):
    mu = np.ascontiguousarray(mu, np.float32)
    sigma = np.ascontiguousarray(sigma, np.float32)
    eps = np.ascontiguousarray(eps, np.float32)
    w = np.empty(mu.shape, np.float32)
    lib().qbo_sample_weights_f32(_p(mu), _p(sigma), _p(eps), C.c_int64(mu.size), _p(w))
    return w


def conv2d_f32(x, w_ohwi, bias, stride, pad, relu=False):
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w_ohwi, np.float32)
    B, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    y = np.empty((B, Ho, Wo, Cout), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    lib().qbo_conv2d_f32(_p(x), _p(w), _p(b), B, H, W, Cin, Cout, KH, KW, stride, pad, int(bool(relu)), _p(y))
    return y


# ----------------------------------------------------- model-level graph ---
def oihw_to_ohwi(w):
    return np.ascontiguousarray(np.transpose(w, (0, 2, 3, 1))) if w.ndim == 4 else np.ascontiguousarray(w)


def ohwi_to_oihw(w):
    return np.ascontiguousarray(np.transpose(w, (0, 3, 1, 2))) if w.ndim == 4 else np.ascontiguousarray(w)


def resnet_layer_table():
    """Execution (= eps draw) order of the 21 stochastic layers of conv_resnet_bbb
    (models_bbb.py:171-178 runs the stem before the shortcut).
    Returns list of (prefix, stride, pad, relu)."""
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


class Int8Layer:
    """One converted int8 BBB layer in the reference's state-dict vocabulary
    (conv_q.py:72-78 / linear_q.py:40-46)."""

    def __init__(self, state, prefix, layer_id, stride, pad, relu, w_bits):
        g = lambda k: state[prefix + k]
        self.prefix, self.layer_id, self.stride, self.pad, self.relu = prefix, layer_id, stride, pad, relu
        self.mu_q = oihw_to_ohwi(np.asarray(g("weight"), np.int8))
        self.sigma_q = oihw_to_ohwi(np.asarray(g("std"), np.int8))
        self.s_w, self.z_w = float(g("weight.q_scale")), int(g("weight.q_zero_point"))
        self.s_sigma, self.z_sigma = float(g("std.q_scale")), int(g("std.q_zero_point"))
        self.s_mul, self.z_mul = float(g("mul_noise.scale")), int(g("mul_noise.zero_point"))
        self.s_add, self.z_add = float(g("add_weight.scale")), int(g("add_weight.zero_point"))
        self.s_y, self.z_y = float(g("scale")), int(g("zero_point"))
        b = state.get(prefix + "bias_", None)
        self.bias = None if b is None or np.asarray(b).size == 0 else np.asarray(b, np.float32)
        self.sp = sample_params(self.s_w, self.z_w, self.s_sigma, self.z_sigma, self.s_mul, self.z_mul,
                                self.s_add, self.z_add, w_bits)

    def sample(self, seed, sample, eps=None):
        if eps is None:
            return sample_weights_i8_philox(self.mu_q, self.sigma_q, self.sp, seed, self.layer_id, sample)
        return sample_weights_i8(self.mu_q, self.sigma_q, eps, self.sp)

    def forward(self, x, s_x, z_x, w_q, a_hi):
        if w_q.ndim == 2:
            return linear_i8(x, w_q, self.bias, s_x, z_x, self.s_add, self.z_add, self.s_y, self.z_y, self.relu, a_hi)
        return conv2d_i8(x, w_q, self.bias, self.stride, self.pad, s_x, z_x, self.s_add, self.z_add,
                         self.s_y, self.z_y, self.relu, a_hi)


class Int8ResNetOracle:
    """conv_resnet_bbb after prepare_model -> convert (quant_utils.py:62-147), int8 path."""

    def __init__(self, state, a_bits=7, w_bits=8):
        self.state = state
        self.a_hi = UINT_BOUNDS[a_bits][1]
        self.w_bits = w_bits
        self.table = resnet_layer_table()
        self.layers = {pfx: Int8Layer(state, pfx, i, st, pd, rl, w_bits) for i, (pfx, st, pd, rl) in enumerate(self.table)}
        self.s_in, self.z_in = float(np.asarray(state["quant.scale"]).reshape(-1)[0]), int(np.asarray(state["quant.zero_point"]).reshape(-1)[0])

    def n_weights(self):
        return sum(l.mu_q.size for l in self.layers.values())

    def forward(self, x_nchw, seed, sample, eps=None, record=None):
        """One stochastic forward (one MC sample).  eps: optional dict prefix -> fp32 OHWI eps (parity mode)."""
        st, a_hi = self.state, self.a_hi

        def run(pfx, x, s_x, z_x):
            L = self.layers[pfx]
            w_q = L.sample(seed, sample, None if eps is None else eps[pfx])
            y = L.forward(x, s_x, z_x, w_q, a_hi)
            if record is not None:
                record[pfx + "w_q"] = w_q
                record[pfx + "out"] = y
            return y, L.s_y, L.z_y

        x = quantize_input_nchw(x_nchw, self.s_in, self.z_in, a_hi)
        if record is not None:
            record["quant.out"] = x
        x, s, z = run("layers.0.", x, self.s_in, self.z_in)
        for li in (3, 4, 5, 6):
            for bi in (0, 1):
                p = f"layers.{li}.{bi}."
                o, so, zo = run(p + "stem.0.", x, s, z)
                o, so, zo = run(p + "stem.3.", o, so, zo)
                if (p + "shortcut.0.") in self.layers:
                    sc, ss, zs = run(p + "shortcut.0.", x, s, z)
                else:
                    sc, ss, zs = x, s, z
                sa, za = float(st[p + "add.add.scale"]), int(st[p + "add.add.zero_point"])
                x = qadd_relu(o, so, zo, sc, ss, zs, sa, za, True, a_hi)
                s, z = sa, za
                if record is not None:
                    record[p + "out"] = x
        x = avgpool_q(x, 4, z, a_hi)
        if record is not None:
            record["avgpool.out"] = x
        x = x.reshape(x.shape[0], -1)
        x, s, z = run("layers.9.", x, s, z)
        return dequant_softmax(x, s, z)

    def mc_predict(self, x_nchw, samples, seed, sample_offset=0, eps_fn=None):
        """experiments/utils.py:342-355 (classification branch): mean over S of softmax probs.
        Returns (mean, per_sample_probs)."""
        ps = []
        for s in range(samples):
            e = None if eps_fn is None else eps_fn(sample_offset + s)
            ps.append(self.forward(x_nchw, seed, sample_offset + s, e))
        ps = np.stack(ps, 0)
        return ps.mean(0, dtype=np.float64).astype(np.float32), ps


# ------------------------------------------------- MC-Dropout LeNet (config 2) ---
class Int8LeNetMCOracle:
    """conv_lenet_mc after prepare_model -> convert: reference mcdropout/models_mc.py:75-111 (graph), dropout.py (masks).
    State keys (flat): layers.{0,3}.weight (+ .q_scale/.q_zero_point, OIHW), layers.{7,10}.weight ([out,in], `in` in the
    reference's NCHW flatten order), layers.N.scale/zero_point, layers.{1,4,9}.mul_mask.scale/zero_point, layers.N.p,
    quant.scale/zero_point."""
    DROPS = (1, 4, 9)

    def __init__(self, state, a_bits=7):
        self.st = state
        self.a_hi = UINT_BOUNDS[a_bits][1]
        g = lambda k: state[k]
        self.s_in, self.z_in = float(np.asarray(g("quant.scale")).reshape(-1)[0]), int(np.asarray(g("quant.zero_point")).reshape(-1)[0])
        self.w = {}
        for i in (0, 3):
            self.w[i] = oihw_to_ohwi(np.asarray(g(f"layers.{i}.weight"), np.int8))
        w7 = np.asarray(g("layers.7.weight"), np.int8)            # [500, 50*7*7] in (c,h,w) order -> (h,w,c)
        self.w[7] = np.ascontiguousarray(w7.reshape(500, 50, 7, 7).transpose(0, 2, 3, 1).reshape(500, 2450))
        self.w[10] = np.asarray(g("layers.10.weight"), np.int8)

    def qp(self, i):
        return float(self.st[f"layers.{i}.scale"]), int(self.st[f"layers.{i}.zero_point"])

    def wqp(self, i):
        return float(self.st[f"layers.{i}.weight.q_scale"]), int(self.st[f"layers.{i}.weight.q_zero_point"])

    def mask(self, di, shape, seed, sample, masks=None):
        if masks is not None:
            return masks[di]
        p = np.float32(np.asarray(self.st[f"layers.{self.DROPS[di]}.p"]).reshape(-1)[0])
        keep = np.float32(1.0) - p
        n = int(np.prod(shape))
        return (fill_uniform(n, seed, di, sample) < keep).astype(np.float32).reshape(shape)

    def drop(self, di, x, s, z, seed, sample, masks):
        li = self.DROPS[di]
        s_m, z_m = float(self.st[f"layers.{li}.mul_mask.scale"]), int(self.st[f"layers.{li}.mul_mask.zero_point"])
        mult = float(np.float32(np.asarray(self.st[f"layers.{li}.multiplier"]).reshape(-1)[0]))
        m = self.mask(di, (x.shape[0], x.shape[-1]), seed, sample, masks)
        y = dropout_q(x, m, z, s, s_m, z_m, self.a_hi)
        return y, s_m * mult, z_m        # mul_scalar: scale (double) * 1/(1-p)

    def forward(self, x_nchw, seed, sample, masks=None, record=None):
        a_hi = self.a_hi
        x = quantize_input_nchw(x_nchw, self.s_in, self.z_in, a_hi)
        s, z = self.s_in, self.z_in
        rec = (lambda k, v: record.__setitem__(k, v)) if record is not None else (lambda k, v: None)
        rec("quant.out", x)
        sw, zw = self.wqp(0); sy, zy = self.qp(0)
        x = conv2d_i8(x, self.w[0], None, 1, 2, s, z, sw, zw, sy, zy, False, a_hi); s, z = sy, zy; rec("layers.0.out", x)
        x, s, z = self.drop(0, x, s, z, seed, sample, masks); rec("layers.1.out", x)
        x = np.minimum(maxpool2_q(x), a_hi); rec("layers.2.out", x)
        sw, zw = self.wqp(3); sy, zy = self.qp(3)
        x = conv2d_i8(x, self.w[3], None, 1, 2, s, z, sw, zw, sy, zy, False, a_hi); s, z = sy, zy; rec("layers.3.out", x)
        x, s, z = self.drop(1, x, s, z, seed, sample, masks); rec("layers.4.out", x)
        x = np.minimum(maxpool2_q(x), a_hi); rec("layers.5.out", x)
        x = x.reshape(x.shape[0], -1)                                            # NHWC flatten; weights were permuted to match
        sw, zw = self.wqp(7); sy, zy = self.qp(7)
        x = linear_i8(x, self.w[7], None, s, z, sw, zw, sy, zy, True, a_hi); s, z = sy, zy; rec("layers.7.out", x)
        x, s, z = self.drop(2, x, s, z, seed, sample, masks); rec("layers.9.out", x)
        sw, zw = self.wqp(10); sy, zy = self.qp(10)
        x = linear_i8(x, self.w[10], None, s, z, sw, zw, sy, zy, False, a_hi); s, z = sy, zy; rec("layers.10.out", x)
        return dequant_softmax(x, s, z)

    def mc_predict(self, x_nchw, samples, seed, sample_offset=0):
        ps = np.stack([self.forward(x_nchw, seed, sample_offset + s) for s in range(samples)], 0)
        return ps.mean(0, dtype=np.float64).astype(np.float32), ps


# ------------------------------------------------- float BBB MLP (config 0) ---
class F32MLPOracle:
    """`linear_bbb` float, eval branch: reference bbb/linear.py:42-50 per layer (sigma = softplus(rho), W = mu + eps*sigma,
    y = x @ W^T + b), graph models_bbb.py:61-78: 3 x (Linear(100) + ReLU), heads mu / log_var -> (mu, exp(log_var)).
    fp32 with fp64 accumulation; tolerance 1e-5 relative against the reference."""
    NAMES = ("layers.0", "layers.2", "layers.4", "mu", "log_var")

    def __init__(self, state):
        self.st = state

    def layer(self, name, lid, x, seed, sample, relu):
        mu, rho, b = (np.asarray(self.st[name + k], np.float32) for k in (".weight", ".std", ".bias"))
        eps = fill_normal(mu.size, seed, lid, sample).reshape(mu.shape)
        w = sample_weights_f32(mu, softplus(rho), eps)
        y = (x.astype(np.float64) @ w.astype(np.float64).T + b.astype(np.float64)).astype(np.float32)
        return np.maximum(y, 0) if relu else y

    def forward(self, x, seed, sample):
        h = np.asarray(x, np.float32)
        for lid, n in enumerate(self.NAMES[:3]):
            h = self.layer(n, lid, h, seed, sample, True)
        mu = self.layer("mu", 3, h, seed, sample, False)
        lv = self.layer("log_var", 4, h, seed, sample, False)
        return mu, np.exp(lv)

    def mc_predict(self, x, samples, seed):
        """experiments/utils.py:348-353: (mean_s mu, var_unbiased_s(mu) + mean_s var)."""
        mus, vs = zip(*[self.forward(x, seed, s) for s in range(samples)])
        mus, vs = np.stack(mus).astype(np.float64), np.stack(vs).astype(np.float64)
        return mus.mean(0).astype(np.float32), (mus.var(0, ddof=1) + vs.mean(0)).astype(np.float32)


# ------------------------------------- deterministic int8 ResNet member (config 3) ---
class Int8ResNetDetOracle:
    """One ensemble member of `conv_resnet_sgld` after convert: reference sgld/models_sgld.py:109-212 (graph identical to
    the BBB ResNet, standard torch.nn.quantized Conv2d / ConvReLU2d / Linear, no weight noise); softmax is applied by the
    wrapper (models_sgld.py:286-287)."""

    def __init__(self, state, a_bits=7):
        self.st, self.a_hi = state, UINT_BOUNDS[a_bits][1]
        self.table = resnet_layer_table()

    def conv(self, pfx, x, s_x, z_x, stride, pad, relu):
        g = lambda k: self.st[pfx + k]
        w = oihw_to_ohwi(np.asarray(g("weight"), np.int8))
        b = self.st.get(pfx + "bias", None)
        sy, zy = float(g("scale")), int(g("zero_point"))
        if w.ndim == 2:
            y = linear_i8(x, w, b, s_x, z_x, float(g("weight.q_scale")), int(g("weight.q_zero_point")), sy, zy, relu, self.a_hi)
        else:
            y = conv2d_i8(x, w, b, stride, pad, s_x, z_x, float(g("weight.q_scale")), int(g("weight.q_zero_point")), sy, zy, relu, self.a_hi)
        return y, sy, zy

    def forward(self, x_nchw, record=None):
        st, a_hi = self.st, self.a_hi
        rec = (lambda k, v: record.__setitem__(k, v)) if record is not None else (lambda k, v: None)
        s, z = float(np.asarray(st["quant.scale"]).reshape(-1)[0]), int(np.asarray(st["quant.zero_point"]).reshape(-1)[0])
        x = quantize_input_nchw(x_nchw, s, z, a_hi)
        x, s, z = self.conv("layers.0.", x, s, z, 1, 1, True); rec("layers.0.out", x)
        for li, fs in ((3, 1), (4, 2), (5, 2), (6, 2)):
            for bi in (0, 1):
                p = f"layers.{li}.{bi}."
                stn = fs if bi == 0 else 1
                o, so, zo = self.conv(p + "stem.0.", x, s, z, stn, 1, True)
                o, so, zo = self.conv(p + "stem.3.", o, so, zo, 1, 1, False)
                if (p + "shortcut.0.weight") in st:
                    sc, ss, zs = self.conv(p + "shortcut.0.", x, s, z, stn, 0, False)
                else:
                    sc, ss, zs = x, s, z
                sa, za = float(st[p + "add.add.scale"]), int(st[p + "add.add.zero_point"])
                x = qadd_relu(o, so, zo, sc, ss, zs, sa, za, True, a_hi)
                s, z = sa, za
                rec(p[:-1] + ".out", x)
        x = avgpool_q(x, 4, z, a_hi).reshape(x.shape[0], -1)
        x, s, z = self.conv("layers.9.", x, s, z, 1, 0, False); rec("layers.9.out", x)
        return dequant_softmax(x, s, z)


# ------------------------------------------------- small int8 BBB graphs (row a6) ---
class _Int8BBBBase:
    def __init__(self, state, a_bits, w_bits, names, relus):
        self.st, self.a_hi, self.w_bits = state, UINT_BOUNDS[a_bits][1], w_bits
        self.L = {n: Int8Layer(state, n + ".", i, 1, 0, r, w_bits) for i, (n, r) in enumerate(zip(names, relus))}
        self.s_in = float(np.asarray(state["quant.scale"]).reshape(-1)[0])
        self.z_in = int(np.asarray(state["quant.zero_point"]).reshape(-1)[0])


class Int8LeNetBBBOracle(_Int8BBBBase):
    """conv_lenet_bbb int8: reference models_bbb.py:98-143 (no ReLU after the convs; layers.5 fused LinearReLU)."""

    def __init__(self, state, a_bits=7, w_bits=8):
        super().__init__(state, a_bits, w_bits, ["layers.0", "layers.2", "layers.5", "layers.7"], [False, False, True, False])
        for n in ("layers.0", "layers.2"):
            self.L[n].stride, self.L[n].pad = 1, 2
        # the reference flattens NCHW (src/utils.py:40-47); NHWC here: permute the 2450 input columns of layers.5
        l5 = self.L["layers.5"]
        l5.mu_q = np.ascontiguousarray(l5.mu_q.reshape(500, 50, 7, 7).transpose(0, 2, 3, 1).reshape(500, 2450))
        l5.sigma_q = np.ascontiguousarray(l5.sigma_q.reshape(500, 50, 7, 7).transpose(0, 2, 3, 1).reshape(500, 2450))
        self.perm5 = np.arange(2450).reshape(50, 7, 7).transpose(1, 2, 0).reshape(-1)     # nhwc position -> reference (c,h,w) index

    def sample(self, n, seed, sample):
        L = self.L[n]
        eps = fill_eps_i8(L.mu_q.size, seed, L.layer_id, sample)       # stream defined on the reference-order OHWI tensor
        if n == "layers.5":
            eps = eps.reshape(500, 2450)[:, self.perm5]
        return L.sample(seed, sample, eps.reshape(L.mu_q.shape))

    def forward(self, x_nchw, seed, sample, record=None):
        a_hi = self.a_hi
        rec = (lambda k, v: record.__setitem__(k, v)) if record is not None else (lambda k, v: None)
        x = quantize_input_nchw(x_nchw, self.s_in, self.z_in, a_hi); s, z = self.s_in, self.z_in; rec("quant.out", x)
        L = self.L["layers.0"]; x = L.forward(x, s, z, self.sample("layers.0", seed, sample), a_hi); s, z = L.s_y, L.z_y; rec("layers.0.out", x)
        x = np.minimum(maxpool2_q(x), a_hi); rec("layers.1.out", x)
        L = self.L["layers.2"]; x = L.forward(x, s, z, self.sample("layers.2", seed, sample), a_hi); s, z = L.s_y, L.z_y; rec("layers.2.out", x)
        x = np.minimum(maxpool2_q(x), a_hi); rec("layers.3.out", x)
        x = x.reshape(x.shape[0], -1)
        L = self.L["layers.5"]; x = L.forward(x, s, z, self.sample("layers.5", seed, sample), a_hi); s, z = L.s_y, L.z_y; rec("layers.5.out", x)
        L = self.L["layers.7"]; x = L.forward(x, s, z, self.sample("layers.7", seed, sample), a_hi); s, z = L.s_y, L.z_y; rec("layers.7.out", x)
        return dequant_softmax(x, s, z)


class Int8MLPBBBOracle(_Int8BBBBase):
    """linear_bbb with q=True: reference models_bbb.py:32-95 (3 x LinearReLU, heads mu / log_var, DeQuant, exp)."""

    def __init__(self, state, a_bits=7, w_bits=8):
        super().__init__(state, a_bits, w_bits, ["layers.0", "layers.2", "layers.4", "mu", "log_var"], [True, True, True, False, False])

    def forward(self, x, seed, sample, record=None):
        a_hi = self.a_hi
        rec = (lambda k, v: record.__setitem__(k, v)) if record is not None else (lambda k, v: None)
        x = np.asarray(x, np.float32)
        q = quantize_input_nchw(x.reshape(x.shape[0], x.shape[1], 1, 1), self.s_in, self.z_in, a_hi).reshape(x.shape)
        s, z = self.s_in, self.z_in; rec("quant.out", q)
        for n in ("layers.0", "layers.2", "layers.4"):
            L = self.L[n]; q = L.forward(q, s, z, L.sample(seed, sample), a_hi); s, z = L.s_y, L.z_y; rec(n + ".out", q)
        Lm, Lv = self.L["mu"], self.L["log_var"]
        qm = Lm.forward(q, s, z, Lm.sample(seed, sample), a_hi); rec("mu.out", qm)
        qv = Lv.forward(q, s, z, Lv.sample(seed, sample), a_hi); rec("log_var.out", qv)
        mu = (qm.astype(np.float32) - np.float32(Lm.z_y)) * np.float32(Lm.s_y)
        lv = (qv.astype(np.float32) - np.float32(Lv.z_y)) * np.float32(Lv.s_y)
        return mu, np.exp(lv)


# ------------------------------------------- fp32 convolutional BBB graphs (row a1) ---
def _bn_eval(x, st, name, eps=1e-5):
    """nn.BatchNorm2d eval as ATen evaluates it: alpha = weight / sqrt(var + eps), beta = bias - mean * alpha,
    y = x * alpha + beta (fp32, two roundings).  x NHWC."""
    g, b, rm, rv = (np.asarray(st[f"{name}.{k}"], np.float32) for k in ("weight", "bias", "running_mean", "running_var"))
    invstd = (np.float32(1.0) / np.sqrt(rv + np.float32(eps))).astype(np.float32)
    alpha = (g * invstd).astype(np.float32)
    beta = (b - (rm * alpha).astype(np.float32)).astype(np.float32)
    return ((x * alpha).astype(np.float32) + beta).astype(np.float32)


def _pool_f32(x, k, avg):
    B, H, W, C = x.shape
    v = x.reshape(B, H // k, k, W // k, k, C)
    if not avg:
        return v.max(axis=(2, 4))
    acc = np.zeros((B, H // k, W // k, C), np.float32)
    for dh in range(k):                     # same summation order as the kernel / ATen's avg_pool2d
        for dw in range(k):
            acc = (acc + v[:, :, dh, :, dw, :]).astype(np.float32)
    return (acc / np.float32(k * k)).astype(np.float32)


def _softmax_f32(z):
    z = z.astype(np.float32)
    e = np.exp(z - z.max(-1, keepdims=True)).astype(np.float32)
    return (e / e.sum(-1, keepdims=True, dtype=np.float32)).astype(np.float32)


class F32ConvOracle:
    """Float `conv_lenet_bbb` / `conv_resnet_bbb`, eval branch: reference bbb/conv.py:33-39, bbb/linear.py:42-50 per
    layer, graphs models_bbb.py:100-133 and :191-245.  fp32 with fp64 accumulation inside conv / matmul."""

    def __init__(self, state):
        self.st = state

    def conv(self, name, lid, x, seed, sample, stride, pad):
        mu, rho = (np.asarray(self.st[name + k], np.float32) for k in (".weight", ".std"))
        eps = fill_normal(mu.size, seed, lid, sample).reshape(mu.shape)
        w = sample_weights_f32(mu, softplus(rho), eps)                        # [Cout, Cin, KH, KW]
        return conv2d_f32(x, np.ascontiguousarray(w.transpose(0, 2, 3, 1)), None, stride, pad)

    def linear(self, name, lid, x, seed, sample, relu):
        mu, rho = (np.asarray(self.st[name + k], np.float32) for k in (".weight", ".std"))
        eps = fill_normal(mu.size, seed, lid, sample).reshape(mu.shape)
        w = sample_weights_f32(mu, softplus(rho), eps)
        y = (x.astype(np.float64) @ w.astype(np.float64).T).astype(np.float32)
        return np.maximum(y, 0) if relu else y

    def lenet(self, x_nchw, seed, sample):
        h = np.ascontiguousarray(np.asarray(x_nchw, np.float32).transpose(0, 2, 3, 1))
        h = _pool_f32(self.conv("layers.0", 0, h, seed, sample, 1, 2), 2, False)
        h = _pool_f32(self.conv("layers.2", 1, h, seed, sample, 1, 2), 2, False)
        h = np.ascontiguousarray(h.transpose(0, 3, 1, 2)).reshape(h.shape[0], -1)
        h = self.linear("layers.5", 2, h, seed, sample, True)
        return _softmax_f32(self.linear("layers.7", 3, h, seed, sample, False))

    def resnet(self, x_nchw, seed, sample):
        h = np.ascontiguousarray(np.asarray(x_nchw, np.float32).transpose(0, 2, 3, 1))
        lid = 0
        h = np.maximum(_bn_eval(self.conv("layers.0", lid, h, seed, sample, 1, 1), self.st, "layers.1"), 0); lid += 1
        inp = 24
        for li, planes, stride in ((3, 24, 1), (4, 48, 2), (5, 96, 2), (6, 192, 2)):
            for bi, st in enumerate((stride, 1)):
                p = f"layers.{li}.{bi}"
                out = np.maximum(_bn_eval(self.conv(p + ".stem.0", lid, h, seed, sample, st, 1), self.st, p + ".stem.1"), 0); lid += 1
                out = _bn_eval(self.conv(p + ".stem.3", lid, out, seed, sample, 1, 1), self.st, p + ".stem.4"); lid += 1
                sc = h
                if st != 1 or inp != planes:
                    sc = _bn_eval(self.conv(p + ".shortcut.0", lid, h, seed, sample, st, 0), self.st, p + ".shortcut.1"); lid += 1
                h = np.maximum((out + sc).astype(np.float32), 0)
                inp = planes
        h = _pool_f32(h, 4, True).reshape(h.shape[0], -1)
        return _softmax_f32(self.linear("layers.9", lid, h, seed, sample, False))


# --------------------------------- QAT fake-quant evaluation with live observers (row a2) ---
class EmaObserver:
    """MovingAverageMinMaxObserver (averaging_constant 0.01) + calculate_qparams, per-tensor affine, fp32 arithmetic
    (torch/ao/quantization/observer.py); `state` = (min, max) or None before the first batch."""

    def __init__(self, qmin, qmax, state=None, c=0.01):
        self.qmin, self.qmax, self.c = int(qmin), int(qmax), np.float32(c)
        self.state = None if state is None else (np.float32(state[0]), np.float32(state[1]))

    def update(self, x):
        mn, mx = np.float32(x.min()), np.float32(x.max())
        if self.state is None:
            self.state = (mn, mx)
        else:
            a, b = self.state
            self.state = (np.float32(a + self.c * np.float32(mn - a)), np.float32(b + self.c * np.float32(mx - b)))
        lo, hi = min(self.state[0], np.float32(0)), max(self.state[1], np.float32(0))
        scale = max(np.float32(np.float32(hi - lo) / np.float32(self.qmax - self.qmin)), np.float32(1.1920928955078125e-07))
        zp = int(np.clip(self.qmin - np.rint(np.float32(lo / scale)), self.qmin, self.qmax))
        return np.float32(scale), zp

    def fake_quant(self, x):
        """FakeQuantize.forward: observer update, then fake_quantize_per_tensor_affine with the fresh qparams."""
        s, z = self.update(x)
        inv = np.float32(1.0) / s
        q = np.clip(np.rint((np.asarray(x, np.float32) * inv).astype(np.float32)) + np.float32(z), self.qmin, self.qmax)
        return ((q - np.float32(z)).astype(np.float32) * s).astype(np.float32)


class QATOracle:
    """Prepared (QAT) BBB models in eval mode with live observers: reference conv_qat.py:26-49,139-167,228-251,
    linear_qat.py:18-41,78-79 on the graphs of models_bbb.py.  STATEFUL: every forward advances the observers, so the
    samples must be evaluated in order s = 0, 1, ... exactly like S reference forwards.  `state` uses the reference's
    state_dict names of the prepared model."""

    def __init__(self, state, a_bits=7, w_bits=8):
        self.st = state
        self.ab, self.wb = UINT_BOUNDS[a_bits], INT_BOUNDS[w_bits]
        self.obs = {}

    def fq(self, prefix, x, weight_like):
        if prefix not in self.obs:
            mn = float(np.asarray(self.st[prefix + ".activation_post_process.min_val"]))
            mx = float(np.asarray(self.st[prefix + ".activation_post_process.max_val"]))
            lo, hi = self.wb if weight_like else self.ab
            self.obs[prefix] = EmaObserver(lo, hi, (mn, mx) if np.isfinite(mn) and np.isfinite(mx) else None)
        return self.obs[prefix].fake_quant(np.asarray(x, np.float32))

    def weights(self, name, lid, seed, sample, c=None):
        mu, rho = (np.asarray(self.st[name + k], np.float32) for k in (".weight", ".std"))
        sg = softplus(rho)
        if c is not None:
            shape = [-1] + [1] * (mu.ndim - 1)
            mu, sg = (mu * c.reshape(shape)).astype(np.float32), (sg * c.reshape(shape)).astype(np.float32)
        w = self.fq(name + ".weight_fake_quant", mu, True)
        s = self.fq(name + ".std_fake_quant", sg, True)
        eps = fill_normal(mu.size, seed, lid, sample).reshape(mu.shape)
        t = self.fq(name + ".mul_noise.activation_post_process", (eps * s).astype(np.float32), True)
        return self.fq(name + ".add_weight.activation_post_process", (w + t).astype(np.float32), True)

    def conv(self, name, lid, x, seed, sample, stride, pad, bn, relu):
        c = None
        if bn:
            g, rv = np.asarray(self.st[name + ".bn.weight"], np.float32), np.asarray(self.st[name + ".bn.running_var"], np.float32)
            c = (g / np.sqrt(rv + np.float32(1e-5))).astype(np.float32)
        W = self.weights(name, lid, seed, sample, c)
        z = conv2d_f32(x, np.ascontiguousarray(W.transpose(0, 2, 3, 1)), None, stride, pad)
        if bn:
            z = (z / c).astype(np.float32)
            z = _bn_eval(z, self.st, name + ".bn")
        if relu:
            z = np.maximum(z, 0)
        return self.fq(name + ".activation_post_process", z, False)

    def linear(self, name, lid, x, seed, sample, relu, bias):
        W = self.weights(name, lid, seed, sample)
        y = (x.astype(np.float64) @ W.astype(np.float64).T).astype(np.float32)
        if bias:
            y = (y + np.asarray(self.st[name + ".bias"], np.float32)).astype(np.float32)
        if relu:
            y = np.maximum(y, 0)
        return self.fq(name + ".activation_post_process", y, False)

    def lenet(self, x_nchw, seed, sample):
        h = np.ascontiguousarray(np.asarray(x_nchw, np.float32).transpose(0, 2, 3, 1))
        h = self.fq("quant.activation_post_process", h, False)
        h = _pool_f32(self.conv("layers.0", 0, h, seed, sample, 1, 2, False, False), 2, False)
        h = _pool_f32(self.conv("layers.2", 1, h, seed, sample, 1, 2, False, False), 2, False)
        h = np.ascontiguousarray(h.transpose(0, 3, 1, 2)).reshape(h.shape[0], -1)
        h = self.linear("layers.5", 2, h, seed, sample, True, False)
        return _softmax_f32(self.linear("layers.7", 3, h, seed, sample, False, False))

    def mlp(self, x, seed, sample):
        h = self.fq("quant.activation_post_process", np.asarray(x, np.float32), False)
        for lid, n in enumerate(("layers.0", "layers.2", "layers.4")):
            h = self.linear(n, lid, h, seed, sample, True, True)
        mu = self.linear("mu", 3, h, seed, sample, False, True)
        lv = self.linear("log_var", 4, h, seed, sample, False, True)
        return mu, np.exp(lv)

    def resnet(self, x_nchw, seed, sample):
        h = np.ascontiguousarray(np.asarray(x_nchw, np.float32).transpose(0, 2, 3, 1))
        h = self.fq("quant.activation_post_process", h, False)
        lid = 0
        h = self.conv("layers.0", lid, h, seed, sample, 1, 1, True, True); lid += 1
        inp = 24
        for li, planes, stride in ((3, 24, 1), (4, 48, 2), (5, 96, 2), (6, 192, 2)):
            for bi, st in enumerate((stride, 1)):
                p = f"layers.{li}.{bi}"
                out = self.conv(p + ".stem.0", lid, h, seed, sample, st, 1, True, True); lid += 1
                out = self.conv(p + ".stem.3", lid, out, seed, sample, 1, 1, True, False); lid += 1
                sc = h
                if st != 1 or inp != planes:
                    sc = self.conv(p + ".shortcut.0", lid, h, seed, sample, st, 0, True, False); lid += 1
                h = np.maximum(self.fq(p + ".add.add.activation_post_process", (out + sc).astype(np.float32), False), 0)
                inp = planes
        h = _pool_f32(h, 4, True).reshape(h.shape[0], -1)
        return _softmax_f32(self.linear("layers.9", lid, h, seed, sample, False, False))


# --------------------------------------------- QAT evaluation of the non-BBB graphs (quant_utils.py:139-140) ---
class QATMCOracle(QATOracle):
    """`linear_mc`, `conv_lenet_mc`, `conv_resnet_mc` and the SGHMC member templates (`*_sgld`: `main_net.` = the pointwise graphs of
    models_sgld.py, no dropout) after quant_utils.prepare_model's `prepare_qat` branch (:139-140), in eval mode with live observers:
    torch.ao.nn.qat Linear / Conv2d and intrinsic.qat LinearReLU / ConvBn2d / ConvBnReLU2d -- W = weight_fake_quant(weight * c), c = the
    BatchNorm fold gamma / sqrt(var + eps) (conv-bn only), Z = conv(X, W), Z / c (+ bias), bn, (ReLU), activation FakeQuantize -- and
    mcdropout/dropout.py:15-40 with the FloatFunctionals prepared: y = FQ_mul_mask(x * mask) * multiplier (mul_scalar is not observed).
    Add = FQ(out + shortcut) (src/utils.py:49-55).  STATEFUL like QATOracle: samples in order s = 0, 1, ..."""

    def __init__(self, state, a_bits=7, w_bits=8, prefix=""):
        super().__init__(state, a_bits, w_bits)
        self.pre = prefix

    def det_weights(self, name, c=None):
        w = np.asarray(self.st[name + ".weight"], np.float32)
        if c is not None:
            w = (w * c.reshape([-1] + [1] * (w.ndim - 1))).astype(np.float32)
        return self.fq(name + ".weight_fake_quant", w, True)

    def conv(self, name, x, stride, pad, bn, relu):
        name = self.pre + name
        c = None
        if bn:
            g, rv = np.asarray(self.st[name + ".bn.weight"], np.float32), np.asarray(self.st[name + ".bn.running_var"], np.float32)
            c = (g / np.sqrt(rv + np.float32(1e-5))).astype(np.float32)
        W = self.det_weights(name, c)
        z = conv2d_f32(x, np.ascontiguousarray(W.transpose(0, 2, 3, 1)), None, stride, pad)
        if bn:
            z = _bn_eval((z / c).astype(np.float32), self.st, name + ".bn")
        if relu:
            z = np.maximum(z, 0)
        return self.fq(name + ".activation_post_process", z, False)

    def linear(self, name, x, relu):
        name = self.pre + name
        W = self.det_weights(name)
        y = (x.astype(np.float64) @ W.astype(np.float64).T).astype(np.float32)
        if (name + ".bias") in self.st:
            y = (y + np.asarray(self.st[name + ".bias"], np.float32)).astype(np.float32)
        if relu:
            y = np.maximum(y, 0)
        return self.fq(name + ".activation_post_process", y, False)

    def drop(self, name, di, x, seed, sample):
        name = self.pre + name
        if (name + ".p") not in self.st:
            return x
        p = np.float32(np.asarray(self.st[name + ".p"]).reshape(-1)[0])
        if p <= 0:
            return x
        mult = np.float32(np.asarray(self.st[name + ".multiplier"]).reshape(-1)[0])
        B, C = x.shape[0], x.shape[-1]
        m = (fill_uniform(B * C, seed, di, sample) < (np.float32(1.0) - p)).astype(np.float32).reshape((B,) + (1,) * (x.ndim - 2) + (C,))
        y = self.fq(name + ".mul_mask.activation_post_process", (x * m).astype(np.float32), False)
        return (y * mult).astype(np.float32)

    def quant(self, h):
        return self.fq(self.pre + "quant.activation_post_process", h, False)

    def mlp_mc(self, x, seed, sample):
        h = self.quant(np.asarray(x, np.float32))
        h = self.drop("layers.2", 0, self.linear("layers.0", h, True), seed, sample)
        h = self.drop("layers.5", 1, self.linear("layers.3", h, True), seed, sample)
        h = self.linear("layers.6", h, True)
        mu = self.linear("mu.1", self.drop("mu.0", 2, h, seed, sample), False)
        lv = self.linear("log_var.1", self.drop("log_var.0", 3, h, seed, sample), False)
        return mu, np.exp(lv)

    def lenet_mc(self, x_nchw, seed, sample):
        h = self.quant(np.ascontiguousarray(np.asarray(x_nchw, np.float32).transpose(0, 2, 3, 1)))
        h = _pool_f32(self.drop("layers.1", 0, self.conv("layers.0", h, 1, 2, False, False), seed, sample), 2, False)
        h = _pool_f32(self.drop("layers.4", 1, self.conv("layers.3", h, 1, 2, False, False), seed, sample), 2, False)
        h = np.ascontiguousarray(h.transpose(0, 3, 1, 2)).reshape(h.shape[0], -1)
        h = self.drop("layers.9", 2, self.linear("layers.7", h, True), seed, sample)
        return _softmax_f32(self.linear("layers.10", h, False))

    def resnet_mc(self, x_nchw, seed, sample):
        """mcdropout/models_mc.py:116-226 prepared: layers.0 ConvBnReLU2d, layers.3 dropout, blocks layers.4 .. 7 (stem.0 ConvBnReLU2d, stem.3 dropout,
        stem.4 ConvBn2d, stem.6 dropout, shortcut.0 ConvBn2d, shortcut.2 dropout, add), AvgPool, Flatten, layers.10 Linear."""
        h = self.quant(np.ascontiguousarray(np.asarray(x_nchw, np.float32).transpose(0, 2, 3, 1)))
        di = 0
        h = self.drop("layers.3", di, self.conv("layers.0", h, 1, 1, True, True), seed, sample); di += 1
        inp = 24
        for li, planes, stride in ((4, 24, 1), (5, 48, 2), (6, 96, 2), (7, 192, 2)):
            for bi, st in enumerate((stride, 1)):
                p = f"layers.{li}.{bi}"
                out = self.drop(p + ".stem.3", di, self.conv(p + ".stem.0", h, st, 1, True, True), seed, sample); di += 1
                out = self.drop(p + ".stem.6", di, self.conv(p + ".stem.4", out, 1, 1, True, False), seed, sample); di += 1
                sc = h
                if st != 1 or inp != planes:
                    sc = self.drop(p + ".shortcut.2", di, self.conv(p + ".shortcut.0", h, st, 0, True, False), seed, sample); di += 1
                h = np.maximum(self.fq(self.pre + p + ".add.add.activation_post_process", (out + sc).astype(np.float32), False), 0)
                inp = planes
        h = _pool_f32(h, 4, True).reshape(h.shape[0], -1)
        return _softmax_f32(self.linear("layers.10", h, False))

    # the SGHMC member templates (models_sgld.py: the pointwise graphs; Network.forward applies the softmax, :285-287)
    def mlp_p(self, x):
        h = self.quant(np.asarray(x, np.float32))
        for n in ("layers.0", "layers.2", "layers.4"):
            h = self.linear(n, h, True)
        return self.linear("mu", h, False), np.exp(self.linear("log_var", h, False))

    def lenet_p(self, x_nchw):
        h = self.quant(np.ascontiguousarray(np.asarray(x_nchw, np.float32).transpose(0, 2, 3, 1)))
        h = _pool_f32(self.conv("layers.0", h, 1, 2, False, False), 2, False)
        h = _pool_f32(self.conv("layers.2", h, 1, 2, False, False), 2, False)
        h = np.ascontiguousarray(h.transpose(0, 3, 1, 2)).reshape(h.shape[0], -1)
        return _softmax_f32(self.linear("layers.7", self.linear("layers.5", h, True), False))

    def resnet_p(self, x_nchw):
        h = self.quant(np.ascontiguousarray(np.asarray(x_nchw, np.float32).transpose(0, 2, 3, 1)))
        h = self.conv("layers.0", h, 1, 1, True, True)
        inp = 24
        for li, planes, stride in ((3, 24, 1), (4, 48, 2), (5, 96, 2), (6, 192, 2)):
            for bi, st in enumerate((stride, 1)):
                p = f"layers.{li}.{bi}"
                out = self.conv(p + ".stem.3", self.conv(p + ".stem.0", h, st, 1, True, True), 1, 1, True, False)
                sc = self.conv(p + ".shortcut.0", h, st, 0, True, False) if (st != 1 or inp != planes) else h
                h = np.maximum(self.fq(self.pre + p + ".add.add.activation_post_process", (out + sc).astype(np.float32), False), 0)
                inp = planes
        h = _pool_f32(h, 4, True).reshape(h.shape[0], -1)
        return _softmax_f32(self.linear("layers.9", h, False))


# --------------------------------------------- MC-Dropout ResNet, int8 ---
class Int8ResNetMCOracle:
    """`conv_resnet_mc` after prepare_model -> convert: reference mcdropout/models_mc.py:116-211 (graph: a channel dropout
    after every conv), dropout.py:15-40 (quantised masks).  Mask draw order = execution order: layers.3, then per block
    stem.3, stem.6, shortcut.2."""

    def __init__(self, state, a_bits=7):
        self.st, self.a_hi = state, UINT_BOUNDS[a_bits][1]

    def conv(self, name, x, s, z, stride, pad, relu):
        w = oihw_to_ohwi(np.asarray(self.st[name + ".weight"], np.int8))
        sw, zw = float(self.st[name + ".weight.q_scale"]), int(self.st[name + ".weight.q_zero_point"])
        sy, zy = float(self.st[name + ".scale"]), int(self.st[name + ".zero_point"])
        b = self.st.get(name + ".bias", None)
        return conv2d_i8(x, w, None if b is None else np.asarray(b, np.float32), stride, pad, s, z, sw, zw, sy, zy, relu, self.a_hi), sy, zy

    def drop(self, name, di, x, s, z, seed, sample, masks):
        s_m, z_m = float(self.st[name + ".mul_mask.scale"]), int(self.st[name + ".mul_mask.zero_point"])
        mult = float(np.float32(np.asarray(self.st[name + ".multiplier"]).reshape(-1)[0]))
        if masks is not None:
            m = masks[di]
        else:
            keep = np.float32(1.0) - np.float32(np.asarray(self.st[name + ".p"]).reshape(-1)[0])
            m = (fill_uniform(x.shape[0] * x.shape[-1], seed, di, sample) < keep).astype(np.float32).reshape(x.shape[0], x.shape[-1])
        return dropout_q(x, m, z, s, s_m, z_m, self.a_hi), s_m * mult, z_m

    def forward(self, x_nchw, seed, sample, masks=None, record=None):
        st = self.st
        s, z = float(np.asarray(st["quant.scale"]).reshape(-1)[0]), int(np.asarray(st["quant.zero_point"]).reshape(-1)[0])
        h = quantize_input_nchw(x_nchw, s, z, self.a_hi)
        di = 0
        h, s, z = self.conv("layers.0", h, s, z, 1, 1, True)
        h, s, z = self.drop("layers.3", di, h, s, z, seed, sample, masks); di += 1
        if record is not None:
            record["layers.3.out"] = h
        inp = 24
        for li, planes, stride in ((4, 24, 1), (5, 48, 2), (6, 96, 2), (7, 192, 2)):
            for bi, stv in enumerate((stride, 1)):
                p = f"layers.{li}.{bi}"
                o, so, zo = self.conv(p + ".stem.0", h, s, z, stv, 1, True)
                o, so, zo = self.drop(p + ".stem.3", di, o, so, zo, seed, sample, masks); di += 1
                o, so, zo = self.conv(p + ".stem.4", o, so, zo, 1, 1, False)
                o, so, zo = self.drop(p + ".stem.6", di, o, so, zo, seed, sample, masks); di += 1
                sc, ss, zs = h, s, z
                if stv != 1 or inp != planes:
                    sc, ss, zs = self.conv(p + ".shortcut.0", h, s, z, stv, 0, False)
                    sc, ss, zs = self.drop(p + ".shortcut.2", di, sc, ss, zs, seed, sample, masks); di += 1
                s, z = float(st[p + ".add.add.scale"]), int(st[p + ".add.add.zero_point"])
                h = qadd_relu(o, so, zo, sc, ss, zs, s, z, True, self.a_hi)
                inp = planes
                if record is not None:
                    record[p + ".out"] = h
        h = avgpool_q(h, 4, z, self.a_hi).reshape(h.shape[0], -1)
        w = np.asarray(st["layers.10.weight"], np.int8)
        sw, zw = float(st["layers.10.weight.q_scale"]), int(st["layers.10.weight.q_zero_point"])
        sy, zy = float(st["layers.10.scale"]), int(st["layers.10.zero_point"])
        b = st.get("layers.10.bias", None)
        logits = linear_i8(h, w, None if b is None else np.asarray(b, np.float32), s, z, sw, zw, sy, zy, False, self.a_hi)
        return dequant_softmax(logits, sy, zy)


# --------------------------------------------- MC-Dropout MLP, int8 ---
class Int8MLPMCOracle:
    """`linear_mc` after prepare_model -> convert: reference mcdropout/models_mc.py:10-73 (3 x LinearReLU(100) with a
    per-element BernoulliDropout between them and one in front of each of the heads `mu` and `log_var`), dropout.py:15-40
    (quantised masks, one draw per element of the 2-D activation).  Mask draw order = execution order: layers.2, layers.5,
    mu.0, log_var.0.  State: torch.nn.quantized Linear(ReLU) as `<name>.weight` (+ .q_scale / .q_zero_point), `.bias`,
    `.scale`, `.zero_point`; dropouts as `<name>.mul_mask.scale / .zero_point`, `.p`, `.multiplier`."""
    DROPS = ("layers.2", "layers.5", "mu.0", "log_var.0")

    def __init__(self, state, a_bits=7):
        self.st, self.a_hi = state, UINT_BOUNDS[a_bits][1]

    def linear(self, name, x, s, z, relu):
        g = lambda k: self.st[name + k]
        w = np.asarray(g(".weight"), np.int8)
        b = self.st.get(name + ".bias", None)
        sy, zy = float(g(".scale")), int(g(".zero_point"))
        y = linear_i8(x, w, None if b is None or np.asarray(b).size == 0 else np.asarray(b, np.float32), s, z,
                      float(g(".weight.q_scale")), int(g(".weight.q_zero_point")), sy, zy, relu, self.a_hi)
        return y, sy, zy

    def drop(self, di, x, s, z, seed, sample, masks):
        name = self.DROPS[di]
        s_m, z_m = float(self.st[name + ".mul_mask.scale"]), int(self.st[name + ".mul_mask.zero_point"])
        mult = float(np.float32(np.asarray(self.st[name + ".multiplier"]).reshape(-1)[0]))
        if masks is not None:
            m = masks[di]
        else:
            keep = np.float32(1.0) - np.float32(np.asarray(self.st[name + ".p"]).reshape(-1)[0])
            m = (fill_uniform(x.size, seed, di, sample) < keep).astype(np.float32).reshape(x.shape)
        return dropout_q(x, m, z, s, s_m, z_m, self.a_hi), s_m * mult, z_m

    def forward(self, x, seed, sample, masks=None, record=None):
        rec = (lambda k, v: record.__setitem__(k, v)) if record is not None else (lambda k, v: None)
        x = np.asarray(x, np.float32)
        s, z = float(np.asarray(self.st["quant.scale"]).reshape(-1)[0]), int(np.asarray(self.st["quant.zero_point"]).reshape(-1)[0])
        h = quantize_input_nchw(x.reshape(x.shape[0], x.shape[1], 1, 1), s, z, self.a_hi).reshape(x.shape); rec("quant.out", h)
        h, s, z = self.linear("layers.0", h, s, z, True); rec("layers.0.out", h)
        h, s, z = self.drop(0, h, s, z, seed, sample, masks); rec("layers.2.out", h)
        h, s, z = self.linear("layers.3", h, s, z, True); rec("layers.3.out", h)
        h, s, z = self.drop(1, h, s, z, seed, sample, masks); rec("layers.5.out", h)
        h, s, z = self.linear("layers.6", h, s, z, True); rec("layers.6.out", h)
        hm, sm, zm = self.drop(2, h, s, z, seed, sample, masks); rec("mu.0.out", hm)
        qm, sm, zm = self.linear("mu.1", hm, sm, zm, False); rec("mu.1.out", qm)
        hv, sv, zv = self.drop(3, h, s, z, seed, sample, masks); rec("log_var.0.out", hv)
        qv, sv, zv = self.linear("log_var.1", hv, sv, zv, False); rec("log_var.1.out", qv)
        mu = (qm.astype(np.float32) - np.float32(zm)) * np.float32(sm)
        lv = (qv.astype(np.float32) - np.float32(zv)) * np.float32(sv)
        return mu, np.exp(lv)

    def mc_predict(self, x, samples, seed):
        """experiments/utils.py:348-353: (mean_s mu, var_unbiased_s(mu) + mean_s var)."""
        mus, vs = zip(*[self.forward(x, seed, s) for s in range(samples)])
        mus, vs = np.stack(mus).astype(np.float64), np.stack(vs).astype(np.float64)
        return mus.mean(0).astype(np.float32), (mus.var(0, ddof=1) + vs.mean(0)).astype(np.float32)


# ------------------------------------------- float MC-Dropout graphs (rows a6 / a7, q=False) ---
class F32MCOracle:
    """`linear_mc`, `conv_lenet_mc`, `conv_resnet_mc` with q=False in eval mode: reference mcdropout/models_mc.py:10-226 (graphs) and
    dropout.py:15-40 with FloatFunctional (`(x * mask) * multiplier`, two fp32 roundings; 4-D inputs drop whole channels).  Deterministic
    nn.Linear / nn.Conv2d / nn.BatchNorm2d (eval) weights; masks from the Philox uniform stream (seed, dropout index in execution
    order, sample) or injected.  fp32 with fp64 accumulation inside conv / matmul."""

    def __init__(self, state):
        self.st = state

    def drop(self, name, di, x, seed, sample, masks):
        p = np.float32(np.asarray(self.st[name + ".p"]).reshape(-1)[0])
        mult = np.float32(np.asarray(self.st[name + ".multiplier"]).reshape(-1)[0])
        B, C = x.shape[0], x.shape[-1]
        if masks is not None:
            m = np.asarray(masks[di], np.float32).reshape(B, C)
        else:
            m = (fill_uniform(B * C, seed, di, sample) < (np.float32(1.0) - p)).astype(np.float32).reshape(B, C)
        m = m.reshape((B,) + (1,) * (x.ndim - 2) + (C,))
        return ((x * m).astype(np.float32) * mult).astype(np.float32)

    def linear(self, name, x, relu=False):
        w = np.asarray(self.st[name + ".weight"], np.float32)
        y = x.astype(np.float64) @ w.astype(np.float64).T
        if (name + ".bias") in self.st:
            y = y + np.asarray(self.st[name + ".bias"], np.float32).astype(np.float64)
        y = y.astype(np.float32)
        return np.maximum(y, 0) if relu else y

    def conv(self, name, x, stride, pad):
        w = np.asarray(self.st[name + ".weight"], np.float32)
        return conv2d_f32(x, np.ascontiguousarray(w.transpose(0, 2, 3, 1)), None, stride, pad)

    def mlp(self, x, seed, sample, masks=None):
        h = np.asarray(x, np.float32)
        h = self.drop("layers.2", 0, self.linear("layers.0", h, True), seed, sample, masks)
        h = self.drop("layers.5", 1, self.linear("layers.3", h, True), seed, sample, masks)
        h = self.linear("layers.6", h, True)
        mu = self.linear("mu.1", self.drop("mu.0", 2, h, seed, sample, masks))
        lv = self.linear("log_var.1", self.drop("log_var.0", 3, h, seed, sample, masks))
        return mu, np.exp(lv)

    def lenet(self, x_nchw, seed, sample, masks=None):
        h = np.ascontiguousarray(np.asarray(x_nchw, np.float32).transpose(0, 2, 3, 1))
        h = _pool_f32(self.drop("layers.1", 0, self.conv("layers.0", h, 1, 2), seed, sample, masks), 2, False)
        h = _pool_f32(self.drop("layers.4", 1, self.conv("layers.3", h, 1, 2), seed, sample, masks), 2, False)
        h = np.ascontiguousarray(h.transpose(0, 3, 1, 2)).reshape(h.shape[0], -1)
        h = self.drop("layers.9", 2, self.linear("layers.7", h, True), seed, sample, masks)
        return _softmax_f32(self.linear("layers.10", h))

    def resnet(self, x_nchw, seed, sample, masks=None):
        h = np.ascontiguousarray(np.asarray(x_nchw, np.float32).transpose(0, 2, 3, 1))
        di = 0
        h = np.maximum(_bn_eval(self.conv("layers.0", h, 1, 1), self.st, "layers.1"), 0)
        h = self.drop("layers.3", di, h, seed, sample, masks); di += 1
        inp = 24
        for li, planes, stride in ((4, 24, 1), (5, 48, 2), (6, 96, 2), (7, 192, 2)):
            for bi, st in enumerate((stride, 1)):
                p = f"layers.{li}.{bi}"
                out = np.maximum(_bn_eval(self.conv(p + ".stem.0", h, st, 1), self.st, p + ".stem.1"), 0)
                out = self.drop(p + ".stem.3", di, out, seed, sample, masks); di += 1
                out = _bn_eval(self.conv(p + ".stem.4", out, 1, 1), self.st, p + ".stem.5")
                out = self.drop(p + ".stem.6", di, out, seed, sample, masks); di += 1
                sc = h
                if st != 1 or inp != planes:
                    sc = _bn_eval(self.conv(p + ".shortcut.0", h, st, 0), self.st, p + ".shortcut.1")
                    sc = self.drop(p + ".shortcut.2", di, sc, seed, sample, masks); di += 1
                h = np.maximum((out + sc).astype(np.float32), 0)
                inp = planes
        h = _pool_f32(h, 4, True).reshape(h.shape[0], -1)
        return _softmax_f32(self.linear("layers.10", h))
