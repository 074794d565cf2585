/*
 * qbnn_oracle.c -- CPU restatement (ORACLE) of the Monte-Carlo int8 / fp32
 * Bayesian-NN inference path of martinferianc/quantised-bayesian-nets.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke()
 * function of __graft_entry__.py and bench.py's cpu_baseline leg may load it.
 * The product path (quantised_bayesian_nets_amd/) never links or calls it.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit (int8)
 * or to 1e-5 relative (fp32) against outputs of the reference itself, imported
 * from /root/reference in the build container by tests/golden/make_golden.py;
 * the resulting vectors live in tests/golden/ and tests/test_oracle_golden.py
 * replays them.
 * The arithmetic of the quantised ops lives in PyTorch ATen / FBGEMM
 * (third-party, pinned torch==1.7.1 in the reference's requirements.txt:54;
 * torch 2.10.0 here); the formulas were re-derived by adversarial probing of
 * those ops (values placed on rounding ties), see DESIGN.md "Arithmetic
 * contracts".
 *
 * Layouts: activations NHWC uint8 (quint8 integer representation); weights
 * OHWI int8 (channels-last flattening of the reference's OIHW tensor).
 *
 * Compile with -ffp-contract=off: every fused multiply-add below is explicit
 * (fmaf) because bit-exactness depends on where ATen/FBGEMM do and do not fuse.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define QBO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11).  The reference draws eps with        */
/* torch's global generator (conv_q.py:113, linear_q.py:86, conv.py:34);      */
/* the build replaces that stream by this counter-based one.  Parity against  */
/* the reference always goes through injected eps.             */
/* ------------------------------------------------------------------------ */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

QBO_API void qbo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
    uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += PHILOX_W0; k1 += PHILOX_W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Deterministic fp32 log / sincos built only from IEEE add, mul, fma so that  */
/* gcc (x86) and hipcc (gfx950) produce identical bits.  Polynomials: Cephes.  */
static inline float qbo_logf(float x) { /* x in (0,1], normal */
  union { float f; uint32_t u; } v = { x };
  int e = (int)((v.u >> 23) & 0xffu) - 126;           /* x = m * 2^e, m in [0.5,1) */
  v.u = (v.u & 0x007fffffu) | 0x3f000000u;
  float m = v.f;
  if (m < 0.70710678118654752440f) { e -= 1; m = m + m; }
  float t = m - 1.0f;
  float z = t * t;
  float y = 7.0376836292E-2f;
  y = fmaf(y, t, -1.1514610310E-1f);
  y = fmaf(y, t, 1.1676998740E-1f);
  y = fmaf(y, t, -1.2420140846E-1f);
  y = fmaf(y, t, 1.4249322787E-1f);
  y = fmaf(y, t, -1.6668057665E-1f);
  y = fmaf(y, t, 2.0000714765E-1f);
  y = fmaf(y, t, -2.4999993993E-1f);
  y = fmaf(y, t, 3.3333331174E-1f);
  y = y * t * z;
  float fe = (float)e;
  y = fmaf(-2.12194440e-4f, fe, y);
  y = fmaf(-0.5f, z, y);
  float r = t + y;
  r = fmaf(0.693359375f, fe, r);
  return r;
}

/* sin and cos of 2*pi*u for u = k * 2^-24, k in [0, 2^24) */
static inline void qbo_sincos2pi(float u, float* s, float* c) {
  float t = u * 4.0f;                 /* exact */
  int q = (int)t;                     /* quadrant 0..3 */
  float f = t - (float)q;             /* exact, [0,1) */
  int flip = f > 0.5f;
  float g = flip ? 1.0f - f : f;      /* exact, [0,0.5] */
  float a = g * 1.57079632679489661923f;
  float z = a * a;
  float ps = -1.9515295891E-4f;
  ps = fmaf(ps, z, 8.3321608736E-3f);
  ps = fmaf(ps, z, -1.6666654611E-1f);
  float sn = fmaf(ps * z, a, a);
  float pc = 2.443315711809948E-5f;
  pc = fmaf(pc, z, -1.388731625493765E-3f);
  pc = fmaf(pc, z, 4.166664568298827E-2f);
  float cs = fmaf(pc, z * z, fmaf(-0.5f, z, 1.0f));
  float s0 = flip ? cs : sn;          /* sin(pi/2 * f) */
  float c0 = flip ? sn : cs;          /* cos(pi/2 * f) */
  switch (q & 3) {
    case 0: *s = s0;  *c = c0;  break;
    case 1: *s = c0;  *c = -s0; break;
    case 2: *s = -s0; *c = -c0; break;
    default: *s = -c0; *c = s0; break;
  }
}

/* Four N(0,1) draws from one Philox block (two Box-Muller pairs). */
QBO_API void qbo_normal4(const uint32_t r[4], float out[4]) {
  for (int p = 0; p < 2; ++p) {
    float u1 = (float)((r[2 * p] >> 8) + 1u) * 5.9604644775390625e-8f;   /* (0,1] */
    float u2 = (float)(r[2 * p + 1] >> 8) * 5.9604644775390625e-8f;      /* [0,1) */
    float rad = sqrtf(-2.0f * qbo_logf(u1));
    float s, c;
    qbo_sincos2pi(u2, &s, &c);
    out[2 * p] = rad * c;
    out[2 * p + 1] = rad * s;
  }
}

/* Stream definition shared with the HIP kernels:
 *   element i of a tensor (OHWI flattening), tensor id `layer`, MC sample
 *   `sample`, stream tag `stream` (0 = weight eps, 1 = dropout mask):
 *   block = philox(ctr = {i >> 2, layer, sample, stream}, key = {seed lo, seed hi});
 *   value = normal4(block)[i & 3]                                            */
QBO_API void qbo_fill_normal(float* eps, int64_t n, uint64_t seed, uint32_t layer, uint32_t sample) {
  uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
  for (int64_t i = 0; i < n; i += 4) {
    uint32_t ctr[4] = { (uint32_t)(i >> 2), layer, sample, 0u }, r[4];
    float v[4];
    qbo_philox4x32_10(ctr, key, r);
    qbo_normal4(r, v);
    for (int j = 0; j < 4 && i + j < n; ++j) eps[i + j] = v[j];
  }
}

/* Uniform [0,1) 24-bit draws for Bernoulli masks (stream tag 1). */
QBO_API void qbo_fill_uniform(float* u, int64_t n, uint64_t seed, uint32_t layer, uint32_t sample) {
  uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
  for (int64_t i = 0; i < n; i += 4) {
    uint32_t ctr[4] = { (uint32_t)(i >> 2), layer, sample, 1u }, r[4];
    qbo_philox4x32_10(ctr, key, r);
    for (int j = 0; j < 4 && i + j < n; ++j) u[i + j] = (float)(r[j] >> 8) * 5.9604644775390625e-8f;
  }
}

/* ------------------------------------------------------------------------ */
/* int8 weight sampling chain: conv_q.py:113-119 / linear_q.py:86-92          */
/* ------------------------------------------------------------------------ */
typedef struct {
  float inv_noise_scale;   /* 1.0f / (float)NOISE_SCALE, quantized/__init__.py:1            */
  float mul_multiplier;    /* (float)((double)s_sigma * (double)NOISE_SCALE / (double)s_mul)  */
  int32_t z_sigma;         /* zero point of std (qint8)                                       */
  int32_t z_mul;           /* mul_noise.zero_point                                            */
  float s_w, nzs_w;        /* weight scale, (float)(-z_w) * s_w                               */
  float s_mul, nzs_mul;    /* mul_noise.scale, (float)(-z_mul) * s_mul                        */
  float inv_s_add;         /* 1.0f / add_weight.scale                                         */
  int32_t z_add;           /* add_weight.zero_point                                           */
  int32_t w_lo, w_hi;      /* INT_BOUNDS[weight_precision], src/utils.py:19-20                */
} qbo_sample_params;

static inline int32_t qbo_clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* rne of an fp32 value that may be far outside int range: clamp first (monotone) */
static inline int32_t qbo_rne_sat(float v) {
  if (!(v > -1.0e9f)) return -1000000000;
  if (v > 1.0e9f) return 1000000000;
  return (int32_t)lrintf(v);
}

/* eps -> eps_q : torch.quantize_per_tensor(noise, NOISE_SCALE, 0, qint8), conv_q.py:115 */
static inline int32_t qbo_quant_eps(float eps, float inv_noise_scale) {
  return qbo_clampi(qbo_rne_sat(eps * inv_noise_scale), -128, 127);
}

/* one weight: quantized::mul (conv_q.py:118 inner), quantized::add (outer), clamp_weight (:119) */
static inline int32_t qbo_sample_one(int32_t mu_q, int32_t sigma_q, int32_t eps_q, const qbo_sample_params* p) {
  int32_t prod = (sigma_q - p->z_sigma) * eps_q;                       /* NOISE_ZERO_POINT = 0 */
  int32_t t_q = qbo_clampi(p->z_mul + qbo_rne_sat((float)prod * p->mul_multiplier), -128, 127);
  float dw = fmaf(p->s_w, (float)mu_q, p->nzs_w);                      /* ATen vectorised dequant: fma */
  float dt = fmaf(p->s_mul, (float)t_q, p->nzs_mul);
  int32_t w_q = qbo_clampi(p->z_add + qbo_rne_sat((dw + dt) * p->inv_s_add), -128, 127);
  return qbo_clampi(w_q, p->w_lo, p->w_hi);
}

QBO_API void qbo_quantize_eps(const float* eps, int64_t n, float inv_noise_scale, int8_t* eps_q) {
  for (int64_t i = 0; i < n; ++i) eps_q[i] = (int8_t)qbo_quant_eps(eps[i], inv_noise_scale);
}

/* stage outputs for layer-level fixtures: t_q = quantized::mul result, w_q = final weight */
QBO_API void qbo_sample_weights_i8(const int8_t* mu_q, const int8_t* sigma_q, const float* eps, int64_t n,
                                   const qbo_sample_params* p, int8_t* t_q_out, int8_t* w_q_out) {
  for (int64_t i = 0; i < n; ++i) {
    int32_t e = qbo_quant_eps(eps[i], p->inv_noise_scale);
    if (t_q_out) {
      int32_t prod = ((int32_t)sigma_q[i] - p->z_sigma) * e;
      t_q_out[i] = (int8_t)qbo_clampi(p->z_mul + qbo_rne_sat((float)prod * p->mul_multiplier), -128, 127);
    }
    w_q_out[i] = (int8_t)qbo_sample_one(mu_q[i], sigma_q[i], e, p);
  }
}

/* The int8 noise stream.  The reference quantises its N(0,1) draw one instruction after drawing it
 * (conv_q.py:113-115, linear_q.py:86-88): eps_q = clamp(rne(eps * (1/s_n)), -128, 127) is a discrete random variable
 * with P(k) = Phi((k + 1/2) s_n') - Phi((k - 1/2) s_n') (tails folded into -128 / 127).  The int8 mode therefore draws
 * eps_q DIRECTLY: one 32-bit Philox word per weight through Walker's alias table of that distribution
 * (qbnn_eps_table.h, generated by tools/make_eps_table.py; |P_table - P| < 2^-31), integer compare only:
 *   block = philox(ctr = {i >> 2, layer, sample, 0}, key = seed);  u = block[i & 3];
 *   c = u >> 24;  eps_q = ((u & 0xffffff) < thr[c] ? c : alias[c]) - 128.
 * Parity with the reference is by injection: eps = (float)eps_q * s_n quantises back to eps_q exactly (checked for all 256
 * values by the fixture generators and by tests/test_oracle_golden.py). */
#include "qbnn_eps_table.h"
static inline int32_t qbo_eps_q_from_u32(uint32_t u) {
  const uint32_t e = QBNN_EPS_ALIAS[u >> 24];
  return (int32_t)(((u & 0xffffffu) < (e >> 8)) ? (u >> 24) : (e & 0xffu)) - 128;
}

QBO_API void qbo_fill_eps_q(int8_t* eps_q, int64_t n, uint64_t seed, uint32_t layer, uint32_t sample) {
  uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
  for (int64_t i = 0; i < n; i += 4) {
    uint32_t ctr[4] = { (uint32_t)(i >> 2), layer, sample, 0u }, r[4];
    qbo_philox4x32_10(ctr, key, r);
    for (int j = 0; j < 4 && i + j < n; ++j) eps_q[i + j] = (int8_t)qbo_eps_q_from_u32(r[j]);
  }
}

/* Philox-driven variant (what the GPU sampler does in-kernel) */
QBO_API void qbo_sample_weights_i8_philox(const int8_t* mu_q, const int8_t* sigma_q, int64_t n,
                                          const qbo_sample_params* p, uint64_t seed, uint32_t layer,
                                          uint32_t sample, int8_t* w_q_out) {
  uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
  for (int64_t i = 0; i < n; i += 4) {
    uint32_t ctr[4] = { (uint32_t)(i >> 2), layer, sample, 0u }, r[4];
    qbo_philox4x32_10(ctr, key, r);
    for (int j = 0; j < 4 && i + j < n; ++j) {
      int32_t e = qbo_eps_q_from_u32(r[j]);
      w_q_out[i + j] = (int8_t)qbo_sample_one(mu_q[i + j], sigma_q[i + j], e, p);
    }
  }
}

/* ------------------------------------------------------------------------ */
/* int8 conv / linear with FBGEMM requantisation:                             */
/*   conv_q.py:120-125 (quantized.functional.conv2d), :206-209 (conv2d_relu)  */
/*   linear_q.py:93-94, :168-172                                              */
/* followed by clamp_activation (src/utils.py:25-30) to [0, 2^A - 1].         */
/* ------------------------------------------------------------------------ */
typedef struct {
  int32_t B, H, W, Cin, Cout, KH, KW, stride, pad;
  int32_t z_x;        /* input zero point                                     */
  int32_t z_w;        /* sampled-weight zero point (= add_weight.zero_point)  */
  float s_x, s_w, s_y;
  int32_t z_y;
  int32_t relu;       /* ConvReLU2d / LinearReLU: lower clamp = z_y           */
  int32_t a_hi;       /* UINT_BOUNDS[activation_precision][1], e.g. 127       */
} qbo_conv_params;

static inline uint8_t qbo_requant(int32_t acc, float bias, int has_bias, float rcp, float mult,
                                  int32_t z_y, int32_t lo, int32_t hi) {
  float xf = (float)acc;
  if (has_bias) xf = fmaf(bias, rcp, xf);           /* FBGEMM float-bias path contracts to fma */
  int32_t q = z_y + qbo_rne_sat(xf * mult);
  q = qbo_clampi(q, lo, 255);
  if (q > hi) q = hi;                               /* clamp_activation */
  return (uint8_t)q;
}

QBO_API void qbo_conv2d_i8(const uint8_t* x, const int8_t* w, const float* bias, const qbo_conv_params* p, uint8_t* y) {
  const int Ho = (p->H + 2 * p->pad - p->KH) / p->stride + 1;
  const int Wo = (p->W + 2 * p->pad - p->KW) / p->stride + 1;
  const float atw = p->s_x * p->s_w;                /* qconv.cpp: float * float */
  const float rcp = 1.0f / atw;
  const float mult = atw / p->s_y;
  const int lo = p->relu ? p->z_y : 0;
  const int Cin = p->Cin, Cout = p->Cout;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < p->B; ++b)
    for (int oh = 0; oh < Ho; ++oh) {
      int32_t* accv = (int32_t*)malloc(sizeof(int32_t) * (size_t)Cout);
      int16_t* xrow = (int16_t*)malloc(sizeof(int16_t) * (size_t)Cin);
      for (int ow = 0; ow < Wo; ++ow) {
        for (int co = 0; co < Cout; ++co) accv[co] = 0;
        for (int kh = 0; kh < p->KH; ++kh) {
          int ih = oh * p->stride - p->pad + kh;
          if (ih < 0 || ih >= p->H) continue;        /* zero padding == z_x, contributes 0 */
          for (int kw = 0; kw < p->KW; ++kw) {
            int iw = ow * p->stride - p->pad + kw;
            if (iw < 0 || iw >= p->W) continue;
            const uint8_t* xp = x + (((int64_t)b * p->H + ih) * p->W + iw) * Cin;
            for (int c = 0; c < Cin; ++c) xrow[c] = (int16_t)((int)xp[c] - p->z_x);
            for (int co = 0; co < Cout; ++co) {
              const int8_t* wp = w + (((int64_t)co * p->KH + kh) * p->KW + kw) * Cin;
              int32_t s = 0;
              for (int c = 0; c < Cin; ++c) s += (int32_t)xrow[c] * ((int32_t)wp[c] - p->z_w);
              accv[co] += s;
            }
          }
        }
        uint8_t* yp = y + (((int64_t)b * Ho + oh) * Wo + ow) * Cout;
        for (int co = 0; co < Cout; ++co)
          yp[co] = qbo_requant(accv[co], bias ? bias[co] : 0.0f, bias != NULL, rcp, mult, p->z_y, lo, p->a_hi);
      }
      free(accv);
      free(xrow);
    }
}

/* linear: x [B,K] uint8, w [N,K] int8 */
QBO_API void qbo_linear_i8(const uint8_t* x, const int8_t* w, const float* bias, const qbo_conv_params* p, uint8_t* y) {
  qbo_conv_params q = *p;
  q.H = q.W = q.KH = q.KW = q.stride = 1; q.pad = 0;
  qbo_conv2d_i8(x, w, bias, &q, y);
}

/* quantized::add (FloatFunctional->QFunctional `Add`, src/utils.py:49-55) then
 * clamp_activation, ReLU on quint8 (= max(q, z)), clamp_activation:
 * models_bbb.py:179-182.  relu=0 gives the bare add+clamp.                    */
QBO_API void qbo_qadd_relu(const uint8_t* a, float s_a, int32_t z_a, const uint8_t* b, float s_b, int32_t z_b,
                           float s_o, int32_t z_o, int32_t relu, int32_t a_hi, int64_t n, uint8_t* out) {
  const float nzs_a = (float)(-z_a) * s_a, nzs_b = (float)(-z_b) * s_b, inv = 1.0f / s_o;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    float da = fmaf(s_a, (float)a[i], nzs_a);
    float db = fmaf(s_b, (float)b[i], nzs_b);
    int32_t q = qbo_clampi(z_o + qbo_rne_sat((da + db) * inv), 0, 255);
    if (q > a_hi) q = a_hi;
    if (relu && q < z_o) q = z_o;
    out[i] = (uint8_t)q;
  }
}

/* QuantStub: torch.quantize_per_tensor(x, s, z, quint8) then clamp_activation; models_bbb.py:227-229.
 * Input NCHW fp32 (as the reference's loaders deliver), output NHWC uint8.    */
QBO_API void qbo_quantize_input_nchw(const float* x, int32_t B, int32_t C, int32_t H, int32_t W,
                                     float s, int32_t z, int32_t a_hi, uint8_t* out) {
  const float inv = 1.0f / s;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w) {
          float v = x[(((int64_t)b * C + c) * H + h) * W + w];
          int32_t q = qbo_clampi(z + qbo_rne_sat(v * inv), 0, 255);
          if (q > a_hi) q = a_hi;
          out[(((int64_t)b * H + h) * W + w) * C + c] = (uint8_t)q;
        }
}

/* nn.AvgPool2d(k) on a channels-last quint8 tensor (the layout the reference's
 * quantised convs emit), models_bbb.py:209: keeps (s,z);
 * q = clamp(rne((sum - k*k*z) * (1/(k*k))) + z, 0, 255), then clamp_activation. */
QBO_API void qbo_avgpool_q(const uint8_t* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                           int32_t z, int32_t a_hi, uint8_t* out) {
  const int Ho = H / k, Wo = W / k;
  const float inv = 1.0f / (float)(k * k);
  for (int b = 0; b < B; ++b)
    for (int oh = 0; oh < Ho; ++oh)
      for (int ow = 0; ow < Wo; ++ow)
        for (int c = 0; c < C; ++c) {
          int32_t s = 0;
          for (int i = 0; i < k; ++i)
            for (int j = 0; j < k; ++j)
              s += x[(((int64_t)b * H + oh * k + i) * W + ow * k + j) * C + c];
          int32_t q = qbo_clampi(qbo_rne_sat((float)(s - k * k * z) * inv) + z, 0, 255);
          if (q > a_hi) q = a_hi;
          out[(((int64_t)b * Ho + oh) * Wo + ow) * C + c] = (uint8_t)q;
        }
}

/* nn.MaxPool2d(2,2) on quint8 (LeNet, models_bbb.py:106): integer max, keeps (s,z). */
QBO_API void qbo_maxpool2_q(const uint8_t* x, int32_t B, int32_t H, int32_t W, int32_t C, uint8_t* out) {
  const int Ho = H / 2, Wo = W / 2;
  for (int b = 0; b < B; ++b)
    for (int oh = 0; oh < Ho; ++oh)
      for (int ow = 0; ow < Wo; ++ow)
        for (int c = 0; c < C; ++c) {
          int m = 0;
          for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) {
              int v = x[(((int64_t)b * H + oh * 2 + i) * W + ow * 2 + j) * C + c];
              if (v > m) m = v;
            }
          out[(((int64_t)b * Ho + oh) * Wo + ow) * C + c] = (uint8_t)m;
        }
}

/* Quantised BernoulliDropout (mcdropout/dropout.py:15-40) on a channels-last tensor [B][HW][C]:
 *   mask (0/1, one value per (b, c): per-channel for 4-D inputs, per-element for 2-D where HW == 1)
 *   mask_q = quantize_per_tensor(mask, s_m, z_m, quint8)                               dropout.py:31-34
 *   y      = mul_mask.mul(x, mask_q)  -> quantized::mul, output qparams (s_m, z_m)      dropout.py:38
 *   y      = mul_scalar.mul_scalar(y, 1/(1-p)) : integers and zero point unchanged, scale *= 1/(1-p)   :39
 * then clamp_activation.  `mult` = (float)((double)s_x * (double)s_m / (double)s_m)  (ATen qmul multiplier). */
QBO_API void qbo_dropout_q(const uint8_t* x, int64_t B, int64_t HW, int64_t C, const float* mask, int32_t z_x,
                           float s_m, int32_t z_m, float mult, int32_t a_hi, uint8_t* out) {
  const float inv_sm = 1.0f / s_m;
  for (int64_t b = 0; b < B; ++b)
    for (int64_t c = 0; c < C; ++c) {
      const int32_t mq = qbo_clampi(z_m + qbo_rne_sat(mask[b * C + c] * inv_sm), 0, 255);
      for (int64_t p = 0; p < HW; ++p) {
        const int64_t i = (b * HW + p) * C + c;
        const int32_t prod = ((int32_t)x[i] - z_x) * (mq - z_m);
        int32_t q = qbo_clampi(z_m + qbo_rne_sat((float)prod * mult), 0, 255);
        if (q > a_hi) q = a_hi;
        out[i] = (uint8_t)q;
      }
    }
}

/* DeQuantStub + F.softmax(dim=-1): models_bbb.py:240-243 */
QBO_API void qbo_dequant_softmax(const uint8_t* q, int32_t B, int32_t C, float s, int32_t z, float* probs) {
  for (int b = 0; b < B; ++b) {
    float v[64], m = -INFINITY, sum = 0.f;
    for (int c = 0; c < C; ++c) { v[c] = (float)((int)q[b * C + c] - z) * s; if (v[c] > m) m = v[c]; }
    for (int c = 0; c < C; ++c) { v[c] = expf(v[c] - m); sum += v[c]; }
    for (int c = 0; c < C; ++c) probs[b * C + c] = v[c] / sum;
  }
}

/* MC reduction, experiments/utils.py:342-355 (classification): running sums of p and p^2. */
QBO_API void qbo_accumulate_moments(const float* probs, int64_t n, float* sum_p, float* sum_p2) {
  for (int64_t i = 0; i < n; ++i) { sum_p[i] += probs[i]; sum_p2[i] += probs[i] * probs[i]; }
}

/* ------------------------------------------------------------------------ */
/* fp32 path: bbb/conv.py:33-39, bbb/linear.py:42-50 (eval branch)            */
/* ------------------------------------------------------------------------ */
QBO_API void qbo_softplus(const float* rho, int64_t n, float* sigma) {
  for (int64_t i = 0; i < n; ++i) {
    float x = rho[i];
    sigma[i] = x > 20.0f ? x : log1pf(expf(x));    /* F.softplus, beta=1, threshold=20 */
  }
}

QBO_API void qbo_sample_weights_f32(const float* mu, const float* sigma, const float* eps, int64_t n, float* w) {
  for (int64_t i = 0; i < n; ++i) w[i] = mu[i] + eps[i] * sigma[i];   /* mul then add: two roundings */
}

/* x [B,H,W,Cin] fp32 NHWC, w [Cout,KH,KW,Cin]; accumulates in double (reference: mkldnn/cuDNN fp32) */
QBO_API void qbo_conv2d_f32(const float* x, const float* w, const float* bias, int32_t B, int32_t H, int32_t W,
                            int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                            int32_t relu, float* y) {
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int oh = 0; oh < Ho; ++oh)
      for (int ow = 0; ow < Wo; ++ow)
        for (int co = 0; co < Cout; ++co) {
          double acc = bias ? (double)bias[co] : 0.0;
          for (int kh = 0; kh < KH; ++kh) {
            int ih = oh * stride - pad + kh;
            if (ih < 0 || ih >= H) continue;
            for (int kw = 0; kw < KW; ++kw) {
              int iw = ow * stride - pad + kw;
              if (iw < 0 || iw >= W) continue;
              const float* xp = x + (((int64_t)b * H + ih) * W + iw) * Cin;
              const float* wp = w + (((int64_t)co * KH + kh) * KW + kw) * Cin;
              for (int c = 0; c < Cin; ++c) acc += (double)xp[c] * (double)wp[c];
            }
          }
          float v = (float)acc;
          if (relu && v < 0.f) v = 0.f;
          y[(((int64_t)b * Ho + oh) * Wo + ow) * Cout + co] = v;
        }
}

QBO_API int qbo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* bench.py's cpu_baseline pins the team to the cores the process may use (a GPU box reports every hardware thread of its host) */
QBO_API void qbo_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
