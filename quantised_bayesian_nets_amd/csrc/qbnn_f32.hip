// libqbnn_hip.so, second translation unit: the fp32 side of the path.
//   a1: float Bayes-by-backprop convolution, eval branch (reference bbb/conv.py:33-39): per-MC-sample weights
//       W_s = mu + eps_s * softplus(rho) (qbnn_sample_weights_f32), Z_s = conv2d(X_s, W_s); BatchNorm (eval), ReLU, Add,
//       pooling, flatten, softmax of the float graphs (models_bbb.py:100-245).
//   a2: QAT fake-quant evaluation with LIVE observers (reference quantized/conv_qat.py:26-49,139-167,
//       linear_qat.py:18-41): MovingAverageMinMax observer update + fake_quantize_per_tensor_affine around every
//       tensor.  The observers' EMA makes sample s depend on samples < s only through two scalars per observer,
//       so all S samples are still evaluated layer by layer in one pass: per-sample min/max (parallel), the EMA
//       recurrence over S scalars on the device (qbnn_observer_scan), fake-quant with per-sample (scale, zero point).
// fp32 accumulation order differs from the reference's mkldnn kernels; the tolerance is stated in the tests.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <stdlib.h>

#include "../../include/qbnn.h"
#include "qbnn_common.h"
#include "qbnn_rng.h"
#include "qbnn_q8.h"

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

// =====================================================================================
// conv2d, fp32, per-sample weights, any geometry: one implicit-GEMM kernel template (conv2d_f32_vec_kernel) on
// v_mfma_f32_32x32x2f32 (fp32 fma chain per output) or v_mfma_f64_16x16x4_f64 (fp64 accumulation), K = KH KW Cin walked in
// 16-wide chunks staged through LDS with the next chunk prefetched into registers; im2col on the fly with zero padding.
// The gather is a float4 per row (VEC) or four scalars with per-element index arithmetic (any Cin, either weight order).
// =====================================================================================
struct ConvF32Args {
  const float* x; int64_t x_ss;      // [S|1][B][H][W][Cin] (NHWC)
  const float* w; int64_t w_ss;      // [S|1][Cout][Cin][KH][KW], or [Cout][KH][KW][Cin] with w_ohwi
  const float* bias;                 // [Cout] or null
  float* y; int64_t y_ss;            // [S][B][Ho][Wo][Cout]
  int B, H, W, Cin, Cout, KS, stride, pad, Ho, Wo, relu;
  int w_ohwi;                        // weights stored [Cout][KH][KW][Cin] (flags bit 2) instead of the reference's [Cout][Cin][KH][KW]
  // fused tail (qbnn_conv2d_f32_fused_mc): v = conv; v = v / div[n]; v = v + bias[n]; v = v * alpha[n]; v = v + beta[n];
  // v = v + res; ReLU -- each step rounded to fp32 exactly as the separate pointwise kernels would
  const float* div; const float* alpha; const float* beta; const float* res; int64_t res_ss;
  float* mm_partials;                // optional [S][workgroups per sample][2]: (min, max) of each workgroup's outputs (QAT observers)
  // float BernoulliDropout behind the conv (mcdropout/dropout.py:15-40 with FloatFunctional: (v * mask) * multiplier), between
  // BatchNorm and the Add: drop_mask [S][B][Cout] of 0 / 1 (qbnn_dropout_mask_f32_mc), one value per (sample, image, channel)
  const float* drop_mask; float drop_mult;
  int tail4;                         // Cout % 4 == 0 and every epilogue operand 16-byte aligned: the fp32 epilogue moves float4s
};

__device__ __forceinline__ float conv_f32_tail(const ConvF32Args& a, float v, int n, int s, int64_t off, const float* drop_row = nullptr) {
  if (a.div) v = v / a.div[n];
  if (a.bias) v = v + a.bias[n];
  if (a.alpha) v = v * a.alpha[n];
  if (a.beta) v = v + a.beta[n];
  if (drop_row) { v = v * drop_row[n]; v = v * a.drop_mult; }     // mul_mask.mul, then mul_scalar.mul_scalar: two roundings
  if (a.res) v = v + a.res[(int64_t)s * a.res_ss + off];
  if (a.relu) v = fmaxf(v, 0.f);
  return v;
}

constexpr int CF_KC = 16;

// VEC (Cin % 4 == 0 and K-contiguous weights): every thread moves one float4 per operand row and
// 16-wide K chunk (thread -> row tid / 4, k = 4 (tid % 4) .. +3 of the chunk; a float4 never straddles a tap) instead of 8
// scalars with per-element index arithmetic (which made the generic form VALU-bound: 21 vs 80-100 TFLOP/s).
// Workgroup tile = PT pixels x NT channels (PT NT = 4096), 4 waves of 32 x 32: 64 x 64, or 128 x 32 where that wastes
// fewer padded channels (Cout = 24: 75 % instead of 37 % useful MFMA work; Cout = 96: 100 % instead of 75 %).
//   ACC64 = false: v_mfma_f32_32x32x2f32, the fp32 fma chain of the float graphs.
//   ACC64 = true : v_mfma_f64_16x16x4_f64 (products of two fp32 values are exact in fp64), 2 x 2 tiles per wave.  A operand =
//                  pixels, B = weights: lane l supplies row / column (l & 15) and holds D[pixel 4 r + (l >> 4)][channel l & 15]
//                  in register r (layout probed on the hardware), so a quarter-wave writes 16 consecutive channels of a pixel.
// The contraction only needs A and B to agree on which k a (step, lane group) pair means: lane half h takes k = 8 h .. 8 h + 7
// of the chunk (fp64: quarter q takes 4 q .. 4 q + 3), so every operand read is a conflict-free ds_read_b128.
constexpr int CF_LD4 = 20;           // row pitch in floats: 16-byte aligned rows for the float4 stores
typedef double v4d __attribute__((ext_vector_type(4)));
//   VEC = 2: the same with the gather's addressing precomputed per row (KS KS <= 32 taps, a sample's input below 2^31 elements) -- see below.
template <int PT, int NT, bool ACC64, int VEC>
__global__ __launch_bounds__(256) void conv2d_f32_vec_kernel(const ConvF32Args a) {
  static_assert(PT * NT == 4096 && PT % 64 == 0 && NT % 32 == 0, "4 waves of 32 x 32");
  __shared__ __attribute__((aligned(16))) float As[NT * CF_LD4];     // weights [n][k]
  __shared__ __attribute__((aligned(16))) float Bs[PT * CF_LD4];     // pixels  [p][k]
  constexpr int WN = NT / 32, XP = PT / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int s = blockIdx.z;
  const int p0 = blockIdx.x * PT, n0 = blockIdx.y * NT;
  const int npix = a.B * a.Ho * a.Wo;
  const int K = a.KS * a.KS * a.Cin;
  const float* xs = a.x + (int64_t)s * a.x_ss;
  const float* ws = a.w + (int64_t)s * a.w_ss;
  const int row = tid >> 2, kq = (tid & 3) * 4;
  int pb[XP], ih0[XP], iw0[XP];
#pragma unroll
  for (int j = 0; j < XP; ++j) {
    const int p = p0 + row + 64 * j;
    pb[j] = -1; ih0[j] = 0; iw0[j] = 0;
    if (p < npix) { pb[j] = p / (a.Ho * a.Wo); const int rem = p - pb[j] * a.Ho * a.Wo; ih0[j] = (rem / a.Wo) * a.stride - a.pad; iw0[j] = (rem % a.Wo) * a.stride - a.pad; }
  }
  const bool wrow_ok = row < NT && n0 + row < a.Cout;
  const float* wrow = ws + (int64_t)(wrow_ok ? n0 + row : 0) * K + kq;
  const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
  // this thread's float4 of the chunk: k = k0 + kq lies in tap (gkh, gkw) at channel gc (Cin % 4 == 0: never straddles a
  // tap); advanced by 16 per chunk without divisions
  int gkh, gkw, gc;
  { const int tap = kq / a.Cin; gc = kq - tap * a.Cin; gkh = tap / a.KS; gkw = tap - gkh * a.KS; }
  // Round 4: the gather's index arithmetic was 40 % of this kernel's vector issue slots (MFMA busy 46 - 53 %, profiles/r04_pmc_util_resnet_f32.json):
  // per row and chunk a chain of 64-bit multiplies and four compares.  Now a row keeps the address of its window's tap (0, 0) and one validity bit
  // per tap (KS KS <= 32; larger kernels keep the compares), and a chunk adds ONE per-thread offset (tap row / column / channel) to it.
  constexpr bool tapmask = VEC == 2;
  int roff[XP];                  // element offset of the row's window tap (0, 0) in the sample's input (may be negative: padding)
  uint32_t rmask[XP];
  if constexpr (tapmask) {
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      roff[j] = (((pb[j] < 0 ? 0 : pb[j]) * a.H + ih0[j]) * a.W + iw0[j]) * a.Cin;
      uint32_t m = 0;
      if (pb[j] >= 0)
        for (int kh = 0; kh < a.KS; ++kh)
          for (int kw = 0; kw < a.KS; ++kw)
            if ((unsigned)(ih0[j] + kh) < (unsigned)a.H && (unsigned)(iw0[j] + kw) < (unsigned)a.W) m |= 1u << (kh * a.KS + kw);
      rmask[j] = m;
    }
  }
  auto gather = [&](int k0, v4f (&xv)[XP], v4f& wv) {
    if constexpr (!VEC) {
      // any Cin, either weight order: the same four k of the chunk, element by element (index arithmetic per element)
      int kk = k0 + kq;
      int tap = kk / a.Cin, c = kk - tap * a.Cin;
      int kh = tap / a.KS, kw = tap - kh * a.KS;
#pragma unroll
      for (int i = 0; i < 4; ++i, ++kk) {
        const bool kok = kk < K;
#pragma unroll
        for (int j = 0; j < XP; ++j) {
          const int ih = ih0[j] + kh, iw = iw0[j] + kw;
          const bool ok = kok && pb[j] >= 0 && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
          xv[j][i] = ok ? xs[(((int64_t)pb[j] * a.H + ih) * a.W + iw) * a.Cin + c] : 0.f;
        }
        wv[i] = (kok && wrow_ok) ? (a.w_ohwi ? wrow[k0 + i] : ws[(((int64_t)(n0 + row) * a.Cin + c) * a.KS + kh) * a.KS + kw]) : 0.f;
        if (++c == a.Cin) { c = 0; if (++kw == a.KS) { kw = 0; ++kh; } }
      }
      return;
    }
    const bool kok = k0 + kq < K;
    if constexpr (tapmask) {
      const int tap = gkh * a.KS + gkw, toff = (gkh * a.W + gkw) * a.Cin + gc;
#pragma unroll
      for (int j = 0; j < XP; ++j)
        xv[j] = (kok && ((rmask[j] >> tap) & 1u)) ? *reinterpret_cast<const v4f*>(xs + (roff[j] + toff)) : zero4;
    } else {
#pragma unroll
      for (int j = 0; j < XP; ++j) {
        const int ih = ih0[j] + gkh, iw = iw0[j] + gkw;
        const bool ok = kok && pb[j] >= 0 && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
        xv[j] = ok ? *reinterpret_cast<const v4f*>(xs + (((int64_t)pb[j] * a.H + ih) * a.W + iw) * a.Cin + gc) : zero4;
      }
    }
    wv = (kok && wrow_ok) ? *reinterpret_cast<const v4f*>(wrow + k0) : zero4;
    gc += CF_KC;
    while (gc >= a.Cin) { gc -= a.Cin; if (++gkw == a.KS) { gkw = 0; ++gkh; } }
  };
  v16f acc;
  v4d acc64[2][2];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc64[i][j] = v4d{0.0, 0.0, 0.0, 0.0};
  const int l15 = lane & 15, q = lane >> 4;
  v4f xv[XP], wv;
  gather(0, xv, wv);
  for (int k0 = 0; k0 < K; k0 += CF_KC) {
#pragma unroll
    for (int j = 0; j < XP; ++j) *reinterpret_cast<v4f*>(&Bs[(row + 64 * j) * CF_LD4 + kq]) = xv[j];
    if (row < NT) *reinterpret_cast<v4f*>(&As[row * CF_LD4 + kq]) = wv;
    __syncthreads();
    if (k0 + CF_KC < K) gather(k0 + CF_KC, xv, wv);           // next chunk in flight under the MFMAs
    if constexpr (!ACC64) {
      float av[8], bv[8];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const v4f a4 = *reinterpret_cast<const v4f*>(&As[(wn * 32 + (lane & 31)) * CF_LD4 + 8 * (lane >> 5) + 4 * t]);
        const v4f b4 = *reinterpret_cast<const v4f*>(&Bs[(wm * 32 + (lane & 31)) * CF_LD4 + 8 * (lane >> 5) + 4 * t]);
#pragma unroll
        for (int i = 0; i < 4; ++i) { av[4 * t + i] = a4[i]; bv[4 * t + i] = b4[i]; }
      }
#pragma unroll
      for (int t = 0; t < CF_KC / 2; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
    } else {
      v4f p4[2], c4[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        p4[t] = *reinterpret_cast<const v4f*>(&Bs[(wm * 32 + t * 16 + l15) * CF_LD4 + 4 * q]);
        c4[t] = *reinterpret_cast<const v4f*>(&As[(wn * 32 + t * 16 + l15) * CF_LD4 + 4 * q]);
      }
#pragma unroll
      for (int k4 = 0; k4 < CF_KC / 4; ++k4)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc64[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)p4[i][k4], (double)c4[j][k4], acc64[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  float vmin = INFINITY, vmax = -INFINITY;       // of the outputs this thread writes (observer fusion)
  if constexpr (!ACC64) {
    // D[i = channel][j = pixel]: lane owns pixel j = lane & 31, register r holds channel 8 (r / 4) + 4 (lane >> 5) + r % 4
    const int po = p0 + wm * 32 + (lane & 31);
    if (po < npix) {
      float* yp = a.y + (int64_t)s * a.y_ss + (int64_t)po * a.Cout;
      const float* dr = a.drop_mask ? a.drop_mask + ((int64_t)s * a.B + po / (a.Ho * a.Wo)) * a.Cout : nullptr;
      if (a.tail4) {
        // the lane's four registers of a group are four CONSECUTIVE channels of its pixel: one 16-byte store (and 16-byte loads of the
        // per-channel parameters and the residual) instead of four 4-byte ones 4 Cout bytes apart from the neighbouring lane's -- with
        // K = 216 (24 channels) the scalar form's 16 scattered stores per lane were a third of the launch
        const float* rp = a.res ? a.res + (int64_t)s * a.res_ss + (int64_t)po * a.Cout : nullptr;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int no = n0 + wn * 32 + 8 * g + 4 * (lane >> 5);
          if (no < a.Cout) {            // Cout % 4 == 0: all four channels exist
            v4f v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            if (a.div) { const v4f d = *reinterpret_cast<const v4f*>(a.div + no); for (int i = 0; i < 4; ++i) v[i] = v[i] / d[i]; }
            if (a.bias) { const v4f d = *reinterpret_cast<const v4f*>(a.bias + no); for (int i = 0; i < 4; ++i) v[i] = v[i] + d[i]; }
            if (a.alpha) { const v4f d = *reinterpret_cast<const v4f*>(a.alpha + no); for (int i = 0; i < 4; ++i) v[i] = v[i] * d[i]; }
            if (a.beta) { const v4f d = *reinterpret_cast<const v4f*>(a.beta + no); for (int i = 0; i < 4; ++i) v[i] = v[i] + d[i]; }
            if (dr) { const v4f d = *reinterpret_cast<const v4f*>(dr + no); for (int i = 0; i < 4; ++i) { v[i] = v[i] * d[i]; v[i] = v[i] * a.drop_mult; } }
            if (rp) { const v4f d = *reinterpret_cast<const v4f*>(rp + no); for (int i = 0; i < 4; ++i) v[i] = v[i] + d[i]; }
            if (a.relu) for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
            *reinterpret_cast<v4f*>(yp + no) = v;
            for (int i = 0; i < 4; ++i) { vmin = fminf(vmin, v[i]); vmax = fmaxf(vmax, v[i]); }
          }
        }
      } else {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int no = n0 + wn * 32 + 8 * g + 4 * (lane >> 5) + i;
          if (no < a.Cout) {
            const float v = conv_f32_tail(a, acc[4 * g + i], no, s, (int64_t)po * a.Cout + no, dr);
            yp[no] = v; vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
          }
        }
      }
    }
  } else {
    // a lane's 16 outputs are 8 pixels of TWO channels: the per-channel parameters are loaded once (inside conv_f32_tail they were re-read for
    // every output -- the stores in between may alias them as far as the compiler knows)
    float pdiv[2], pbias[2], palpha[2], pbeta[2];
    int nn[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      nn[j] = n0 + wn * 32 + j * 16 + l15;
      const int nc = nn[j] < a.Cout ? nn[j] : 0;
      pdiv[j] = a.div ? a.div[nc] : 1.f; pbias[j] = a.bias ? a.bias[nc] : 0.f; palpha[j] = a.alpha ? a.alpha[nc] : 1.f; pbeta[j] = a.beta ? a.beta[nc] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int po = p0 + wm * 32 + i * 16 + 4 * r + q;
        if (po >= npix) continue;
        float* yp = a.y + (int64_t)s * a.y_ss + (int64_t)po * a.Cout;
        const float* dr = a.drop_mask ? a.drop_mask + ((int64_t)s * a.B + po / (a.Ho * a.Wo)) * a.Cout : nullptr;
        const float* rp = a.res ? a.res + (int64_t)s * a.res_ss + (int64_t)po * a.Cout : nullptr;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int no = nn[j];
          if (no < a.Cout) {
            float v = (float)acc64[i][j][r];            // the same steps, each rounded to fp32, as conv_f32_tail
            if (a.div) v = v / pdiv[j];
            if (a.bias) v = v + pbias[j];
            if (a.alpha) v = v * palpha[j];
            if (a.beta) v = v + pbeta[j];
            if (dr) { v = v * dr[no]; v = v * a.drop_mult; }
            if (rp) v = v + rp[no];
            if (a.relu) v = fmaxf(v, 0.f);
            yp[no] = v; vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
          }
        }
      }
  }
  if (a.mm_partials) {
    // (min, max) of this workgroup's outputs -> its slot of the observer's partials (no second pass over the tensor)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { vmin = fminf(vmin, __shfl_xor(vmin, o)); vmax = fmaxf(vmax, __shfl_xor(vmax, o)); }
    float* red = As;                        // the K loop's last barrier has passed: the staging tiles are free
    if (lane == 0) { red[2 * wave] = vmin; red[2 * wave + 1] = vmax; }
    __syncthreads();
    if (tid == 0) {
      const int64_t slot = ((int64_t)s * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      a.mm_partials[2 * slot] = fminf(fminf(red[0], red[2]), fminf(red[4], red[6]));
      a.mm_partials[2 * slot + 1] = fmaxf(fmaxf(red[1], red[3]), fmaxf(red[5], red[7]));
    }
  }
}

static bool conv_f32_narrow(int Cout) { return (Cout + 31) / 32 * 32 < (Cout + 63) / 64 * 64; }

// workgroups per sample of the conv launch = length of one sample's row of `minmax_partials`
QBNN_EXPORT int32_t qbnn_conv2d_f32_blocks(int32_t B, int32_t H, int32_t W, int32_t Cout, int32_t ksize, int32_t stride, int32_t pad) {
  const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
  if (B <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0) return 0;
  const int64_t npix = (int64_t)B * Ho * Wo;
  return conv_f32_narrow(Cout) ? (int32_t)(((npix + 127) / 128) * ((Cout + 31) / 32)) : (int32_t)(((npix + 63) / 64) * ((Cout + 63) / 64));
}

static int conv2d_f32_launch(const float* x, int64_t x_ss, const float* w, int64_t w_ss, const float* div, const float* bias,
                             const float* alpha, const float* beta, const float* drop_mask, float drop_mult, const float* res, int64_t res_ss,
                             float* y, int64_t y_ss, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride,
                             int32_t pad, int32_t relu, int32_t n_samples, float* minmax_partials, void* stream);

QBNN_EXPORT int qbnn_conv2d_f32_fused_mc(const float* x, int64_t x_ss, const float* w, int64_t w_ss, const float* div, const float* bias,
                                         const float* alpha, const float* beta, const float* res, int64_t res_ss, float* y, int64_t y_ss, int32_t B, int32_t H,
                                         int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride, int32_t pad, int32_t relu,
                                         int32_t n_samples, float* minmax_partials, void* stream) {
  return conv2d_f32_launch(x, x_ss, w, w_ss, div, bias, alpha, beta, nullptr, 1.0f, res, res_ss, y, y_ss, B, H, W, Cin, Cout, ksize, stride, pad, relu,
                           n_samples, minmax_partials, stream);
}

QBNN_EXPORT int qbnn_conv2d_f32_drop_mc(const float* x, int64_t x_ss, const float* w, int64_t w_ss, const float* bias, const float* alpha,
                                        const float* beta, const float* drop_mask, float drop_mult, const float* res, int64_t res_ss, float* y,
                                        int64_t y_ss, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride,
                                        int32_t pad, int32_t relu, int32_t n_samples, void* stream) {
  return conv2d_f32_launch(x, x_ss, w, w_ss, nullptr, bias, alpha, beta, drop_mask, drop_mult, res, res_ss, y, y_ss, B, H, W, Cin, Cout, ksize, stride, pad,
                           relu, n_samples, nullptr, stream);
}

static int conv2d_f32_launch(const float* x, int64_t x_ss, const float* w, int64_t w_ss, const float* div, const float* bias,
                             const float* alpha, const float* beta, const float* drop_mask, float drop_mult, const float* res, int64_t res_ss,
                             float* y, int64_t y_ss, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride,
                             int32_t pad, int32_t relu, int32_t n_samples, float* minmax_partials, void* stream) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || ksize <= 0 || stride <= 0 || pad < 0 || n_samples <= 0)
    return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_conv2d_f32_mc: bad argument");
  ConvF32Args a;
  a.drop_mask = drop_mask; a.drop_mult = drop_mult;
  {
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    a.tail4 = (Cout % 4) == 0 && al16(y) && (y_ss % 4) == 0 && al16(div) && al16(bias) && al16(alpha) && al16(beta) && al16(drop_mask) && al16(res) && (res_ss % 4) == 0;
  }
  a.div = div; a.alpha = alpha; a.beta = beta; a.res = res; a.res_ss = res_ss; a.mm_partials = minmax_partials;
  a.x = x; a.x_ss = x_ss; a.w = w; a.w_ss = w_ss; a.bias = bias; a.y = y; a.y_ss = y_ss;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KS = ksize; a.stride = stride; a.pad = pad; a.relu = relu & 1; a.w_ohwi = (relu >> 2) & 1;
  a.Ho = (H + 2 * pad - ksize) / stride + 1; a.Wo = (W + 2 * pad - ksize) / stride + 1;
  if (a.Ho <= 0 || a.Wo <= 0) return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_conv2d_f32_mc: empty output");
  const int64_t npix = (int64_t)B * a.Ho * a.Wo;
  dim3 grid((unsigned)((npix + 63) / 64), (unsigned)((Cout + 63) / 64), (unsigned)n_samples);
  const bool vec = a.w_ohwi && (Cin % 4) == 0 && (x_ss % 4) == 0 && (w_ss % 4) == 0 &&
                   (reinterpret_cast<uintptr_t>(x) % 16) == 0 && (reinterpret_cast<uintptr_t>(w) % 16) == 0;
  static const bool no_tapmask = [] { const char* e = getenv("QBNN_F32_TAPMASK"); return e && e[0] == '0'; }();
  const bool tapmask = !no_tapmask && vec && ksize * ksize <= 32 && (int64_t)B * H * W * Cin < (int64_t(1) << 30);
  const bool acc64 = (relu & 2) != 0;
  hipStream_t st = (hipStream_t)stream;
  // 128 pixels x 32 channels per workgroup where 32-wide channel tiles pad less than 64-wide ones (Cout = 24, 96, ...)
  const bool narrow = conv_f32_narrow(Cout);
  const dim3 g2((unsigned)((npix + 127) / 128), (unsigned)((Cout + 31) / 32), (unsigned)n_samples);
#define QBNN_F32_LAUNCH(A64, V)                                                                                             \
  do {                                                                                                                      \
    if (narrow) hipLaunchKernelGGL((conv2d_f32_vec_kernel<128, 32, A64, V>), g2, dim3(256), 0, st, a);                       \
    else hipLaunchKernelGGL((conv2d_f32_vec_kernel<64, 64, A64, V>), grid, dim3(256), 0, st, a);                             \
  } while (0)
  if (acc64 && tapmask) QBNN_F32_LAUNCH(true, 2);
  else if (acc64 && vec) QBNN_F32_LAUNCH(true, 1);
  else if (acc64) QBNN_F32_LAUNCH(true, 0);
  else if (tapmask) QBNN_F32_LAUNCH(false, 2);
  else if (vec) QBNN_F32_LAUNCH(false, 1);
  else QBNN_F32_LAUNCH(false, 0);
#undef QBNN_F32_LAUNCH
  return qbnn_check_launch_msg("qbnn_conv2d_f32_mc");
}

// =====================================================================================
// Pointwise: per-channel affine (BatchNorm eval as ATen computes it: x * alpha + beta, two roundings;
// QAT conv-bn:  Z / c + b), optional second operand (Add), optional ReLU.  Channels are the fastest axis (NHWC).
//   mode 0: v = x * p0[c] + p1[c]     mode 1: v = x / p0[c] + p1[c]     (p0 / p1 null = skip that step)
//   then v += res (if given), then ReLU (if asked)
// =====================================================================================
__global__ __launch_bounds__(256) void affine_f32_kernel(const float* __restrict__ x, int64_t x_ss, const float* __restrict__ res,
                                                          int64_t res_ss, const float* __restrict__ p0, const float* __restrict__ p1,
                                                          float* __restrict__ y, int64_t y_ss, int64_t n, int C, int mode, int relu) {
  const int s = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    float v = x[(int64_t)s * x_ss + i];
    if (p0) v = mode == 0 ? v * p0[c] : v / p0[c];
    if (p1) v = v + p1[c];
    if (res) v = v + res[(int64_t)s * res_ss + i];
    if (relu) v = fmaxf(v, 0.f);
    y[(int64_t)s * y_ss + i] = v;
  }
}

// the same on float4s (n, C, strides multiples of 4, 16-byte aligned operands, n < 2^31): a quarter of the memory instructions and a 32-bit
// channel index instead of a 64-bit modulo per element
static inline bool f32_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
__global__ __launch_bounds__(256) void affine_f32_v4_kernel(const float* __restrict__ x, int64_t x_ss, const float* __restrict__ res,
                                                             int64_t res_ss, const float* __restrict__ p0, const float* __restrict__ p1,
                                                             float* __restrict__ y, int64_t y_ss, uint32_t n4, uint32_t C, int mode, int relu) {
  const int s = blockIdx.y;
  const v4f* xs = reinterpret_cast<const v4f*>(x + (int64_t)s * x_ss);
  const v4f* rs = res ? reinterpret_cast<const v4f*>(res + (int64_t)s * res_ss) : nullptr;
  v4f* ys = reinterpret_cast<v4f*>(y + (int64_t)s * y_ss);
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u) {
    const uint32_t c = (i * 4u) % C;
    v4f v = xs[i];
    if (p0) { const v4f d = *reinterpret_cast<const v4f*>(p0 + c); for (int k = 0; k < 4; ++k) v[k] = mode == 0 ? v[k] * d[k] : v[k] / d[k]; }
    if (p1) { const v4f d = *reinterpret_cast<const v4f*>(p1 + c); for (int k = 0; k < 4; ++k) v[k] = v[k] + d[k]; }
    if (rs) { const v4f d = rs[i]; for (int k = 0; k < 4; ++k) v[k] = v[k] + d[k]; }
    if (relu) for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
    ys[i] = v;
  }
}

QBNN_EXPORT int qbnn_affine_f32_mc(const float* x, int64_t x_ss, const float* res, int64_t res_ss, const float* p0, const float* p1,
                                   float* y, int64_t y_ss, int64_t n, int32_t C, int32_t mode, int32_t relu, int32_t n_samples,
                                   void* stream) {
  if (!x || !y || n <= 0 || C <= 0 || n_samples <= 0 || mode < 0 || mode > 1)
    return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_affine_f32_mc: bad argument");
  if ((n & 3) == 0 && (C & 3) == 0 && n < (int64_t(1) << 31) && ((x_ss | y_ss | res_ss) & 3) == 0 && f32_al16(x) && f32_al16(y) && f32_al16(res) &&
      f32_al16(p0) && f32_al16(p1)) {
    const int64_t n4 = n / 4;
    const int blocks4 = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(affine_f32_v4_kernel, dim3(blocks4, n_samples), dim3(256), 0, (hipStream_t)stream, x, x_ss, res, res_ss, p0, p1, y, y_ss,
                       (uint32_t)n4, (uint32_t)C, mode, relu);
    return qbnn_check_launch_msg("qbnn_affine_f32_mc");
  }
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(affine_f32_kernel, dim3(blocks, n_samples), dim3(256), 0, (hipStream_t)stream, x, x_ss, res, res_ss, p0, p1, y,
                     y_ss, n, C, mode, relu);
  return qbnn_check_launch_msg("qbnn_affine_f32_mc");
}

// ---- float BernoulliDropout (mcdropout/dropout.py:15-40 with FloatFunctional, q=False) --------------------------------------
// The mask: one Bernoulli(keep) draw per slot (slot = b * C + c for a 4-D activation: whole channels drop; the element index for a
// 2-D one) from the SAME Philox uniform stream as the quantised dropout, {ctr = {i >> 2, layer, sample, 1}}[i & 3], as fp32 0 / 1.
__global__ __launch_bounds__(256) void dropout_mask_f32_kernel(int64_t n_slots, float keep, uint32_t seed_lo, uint32_t seed_hi, uint32_t layer_id,
                                                                uint32_t sample_begin, float* __restrict__ mask, const uint32_t* __restrict__ nd) {
  if (nd) { seed_lo = nd[0]; seed_hi = nd[1]; sample_begin = nd[2]; }      // captured-graph mode: the seed lives in device memory
  const int s = blockIdx.y;
  for (int64_t blk = (int64_t)blockIdx.x * 256 + threadIdx.x; 4 * blk < n_slots; blk += (int64_t)gridDim.x * 256) {
    const qbnn::u32x4 r = qbnn::philox4x32_10((uint32_t)blk, layer_id, sample_begin + s, 1u, seed_lo, seed_hi);
    const uint32_t rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * blk + j < n_slots) mask[(int64_t)s * n_slots + 4 * blk + j] = ((float)(rv[j] >> 8) * 5.9604644775390625e-8f) < keep ? 1.0f : 0.0f;
  }
}

QBNN_EXPORT int qbnn_dropout_mask_f32_mc(int64_t n_slots, float keep_prob, uint64_t seed, uint32_t layer_id, uint32_t sample_begin,
                                         int32_t n_samples, float* mask_out, void* stream) {
  if (!mask_out || n_slots <= 0 || n_samples <= 0) return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_dropout_mask_f32_mc: bad argument");
  const int64_t nb = (n_slots + 3) / 4;
  const int blocks = (int)((nb + 255) / 256 < 1024 ? (nb + 255) / 256 : 1024);
  hipLaunchKernelGGL(dropout_mask_f32_kernel, dim3(blocks, n_samples), dim3(256), 0, (hipStream_t)stream, n_slots, keep_prob, (uint32_t)seed,
                     (uint32_t)(seed >> 32), layer_id, sample_begin, mask_out, qbnn_noise_dev());
  return qbnn_check_launch_msg("qbnn_dropout_mask_f32_mc");
}

// y = ((x * mask[s][b][c]) * multiplier (+ res)) (ReLU) on x [S|1][B][HW][C]: FloatFunctional.mul then mul_scalar, two fp32 roundings.
__global__ __launch_bounds__(256) void dropout_f32_kernel(const float* __restrict__ x, int64_t x_ss, const float* __restrict__ mask, int64_t HWC,
                                                           int C, float mult, const float* __restrict__ res, int64_t res_ss, int relu,
                                                           float* __restrict__ y, int64_t y_ss, int64_t n, int64_t n_slots) {
  const int s = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / HWC;
    const int c = (int)(i % C);
    float v = x[(int64_t)s * x_ss + i] * mask[(int64_t)s * n_slots + b * C + c];
    v = v * mult;
    if (res) v = v + res[(int64_t)s * res_ss + i];
    if (relu) v = fmaxf(v, 0.f);
    y[(int64_t)s * y_ss + i] = v;
  }
}

QBNN_EXPORT int qbnn_dropout_f32_mc(const float* x, int64_t x_ss, const float* mask, int32_t B, int32_t HW, int32_t C, float multiplier,
                                    const float* res, int64_t res_ss, int32_t relu, float* y, int64_t y_ss, int32_t n_samples, void* stream) {
  if (!x || !mask || !y || B <= 0 || HW <= 0 || C <= 0 || n_samples <= 0) return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_dropout_f32_mc: bad argument");
  const int64_t n = (int64_t)B * HW * C;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(dropout_f32_kernel, dim3(blocks, n_samples), dim3(256), 0, (hipStream_t)stream, x, x_ss, mask, (int64_t)HW * C, C, multiplier, res,
                     res_ss, relu, y, y_ss, n, (int64_t)B * C);
  return qbnn_check_launch_msg("qbnn_dropout_f32_mc");
}

// ---- pooling (NHWC, kernel = stride = k, no padding): mode 0 max (nn.MaxPool2d), 1 average (nn.AvgPool2d) ----------
__global__ __launch_bounds__(256) void pool2d_f32_kernel(const float* __restrict__ x, int64_t x_ss, float* __restrict__ y,
                                                          int64_t y_ss, int B, int H, int W, int C, int k, int mode) {
  const int Ho = H / k, Wo = W / k;
  const int64_t n = (int64_t)B * Ho * Wo * C;
  const int s = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t t = i / C;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float m = mode == 0 ? -INFINITY : 0.f;
    for (int dh = 0; dh < k; ++dh)
      for (int dw = 0; dw < k; ++dw) {
        const float v = x[(int64_t)s * x_ss + (((int64_t)b * H + oh * k + dh) * W + ow * k + dw) * C + c];
        m = mode == 0 ? fmaxf(m, v) : m + v;
      }
    y[(int64_t)s * y_ss + i] = mode == 0 ? m : m / (float)(k * k);
  }
}

QBNN_EXPORT int qbnn_pool2d_f32_mc(const float* x, int64_t x_ss, float* y, int64_t y_ss, int32_t B, int32_t H, int32_t W, int32_t C,
                                   int32_t k, int32_t mode, int32_t n_samples, void* stream) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || H % k || W % k || n_samples <= 0 || mode < 0 || mode > 1)
    return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_pool2d_f32_mc: bad argument");
  const int64_t n = (int64_t)B * (H / k) * (W / k) * C;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(pool2d_f32_kernel, dim3(blocks, n_samples), dim3(256), 0, (hipStream_t)stream, x, x_ss, y, y_ss, B, H, W, C, k, mode);
  return qbnn_check_launch_msg("qbnn_pool2d_f32_mc");
}

// ---- Flatten of an NHWC activation in the reference's NCHW order: y[s][b][c * HW + p] = x[s][b][p][c] ---------------
__global__ __launch_bounds__(256) void flatten_nchw_f32_kernel(const float* __restrict__ x, int64_t x_ss, float* __restrict__ y,
                                                                int64_t y_ss, int B, int HW, int C) {
  const int64_t n = (int64_t)B * HW * C;
  const int s = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int p = (int)(i % HW);
    int64_t t = i / HW;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    y[(int64_t)s * y_ss + i] = x[(int64_t)s * x_ss + ((int64_t)b * HW + p) * C + c];
  }
}

QBNN_EXPORT int qbnn_flatten_nchw_f32_mc(const float* x, int64_t x_ss, int32_t B, int32_t HW, int32_t C, float* y, int64_t y_ss,
                                         int32_t n_samples, void* stream) {
  if (!x || !y || B <= 0 || HW <= 0 || C <= 0 || n_samples <= 0) return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_flatten_nchw_f32_mc: bad argument");
  const int64_t n = (int64_t)B * HW * C;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(flatten_nchw_f32_kernel, dim3(blocks, n_samples), dim3(256), 0, (hipStream_t)stream, x, x_ss, y, y_ss, B, HW, C);
  return qbnn_check_launch_msg("qbnn_flatten_nchw_f32_mc");
}

// ---- F.softmax(dim=-1) on [S][B][N] fp32 logits (one thread per row; N is the class count) --------------------------
__global__ __launch_bounds__(256) void softmax_f32_kernel(const float* __restrict__ x, int64_t x_ss, int B, int N, float* __restrict__ probs) {
  const int b = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
  if (b >= B) return;
  const float* xp = x + (int64_t)s * x_ss + (int64_t)b * N;
  float m = -INFINITY;
  for (int j = 0; j < N; ++j) m = fmaxf(m, xp[j]);
  float sum = 0.f;
  for (int j = 0; j < N; ++j) sum += expf(xp[j] - m);
  float* pp = probs + ((int64_t)s * B + b) * N;
  for (int j = 0; j < N; ++j) pp[j] = expf(xp[j] - m) / sum;
}

QBNN_EXPORT int qbnn_softmax_f32_mc(const float* x, int64_t x_ss, int32_t B, int32_t N, float* probs, int32_t n_samples, void* stream) {
  if (!x || !probs || B <= 0 || N <= 0 || n_samples <= 0) return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_softmax_f32_mc: bad argument");
  hipLaunchKernelGGL(softmax_f32_kernel, dim3((B + 255) / 256, n_samples), dim3(256), 0, (hipStream_t)stream, x, x_ss, B, N, probs);
  return qbnn_check_launch_msg("qbnn_softmax_f32_mc");
}

// =====================================================================================
// QAT evaluation: observers and fake quantisation
// =====================================================================================
// per-sample (min, max) of x[s][0..n): partials [S][nblk][2], nblk = gridDim.x
__global__ __launch_bounds__(256) void minmax_f32_kernel(const float* __restrict__ x, int64_t x_ss, int64_t n, float* __restrict__ partials) {
  __shared__ float smin[4], smax[4];
  const int s = blockIdx.y;
  float mn = INFINITY, mx = -INFINITY;
  const float* xs = x + (int64_t)s * x_ss;
  if ((n & 3) == 0 && (x_ss & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {      // float4s: a quarter of the load instructions
    const v4f* x4 = reinterpret_cast<const v4f*>(xs);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += (int64_t)gridDim.x * 256) {
      const v4f v = x4[i];
      mn = fminf(fminf(mn, v[0]), fminf(v[1], fminf(v[2], v[3]))); mx = fmaxf(fmaxf(mx, v[0]), fmaxf(v[1], fmaxf(v[2], v[3])));
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
      const float v = xs[i];
      mn = fminf(mn, v); mx = fmaxf(mx, v);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = mn; smax[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    mn = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
    mx = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    float* o = partials + ((int64_t)s * gridDim.x + blockIdx.x) * 2;
    o[0] = mn; o[1] = mx;
  }
}

// MovingAverageMinMaxObserver.forward + calculate_qparams (torch/quantization/observer.py; per-tensor affine), run for
// the S samples IN ORDER by one workgroup:  first call: (min, max) = (cur_min, cur_max); later:
// min += c (cur_min - min), max += c (cur_max - max).  qparams: lo = min(min, 0), hi = max(max, 0),
// scale = max((hi - lo) / float(qmax - qmin), eps_f32), zp = clamp(qmin - round(lo / scale), qmin, qmax).
// state[0..1] = (min, max), state[2] = 0 until the observer has seen data (all fp32, read and written back).
__global__ __launch_bounds__(1024) void observer_scan_kernel(const float* __restrict__ partials, int nblk, int n_samples, float* __restrict__ state,
                                                             float avg_const, int qmin, int qmax, float* __restrict__ scale, int* __restrict__ zp) {
  // Round 4: every sample's (min, max) over its partials first, all samples in flight at once (wave w takes samples w, w + 4, ...; min / max are
  // exact in any order), then the S-step recurrence on one thread -- instead of S serial rounds of load -> reduce -> two barriers (10.8 us per
  // observer, 114 observers per QAT forward).
  __shared__ float smin[256], smax[256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int s0 = 0; s0 < n_samples; s0 += 256) {             // (more than 256 samples: in rounds)
    const int ns = n_samples - s0 < 256 ? n_samples - s0 : 256;
    const int nw = blockDim.x >> 6;                          // (round 5: 16 waves -- a conv of 4,096 workgroups per sample leaves 40,960 pairs to reduce)
    for (int s = wave; s < ns; s += nw) {
      float mn = INFINITY, mx = -INFINITY;
      const float2* ps = reinterpret_cast<const float2*>(partials + (int64_t)(s0 + s) * nblk * 2);
      for (int i = lane; i < nblk; i += 64) { const float2 v = ps[i]; mn = fminf(mn, v.x); mx = fmaxf(mx, v.y); }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
      if (lane == 0) { smin[s] = mn; smax[s] = mx; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float mn_state = state[0], mx_state = state[1];
      bool init = state[2] != 0.f;
      for (int s = 0; s < ns; ++s) {
        const float mn = smin[s], mx = smax[s];
        if (!init) { mn_state = mn; mx_state = mx; init = true; }
        else { mn_state = mn_state + avg_const * (mn - mn_state); mx_state = mx_state + avg_const * (mx - mx_state); }
        const float lo = fminf(mn_state, 0.f), hi = fmaxf(mx_state, 0.f);
        float sc = (hi - lo) / (float)(qmax - qmin);
        sc = fmaxf(sc, 1.1920928955078125e-07f);
        float z = (float)qmin - rintf(lo / sc);
        z = fminf(fmaxf(z, (float)qmin), (float)qmax);
        scale[s0 + s] = sc; zp[s0 + s] = (int)z;
      }
      state[0] = mn_state; state[1] = mx_state; state[2] = 1.f;
    }
    __syncthreads();
  }
}

// The observer recurrence alone, on per-workgroup (min, max) partials a producer already wrote (qbnn_conv2d_f32_fused_mc's
// minmax_partials): [S][n_blocks][2].
QBNN_EXPORT int qbnn_observe_partials_f32_mc(const float* partials, int32_t n_blocks, int32_t n_samples, float* state, float avg_const,
                                             int32_t qmin, int32_t qmax, float* scale, int32_t* zero_point, void* stream) {
  if (!partials || !state || !scale || !zero_point || n_blocks <= 0 || n_samples <= 0 || qmax <= qmin)
    return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_observe_partials_f32_mc: bad argument");
  hipLaunchKernelGGL(observer_scan_kernel, dim3(1), dim3(n_blocks > 64 ? 1024 : 256), 0, (hipStream_t)stream, partials, n_blocks, n_samples, state, avg_const, qmin,
                     qmax, scale, zero_point);
  return qbnn_check_launch_msg("qbnn_observe_partials_f32_mc");
}

QBNN_EXPORT int qbnn_observe_f32_mc(const float* x, int64_t x_ss, int64_t n, int32_t n_samples, float* state, float avg_const,
                                    int32_t qmin, int32_t qmax, float* workspace, float* scale, int32_t* zero_point, void* stream) {
  if (!x || !state || !workspace || !scale || !zero_point || n <= 0 || n_samples <= 0 || qmax <= qmin)
    return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_observe_f32_mc: bad argument");
  const int nblk = (int)((n + 255) / 256 < QBNN_OBSERVER_BLOCKS ? (n + 255) / 256 : QBNN_OBSERVER_BLOCKS);
  hipLaunchKernelGGL(minmax_f32_kernel, dim3(nblk, n_samples), dim3(256), 0, (hipStream_t)stream, x, x_ss, n, workspace);
  hipLaunchKernelGGL(observer_scan_kernel, dim3(1), dim3(nblk > 64 ? 1024 : 256), 0, (hipStream_t)stream, workspace, nblk, n_samples, state, avg_const, qmin,
                     qmax, scale, zero_point);
  return qbnn_check_launch_msg("qbnn_observe_f32_mc");
}

// fake_quantize_per_tensor_affine with per-sample qparams: y = (clamp(rne(x * (1 / s)) + z, qmin, qmax) - z) * s
__global__ __launch_bounds__(256) void fake_quant_f32_kernel(const float* __restrict__ x, int64_t x_ss, float* __restrict__ y, int64_t y_ss,
                                                              int64_t n, const float* __restrict__ scale, const int* __restrict__ zp,
                                                              int qp_stride, int qmin, int qmax) {
  const int s = blockIdx.y;
  const float sc = scale[s * qp_stride], inv = 1.0f / sc;
  const float z = (float)zp[s * qp_stride], lo = (float)qmin, hi = (float)qmax;
  if ((n & 3) == 0 && ((x_ss | y_ss) & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {      // float4s
    const v4f* x4 = reinterpret_cast<const v4f*>(x + (int64_t)s * x_ss);
    v4f* y4 = reinterpret_cast<v4f*>(y + (int64_t)s * y_ss);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += (int64_t)gridDim.x * 256) {
      v4f v = x4[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (fminf(fmaxf(rintf(v[k] * inv) + z, lo), hi) - z) * sc;
      y4[i] = v;
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float q = fminf(fmaxf(rintf(x[(int64_t)s * x_ss + i] * inv) + z, lo), hi);
    y[(int64_t)s * y_ss + i] = (q - z) * sc;
  }
}

// ... with the ReLU that follows it in the graph (BasicBlock: Add -> FakeQuantize -> ReLU; a ReLU of grid values stays on the grid) and, optionally, the
// grid integers themselves, m = q - z (after the ReLU: max(m, 0)), as int8 [S][n] -- the operand format of qbnn_conv2d_q8_f32_mc, so the consumer conv
// needs no qbnn_grid_to_i8_mc pass.  q8 requires qmax - qmin <= 127 (m = q - z spans +-(qmax - qmin): |m| <= 127 for the 2- to 7-bit activation grids).
__global__ __launch_bounds__(256) void fake_quant_ex_f32_kernel(const float* __restrict__ x, int64_t x_ss, float* __restrict__ y, int64_t y_ss,
                                                                 int64_t n, const float* __restrict__ scale, const int* __restrict__ zp, int qmin, int qmax,
                                                                 int relu, int8_t* __restrict__ q8) {
  const int s = blockIdx.y;
  const float sc = scale[s], inv = 1.0f / sc;
  const float z = (float)zp[s], lo = (float)qmin, hi = (float)qmax;
  const float mlo = relu ? 0.f : -INFINITY;
  if ((n & 3) == 0 && ((x_ss | y_ss) & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(q8)) & 15) == 0) {
    const v4f* x4 = reinterpret_cast<const v4f*>(x + (int64_t)s * x_ss);
    v4f* y4 = reinterpret_cast<v4f*>(y + (int64_t)s * y_ss);
    uint32_t* q4 = q8 ? reinterpret_cast<uint32_t*>(q8 + (int64_t)s * n) : nullptr;
    auto quant4 = [&](v4f& v) {
      uint32_t pk = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float m = fmaxf(fminf(fmaxf(rintf(v[k] * inv) + z, lo), hi) - z, mlo);      // the grid integer q - z (ReLU: an integer clamp of it)
        v[k] = m * sc;
        pk |= ((uint32_t)(int)m & 0xffu) << (8 * k);
      }
      return pk;
    };
    // four float4s per thread and trip, all four loads in flight before the first is used (round 6: the pass is HBM-bound and ran at 3.5 TB/s with one)
    const int64_t T = (int64_t)gridDim.x * 256, n4 = n / 4;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * T < n4; i += 4 * T) {
      v4f v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = x4[i + u * T];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t pk = quant4(v[u]);
        if (y) y4[i + u * T] = v[u];            // (y NULL: the consumers take the int8 grid tensor + the per-sample scale -- round 6)
        if (q4) q4[i + u * T] = pk;
      }
    }
    for (; i < n4; i += T) {
      v4f v = x4[i];
      const uint32_t pk = quant4(v);
      if (y) y4[i] = v;
      if (q4) q4[i] = pk;
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float m = fmaxf(fminf(fmaxf(rintf(x[(int64_t)s * x_ss + i] * inv) + z, lo), hi) - z, mlo);
    if (y) y[(int64_t)s * y_ss + i] = m * sc;
    if (q8) q8[(int64_t)s * n + i] = (int8_t)(int)m;
  }
}

QBNN_EXPORT int qbnn_fake_quant_ex_f32_mc(const float* x, int64_t x_ss, float* y, int64_t y_ss, int64_t n, const float* scale,
                                          const int32_t* zero_point, int32_t qmin, int32_t qmax, int32_t relu, int8_t* q8_out, int32_t n_samples,
                                          void* stream) {
  if (!x || (!y && !q8_out) || !scale || !zero_point || n <= 0 || n_samples <= 0 || qmax <= qmin)
    return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_fake_quant_ex_f32_mc: bad argument");
  if (q8_out && qmax - qmin > 127) return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_fake_quant_ex_f32_mc: the int8 output (q - z) takes grids of at most 128 steps (qmax - qmin <= 127)");
  const int64_t want = (n / 16 + 255) / 256;          // 16 elements per thread and trip
  const int blocks = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  hipLaunchKernelGGL(fake_quant_ex_f32_kernel, dim3(blocks, n_samples), dim3(256), 0, (hipStream_t)stream, x, x_ss, y, y_ss, n, scale, zero_point,
                     qmin, qmax, relu, q8_out);
  return qbnn_check_launch_msg("qbnn_fake_quant_ex_f32_mc");
}

// FloatFunctional.add of two fake-quantised tensors given as their grid integers (src/utils.py:49-55 `Add`, BasicBlock's out + shortcut in the prepared
// graph): y[s][i] = fl32((float)a[s][i] * s_a[s]) + fl32((float)b[s][i] * s_b[s]) -- each addend is exactly the fp32 value its FakeQuantize would have
// written ((q - z) * scale), so the sum equals the fp32 Add bit for bit while the operands cost 1 byte per element instead of 4 --, with the
// per-workgroup (min, max) of the sums for the Add's own observer (no separate min / max pass over y).
__global__ __launch_bounds__(256) void add_q8_f32_kernel(const int8_t* __restrict__ a, int64_t a_ss, const float* __restrict__ sa, const int8_t* __restrict__ b,
                                                          int64_t b_ss, const float* __restrict__ sb, float* __restrict__ y, int64_t y_ss, int64_t n,
                                                          float* __restrict__ partials) {
  __shared__ float red[8];
  const int s = blockIdx.y, tid = threadIdx.x;
  const float fa = sa[s], fb = sb[s];
  const int8_t* as = a + (int64_t)s * a_ss;
  const int8_t* bs = b + (int64_t)s * b_ss;
  float* ys = y + (int64_t)s * y_ss;
  float vmin = INFINITY, vmax = -INFINITY;
  const bool vec = (n & 15) == 0 && ((a_ss | b_ss) & 15) == 0 && (y_ss & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  if (vec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n / 16; i += (int64_t)gridDim.x * 256) {
      const v4i_q8 av = reinterpret_cast<const v4i_q8*>(as)[i], bv = reinterpret_cast<const v4i_q8*>(bs)[i];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        v4f o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float t = (float)(int8_t)(av[d] >> (8 * k)) * fa + (float)(int8_t)(bv[d] >> (8 * k)) * fb;
          o[k] = t;
          vmin = fminf(vmin, t); vmax = fmaxf(vmax, t);
        }
        if (y) reinterpret_cast<v4f*>(ys)[i * 4 + d] = o;      // (y NULL: only the observer's partials; the fake-quantiser recomputes the sums, qbnn_fake_quant_add_q8_mc)
      }
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n; i += (int64_t)gridDim.x * 256) {
      const float t = (float)as[i] * fa + (float)bs[i] * fb;
      if (y) ys[i] = t;
      vmin = fminf(vmin, t); vmax = fmaxf(vmax, t);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { vmin = fminf(vmin, __shfl_xor(vmin, o)); vmax = fmaxf(vmax, __shfl_xor(vmax, o)); }
  if ((tid & 63) == 0) { red[2 * (tid >> 6)] = vmin; red[2 * (tid >> 6) + 1] = vmax; }
  __syncthreads();
  if (tid == 0 && partials) {
    const int64_t slot = (int64_t)s * gridDim.x + blockIdx.x;
    partials[2 * slot] = fminf(fminf(red[0], red[2]), fminf(red[4], red[6]));
    partials[2 * slot + 1] = fmaxf(fmaxf(red[1], red[3]), fmaxf(red[5], red[7]));
  }
}

// workgroups per sample of qbnn_add_q8_f32_mc = length of one sample's row of `minmax_partials`
QBNN_EXPORT int32_t qbnn_add_q8_blocks(int64_t n) {
  if (n <= 0) return 0;
  const int64_t b = (n / 16 + 255) / 256;
  return (int32_t)(b < 1 ? 1 : (b > 512 ? 512 : b));
}

QBNN_EXPORT int qbnn_add_q8_f32_mc(const int8_t* a, int64_t a_ss, const float* s_a, const int8_t* b, int64_t b_ss, const float* s_b, float* y, int64_t y_ss,
                                   int64_t n, int32_t n_samples, float* minmax_partials, void* stream) {
  if (!a || !b || !s_a || !s_b || (!y && !minmax_partials) || n <= 0 || n_samples <= 0) return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_add_q8_f32_mc: bad argument");
  hipLaunchKernelGGL(add_q8_f32_kernel, dim3(qbnn_add_q8_blocks(n), n_samples), dim3(256), 0, (hipStream_t)stream, a, a_ss, s_a, b, b_ss, s_b, y, y_ss, n,
                     minmax_partials);
  return qbnn_check_launch_msg("qbnn_add_q8_f32_mc");
}

// The Add's FakeQuantize (+ the ReLU behind it) straight from the Add's OPERANDS: x = fl32((float)a s_a) + fl32((float)b s_b) is recomputed per element --
// exactly add_q8_f32_kernel's sum -- so the fp32 sum tensor is neither written by the Add nor read here: 2 + 2 + 1 bytes per element for Add + observer +
// fake-quantise instead of 2 + 4 + 4 + 1.  Outputs as qbnn_fake_quant_ex_f32_mc: the grid integers q - z (int8) and / or the fp32 values.
__global__ __launch_bounds__(256) void fake_quant_add_q8_kernel(const int8_t* __restrict__ a, int64_t a_ss, const float* __restrict__ sa, const int8_t* __restrict__ b,
                                                                 int64_t b_ss, const float* __restrict__ sb, float* __restrict__ y, int64_t y_ss, int64_t n,
                                                                 const float* __restrict__ scale, const int* __restrict__ zp, int qmin, int qmax, int relu,
                                                                 int8_t* __restrict__ q8) {
  const int s = blockIdx.y, tid = threadIdx.x;
  const float fa = sa[s], fb = sb[s];
  const float sc = scale[s], inv = 1.0f / sc;
  const float z = (float)zp[s], lo = (float)qmin, hi = (float)qmax;
  const float mlo = relu ? 0.f : -INFINITY;
  const int8_t* as = a + (int64_t)s * a_ss;
  const int8_t* bs = b + (int64_t)s * b_ss;
  const bool vec = (n & 15) == 0 && ((a_ss | b_ss) & 15) == 0 && (y_ss & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(q8)) & 15) == 0;
  if (vec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n / 16; i += (int64_t)gridDim.x * 256) {
      const v4i_q8 av = reinterpret_cast<const v4i_q8*>(as)[i], bv = reinterpret_cast<const v4i_q8*>(bs)[i];
      v4i_q8 qv;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        v4f o;
        uint32_t pk = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float x = (float)(int8_t)(av[d] >> (8 * k)) * fa + (float)(int8_t)(bv[d] >> (8 * k)) * fb;
          const float m = fmaxf(fminf(fmaxf(rintf(x * inv) + z, lo), hi) - z, mlo);
          o[k] = m * sc;
          pk |= ((uint32_t)(int)m & 0xffu) << (8 * k);
        }
        qv[d] = (int)pk;
        if (y) reinterpret_cast<v4f*>(y + (int64_t)s * y_ss)[i * 4 + d] = o;
      }
      if (q8) reinterpret_cast<v4i_q8*>(q8 + (int64_t)s * n)[i] = qv;
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n; i += (int64_t)gridDim.x * 256) {
    const float x = (float)as[i] * fa + (float)bs[i] * fb;
    const float m = fmaxf(fminf(fmaxf(rintf(x * inv) + z, lo), hi) - z, mlo);
    if (y) y[(int64_t)s * y_ss + i] = m * sc;
    if (q8) q8[(int64_t)s * n + i] = (int8_t)(int)m;
  }
}

QBNN_EXPORT int qbnn_fake_quant_add_q8_mc(const int8_t* a, int64_t a_ss, const float* s_a, const int8_t* b, int64_t b_ss, const float* s_b, float* y, int64_t y_ss,
                                          int64_t n, const float* scale, const int32_t* zero_point, int32_t qmin, int32_t qmax, int32_t relu, int8_t* q8_out,
                                          int32_t n_samples, void* stream) {
  if (!a || !b || !s_a || !s_b || (!y && !q8_out) || !scale || !zero_point || n <= 0 || n_samples <= 0 || qmax <= qmin)
    return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_fake_quant_add_q8_mc: bad argument");
  if (q8_out && qmax - qmin > 127) return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_fake_quant_add_q8_mc: the int8 output (q - z) takes grids of at most 128 steps (qmax - qmin <= 127)");
  const int64_t want = (n / 16 + 255) / 256;
  const int blocks = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  hipLaunchKernelGGL(fake_quant_add_q8_kernel, dim3(blocks, n_samples), dim3(256), 0, (hipStream_t)stream, a, a_ss, s_a, b, b_ss, s_b, y, y_ss, n, scale, zero_point,
                     qmin, qmax, relu, q8_out);
  return qbnn_check_launch_msg("qbnn_fake_quant_add_q8_mc");
}

QBNN_EXPORT int qbnn_fake_quant_f32_mc(const float* x, int64_t x_ss, float* y, int64_t y_ss, int64_t n, const float* scale,
                                       const int32_t* zero_point, int32_t qparam_stride, int32_t qmin, int32_t qmax,
                                       int32_t n_samples, void* stream) {
  if (!x || !y || !scale || !zero_point || n <= 0 || n_samples <= 0 || qmax <= qmin)
    return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_fake_quant_f32_mc: bad argument");
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(fake_quant_f32_kernel, dim3(blocks, n_samples), dim3(256), 0, (hipStream_t)stream, x, x_ss, y, y_ss, n, scale,
                     zero_point, qparam_stride, qmin, qmax);
  return qbnn_check_launch_msg("qbnn_fake_quant_f32_mc");
}


// =====================================================================================
// QAT convs on the INT8 matrix pipe (round 5).  Both operands of a QAT conv are fake-quantised tensors -- integers on a per-sample grid:
//   X = (q_x - z_x) s_x[s], |q_x - z_x| <= 127 (7-bit activations; ReLU and max-pooling keep the grid);  W = (q_w - z_w) s_w[s], q_w an int8.
// So conv(X, W) = s_x s_w sum (q_x - z_x)(q_w - z_w): an exact integer sum on v_mfma_i32_32x32x32_i8 (the weight zero point through the window
// sum R = sum (q_x - z_x): sum m_x (q_w - z_w) = acc - z_w R), scaled once in fp64 and rounded to fp32 -- against the fp64 sum of the fp32-rounded
// operands (conv2d_f32_vec_kernel<.., ACC64>) it differs by the operands' own rounding, <= 1.2e-7 relative.
//   qbnn_grid_to_i8_mc : fp32 grid tensor -> int8  m = rne(x / s[s]) (+ z[s] for weights: the raw q_w), one pass per tensor
//   qbnn_conv2d_q8_f32_mc: implicit GEMM of conv_generic_mfma_i8_kernel's shape (64 pixels x 64 channels per workgroup, K in 32-byte chunks gathered
//                          in 4-byte or 1-byte units) with conv2d_f32_vec_kernel's fused tail (Z / c, + bias, bn, ReLU) and min / max partials.
// =====================================================================================
__global__ __launch_bounds__(256) void grid_to_i8_kernel(const float* __restrict__ x, int64_t x_ss, int64_t n, const float* __restrict__ scale,
                                                         const int* __restrict__ zp, int8_t* __restrict__ out) {
  const int s = blockIdx.y;
  const float inv = 1.0f / scale[s];
  const float z = zp ? (float)zp[s] : 0.f;
  const float* xs = x + (int64_t)s * x_ss;
  int8_t* os = out + (int64_t)s * n;
  if ((n & 3) == 0 && (x_ss & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += (int64_t)gridDim.x * 256) {
      const v4f v = reinterpret_cast<const v4f*>(xs)[i];
      uint32_t pk = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int q = (int)fminf(fmaxf(rintf(v[k] * inv) + z, -128.f), 127.f);
        pk |= ((uint32_t)q & 0xffu) << (8 * k);
      }
      reinterpret_cast<uint32_t*>(os)[i] = pk;
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    os[i] = (int8_t)(int)fminf(fmaxf(rintf(xs[i] * inv) + z, -128.f), 127.f);
}

QBNN_EXPORT int qbnn_grid_to_i8_mc(const float* x, int64_t x_ss, int64_t n, const float* scale, const int32_t* zero_point, int8_t* out,
                                   int32_t n_samples, void* stream) {
  if (!x || !scale || !out || n <= 0 || n_samples <= 0) return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_grid_to_i8_mc: bad argument");
  const int blocks = (int)((n / 4 + 255) / 256 < 2048 ? (n / 4 + 255) / 256 + 1 : 2048);
  hipLaunchKernelGGL(grid_to_i8_kernel, dim3(blocks, n_samples), dim3(256), 0, (hipStream_t)stream, x, x_ss, n, scale, zero_point, out);
  return qbnn_check_launch_msg("qbnn_grid_to_i8_mc");
}

// (ConvQ8Args and the int vector types: qbnn_q8.h, shared with the LDS-tiled form of the 3 x 3 convs in qbnn_q8t.hip)

template <int GB>
__global__ __launch_bounds__(256) void conv2d_q8_kernel(const ConvQ8Args a) {
  constexpr int LD = 48;
  __shared__ __attribute__((aligned(16))) uint8_t As[64 * LD];      // weights [n][k]
  __shared__ __attribute__((aligned(16))) uint8_t Bs[64 * LD];      // pixels  [p][k]
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int s = blockIdx.z;
  const int p0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int npix = a.B * a.Ho * a.Wo;
  const int K = a.KS * a.KS * a.Cin;
  const int8_t* xs = a.x + (int64_t)s * a.x_ss;
  const int8_t* ws = a.w + (int64_t)s * a.w_ss;
  const int row = tid >> 2, kb = (tid & 3) * 8;
  const int p = p0 + row, n = n0 + row;
  int pb = -1, ih0 = 0, iw0 = 0;
  if (p < npix) { pb = p / (a.Ho * a.Wo); const int rem = p - pb * a.Ho * a.Wo; ih0 = (rem / a.Wo) * a.stride - a.pad; iw0 = (rem % a.Wo) * a.stride - a.pad; }
  const int64_t xbase = (int64_t)(pb < 0 ? 0 : pb) * a.H * a.W * a.Cin;
  const int8_t* wrow = ws + (int64_t)(n < a.Cout ? n : 0) * K;
  auto gather = [&](int k0, uint32_t (&xv)[2], uint32_t (&wv)[2]) {
    int kk = k0 + kb;
    int tap = kk / a.Cin, c = kk - tap * a.Cin;
    int kh = tap / a.KS, kw = tap - kh * a.KS;
    xv[0] = xv[1] = wv[0] = wv[1] = 0u;
#pragma unroll
    for (int j = 0; j < 8 / GB; ++j, kk += GB) {
      uint32_t xb = 0u, wb = 0u;
      if (kk < K) {
        if (pb >= 0) {
          const int ih = ih0 + kh, iw = iw0 + kw;
          if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W) {      // a tap outside the map: m_x = 0 (the zero point itself)
            const int8_t* src = xs + xbase + ((int64_t)ih * a.W + iw) * a.Cin + c;
            if constexpr (GB == 4) xb = *reinterpret_cast<const uint32_t*>(src);
            else xb = (uint32_t)(uint8_t)*src;
          }
        }
        if (n < a.Cout) {
          if constexpr (GB == 4) wb = *reinterpret_cast<const uint32_t*>(wrow + kk);
          else wb = (uint32_t)(uint8_t)wrow[kk];
        }
      }
      constexpr int PER = 4 / GB;
      xv[j / PER] |= xb << (8 * GB * (j % PER));
      wv[j / PER] |= wb << (8 * GB * (j % PER));
      c += GB;
      if (c == a.Cin) { c = 0; if (++kw == a.KS) { kw = 0; ++kh; } }
    }
  };
  v16i_q8 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0;
  int rsum = 0;
  uint32_t xv[2], wv[2];
  gather(0, xv, wv);
  for (int k0 = 0; k0 < K; k0 += 32) {
    *reinterpret_cast<v2i_q8*>(&Bs[row * LD + kb]) = v2i_q8{(int)xv[0], (int)xv[1]};
    *reinterpret_cast<v2i_q8*>(&As[row * LD + kb]) = v2i_q8{(int)wv[0], (int)wv[1]};
    __syncthreads();
    if (k0 + 32 < K) gather(k0 + 32, xv, wv);
    const v4i_q8 av = *reinterpret_cast<const v4i_q8*>(&As[(wn * 32 + (lane & 31)) * LD + 16 * (lane >> 5)]);
    const v4i_q8 bv = *reinterpret_cast<const v4i_q8*>(&Bs[(wm * 32 + (lane & 31)) * LD + 16 * (lane >> 5)]);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) rsum = __builtin_amdgcn_sdot4(bv[i], 0x01010101, rsum, false);
    __syncthreads();
  }
  const int R = rsum + __shfl_xor(rsum, 32);              // this lane's pixel (lane & 31): sum of m_x over its window
  const int po = p0 + wm * 32 + (lane & 31);
  float vmin = INFINITY, vmax = -INFINITY;
  if (po < npix) {
    const double sp = (double)a.s_x[s] * (double)a.s_w[s];
    const int zwr = a.z_w[s] * R;
    float* yp = a.y + (int64_t)s * a.y_ss + (int64_t)po * a.Cout;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int no = n0 + wn * 32 + 8 * g + 4 * (lane >> 5) + i;
        if (no < a.Cout) {
          float v = (float)((double)(acc[4 * g + i] - zwr) * sp);
          if (a.div) v = v / a.div[no];
          if (a.bias) v = v + a.bias[no];
          if (a.alpha) v = v * a.alpha[no];
          if (a.beta) v = v + a.beta[no];
          if (a.relu) v = fmaxf(v, 0.f);
          yp[no] = v; vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
        }
      }
  }
  if (a.mm_partials) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { vmin = fminf(vmin, __shfl_xor(vmin, o)); vmax = fmaxf(vmax, __shfl_xor(vmax, o)); }
    if (lane == 0) { red[2 * wave] = vmin; red[2 * wave + 1] = vmax; }
    __syncthreads();
    if (tid == 0) {
      const int64_t slot = ((int64_t)s * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      a.mm_partials[2 * slot] = fminf(fminf(red[0], red[2]), fminf(red[4], red[6]));
      a.mm_partials[2 * slot + 1] = fmaxf(fmaxf(red[1], red[3]), fmaxf(red[5], red[7]));
    }
  }
}


// The Cin % 4 == 0 form (every conv but layers.0): 64-byte K chunks -- each thread moves 16 bytes per operand and chunk, two MFMAs per wave and barrier
// pair -- and, as in conv2d_f32_vec_kernel<.., VEC = 2>, a row keeps the offset of its window's tap (0, 0) and one validity bit per tap
// (KS KS <= 32), so a 4-byte unit costs one offset add and a bit test instead of two bounds compares and a 64-bit address product.
// Workgroup tile = PT pixels x NT channels (PT NT = 4096; 4 waves of 32 x 32): 64 x 64, or 128 x 32 where Cout <= 32 would leave half of the waves
// multiplying padding (the 24-channel layers: five of the twenty convs and the largest by pixels).
template <int PT, int NT>
__global__ __launch_bounds__(256) void conv2d_q8v_kernel(const ConvQ8Args a) {
  static_assert(PT * NT == 4096 && PT % 64 == 0 && NT % 32 == 0, "4 waves of 32 x 32");
  constexpr int LD = 80;                                              // 64 + 16: conflict-free ds_read_b128 fragments (tools/lds_conflicts.py)
  constexpr int WN = NT / 32, XP = PT / 64;
  __shared__ __attribute__((aligned(16))) uint8_t As[NT * LD];
  __shared__ __attribute__((aligned(16))) uint8_t Bs[PT * LD];
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int s = blockIdx.z;
  const int p0 = blockIdx.x * PT, n0 = blockIdx.y * NT;
  const int npix = a.B * a.Ho * a.Wo;
  const int K = a.KS * a.KS * a.Cin;
  const int8_t* xs = a.x + (int64_t)s * a.x_ss;
  const int8_t* ws = a.w + (int64_t)s * a.w_ss;
  const int row = tid >> 2, kb = (tid & 3) * 16;
  int roff[XP];
  uint32_t rmask[XP];
#pragma unroll
  for (int j = 0; j < XP; ++j) {
    const int p = p0 + row + 64 * j;
    roff[j] = 0; rmask[j] = 0;
    if (p < npix) {
      const int pb = p / (a.Ho * a.Wo), rem = p - pb * a.Ho * a.Wo;
      const int ih0 = (rem / a.Wo) * a.stride - a.pad, iw0 = (rem % a.Wo) * a.stride - a.pad;
      roff[j] = ((pb * a.H + ih0) * a.W + iw0) * a.Cin;                 // (a sample's input is below 2^31 bytes)
      uint32_t m = 0;
      for (int kh = 0; kh < a.KS; ++kh)
        for (int kw = 0; kw < a.KS; ++kw)
          if ((unsigned)(ih0 + kh) < (unsigned)a.H && (unsigned)(iw0 + kw) < (unsigned)a.W) m |= 1u << (kh * a.KS + kw);
      rmask[j] = m;
    }
  }
  const int n = n0 + row;
  const bool wrow = row < NT, wok = wrow && n < a.Cout;
  const int8_t* wr = ws + (int64_t)(wok ? n : 0) * K;
  // this thread's first unit of the chunk: k = k0 + kb in tap (gkh, gkw) at channel gc; advanced by 64 per chunk without divisions
  int gkh, gkw, gc;
  { const int tap = kb / a.Cin; gc = kb - tap * a.Cin; gkh = tap / a.KS; gkw = tap - gkh * a.KS; }
  auto gather = [&](int k0, v4i_q8 (&xv)[XP], v4i_q8& wv) {
    int kh = gkh, kw = gkw, c = gc;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kk = k0 + kb + 4 * u;
      const bool kok = kk < K;
      const int tap = kh * a.KS + kw, toff = (kh * a.W + kw) * a.Cin + c;
#pragma unroll
      for (int j = 0; j < XP; ++j) xv[j][u] = (kok && ((rmask[j] >> tap) & 1u)) ? *reinterpret_cast<const int*>(xs + roff[j] + toff) : 0;
      wv[u] = (kok && wok) ? *reinterpret_cast<const int*>(wr + kk) : 0;
      c += 4;
      if (c >= a.Cin) { c -= a.Cin; if (++kw == a.KS) { kw = 0; ++kh; } }
    }
    int cc = gc + 64;                                                 // the next chunk's first unit
    while (cc >= a.Cin) { cc -= a.Cin; if (++gkw == a.KS) { gkw = 0; ++gkh; } }
    gc = cc;
  };
  v16i_q8 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0;
  int rsum = 0;
  v4i_q8 xv[XP], wv;
  gather(0, xv, wv);
  for (int k0 = 0; k0 < K; k0 += 64) {
#pragma unroll
    for (int j = 0; j < XP; ++j) *reinterpret_cast<v4i_q8*>(&Bs[(row + 64 * j) * LD + kb]) = xv[j];
    if (wrow) *reinterpret_cast<v4i_q8*>(&As[row * LD + kb]) = wv;
    __syncthreads();
    if (k0 + 64 < K) gather(k0 + 64, xv, wv);
#pragma unroll
    for (int hk = 0; hk < 2; ++hk) {
      const v4i_q8 av = *reinterpret_cast<const v4i_q8*>(&As[(wn * 32 + (lane & 31)) * LD + 32 * hk + 16 * (lane >> 5)]);
      const v4i_q8 bv = *reinterpret_cast<const v4i_q8*>(&Bs[(wm * 32 + (lane & 31)) * LD + 32 * hk + 16 * (lane >> 5)]);
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) rsum = __builtin_amdgcn_sdot4(bv[i], 0x01010101, rsum, false);
    }
    __syncthreads();
  }
  const int R = rsum + __shfl_xor(rsum, 32);
  const int po = p0 + wm * 32 + (lane & 31);
  float vmin = INFINITY, vmax = -INFINITY;
  if (po < npix) {
    const double sp = (double)a.s_x[s] * (double)a.s_w[s];
    const int zwr = a.z_w[s] * R;
    float* yp = a.y + (int64_t)s * a.y_ss + (int64_t)po * a.Cout;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nb = n0 + wn * 32 + 8 * g + 4 * (lane >> 5);
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int no = nb + i, nc = no < a.Cout ? no : 0;
        float t = (float)((double)(acc[4 * g + i] - zwr) * sp);
        if (a.div) t = t / a.div[nc];
        if (a.bias) t = t + a.bias[nc];
        if (a.alpha) t = t * a.alpha[nc];
        if (a.beta) t = t + a.beta[nc];
        if (a.relu) t = fmaxf(t, 0.f);
        v[i] = t;
        if (no < a.Cout) { vmin = fminf(vmin, t); vmax = fmaxf(vmax, t); }
      }
      if (nb + 3 < a.Cout && (a.Cout & 3) == 0) *reinterpret_cast<v4f*>(yp + nb) = v4f{v[0], v[1], v[2], v[3]};
      else
        for (int i = 0; i < 4; ++i)
          if (nb + i < a.Cout) yp[nb + i] = v[i];
    }
  }
  if (a.mm_partials) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { vmin = fminf(vmin, __shfl_xor(vmin, o)); vmax = fmaxf(vmax, __shfl_xor(vmax, o)); }
    if (lane == 0) { red[2 * wave] = vmin; red[2 * wave + 1] = vmax; }
    __syncthreads();
    if (tid == 0) {
      const int64_t slot = ((int64_t)s * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      a.mm_partials[2 * slot] = fminf(fminf(red[0], red[2]), fminf(red[4], red[6]));
      a.mm_partials[2 * slot + 1] = fmaxf(fmaxf(red[1], red[3]), fmaxf(red[5], red[7]));
    }
  }
}

// workgroups per sample of qbnn_conv2d_q8_f32_mc = length of one sample's row of `minmax_partials`
static bool conv_q8_narrow(int Cin, int Cout, int ksize) { return Cout <= 32 && (Cin % 4) == 0 && ksize * ksize <= 32; }
QBNN_EXPORT int32_t qbnn_conv2d_q8_blocks(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride, int32_t pad) {
  const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
  if (B <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0) return 0;
  const int64_t npix = (int64_t)B * Ho * Wo;
  if (const int tiled = qbnn_conv_q8t_blocks(B, H, W, Cin, Cout, ksize, stride, pad)) return tiled;      // the ResNet's 3 x 3 convs: LDS-tiled (qbnn_q8t.hip)
  return conv_q8_narrow(Cin, Cout, ksize) ? (int32_t)(((npix + 127) / 128) * ((Cout + 31) / 32)) : (int32_t)(((npix + 63) / 64) * ((Cout + 63) / 64));
}

QBNN_EXPORT int qbnn_conv2d_q8_f32_mc(const int8_t* x, int64_t x_ss, const int8_t* w, int64_t w_ss, const float* s_x, const float* s_w,
                                      const int32_t* z_w, const float* div, const float* bias, const float* alpha, const float* beta, float* y,
                                      int64_t y_ss, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride,
                                      int32_t pad, int32_t relu, int32_t n_samples, float* minmax_partials, void* stream) {
  if (!x || !w || !s_x || !s_w || !z_w || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || ksize <= 0 || stride <= 0 || pad < 0 || n_samples <= 0)
    return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_conv2d_q8_f32_mc: bad argument");
  ConvQ8Args a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.x_ss = x_ss; a.w = w; a.w_ss = w_ss; a.s_x = s_x; a.s_w = s_w; a.z_w = z_w; a.y = y; a.y_ss = y_ss;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KS = ksize; a.stride = stride; a.pad = pad; a.relu = relu & 1;
  a.Ho = (H + 2 * pad - ksize) / stride + 1; a.Wo = (W + 2 * pad - ksize) / stride + 1;
  a.div = div; a.bias = bias; a.alpha = alpha; a.beta = beta; a.mm_partials = minmax_partials;
  if (a.Ho <= 0 || a.Wo <= 0) return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_conv2d_q8_f32_mc: empty output");
  if ((int64_t)ksize * ksize * Cin * 127 * 128 >= (1ll << 31)) return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_conv2d_q8_f32_mc: K too large for int32 sums");
  if (qbnn_conv_q8t_blocks(B, H, W, Cin, Cout, ksize, stride, pad) > 0) {
    // (the grid qbnn_conv2d_q8_blocks promises the observer is the tiled form's: no other form may serve this geometry)
    if (!qbnn_conv_q8t_aligned(a))
      return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_conv2d_q8_f32_mc: the tiled 3 x 3 form takes 16-byte aligned operands, outputs and per-channel parameters (8-byte operands at Cin = 24)");
    return qbnn_launch_conv_q8t(a, n_samples, (hipStream_t)stream);
  }
  const int64_t npix = (int64_t)B * a.Ho * a.Wo;
  const bool narrow = conv_q8_narrow(Cin, Cout, ksize);      // (every form launches the grid qbnn_conv2d_q8_blocks promises the observer)
  const dim3 grid(narrow ? (unsigned)((npix + 127) / 128) : (unsigned)((npix + 63) / 64), narrow ? (unsigned)((Cout + 31) / 32) : (unsigned)((Cout + 63) / 64), (unsigned)n_samples);
  const bool u4 = (Cin % 4) == 0 && (x_ss % 4) == 0 && (w_ss % 4) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 3) == 0;
  static const bool wide = [] { const char* e = getenv("QBNN_Q8_WIDE"); return !(e && e[0] == '0'); }();
  // (the 16-byte output stores are taken only when Cout % 4 == 0 -- a head Linear(100, 1) / Linear(500, 10) stores scalars, whatever y_ss = Cout B is)
  const bool y16 = (Cout & 3) != 0 || ((reinterpret_cast<uintptr_t>(y) & 15) == 0 && (y_ss % 4) == 0);
  const bool v_ok = u4 && ksize * ksize <= 32 && y16 && (int64_t)B * H * W * Cin < (1ll << 31);
  if (narrow) {
    if (!v_ok) return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_conv2d_q8_f32_mc: the 128 x 32 form (Cout <= 32, Cin % 4 == 0) takes 4-byte aligned operands, fewer than 2^31 input elements and, when Cout % 4 == 0, 16-byte aligned outputs");
    hipLaunchKernelGGL((conv2d_q8v_kernel<128, 32>), grid, dim3(256), 0, (hipStream_t)stream, a);
  } else if (v_ok && wide) hipLaunchKernelGGL((conv2d_q8v_kernel<64, 64>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (u4) hipLaunchKernelGGL(conv2d_q8_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(conv2d_q8_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a);
  return qbnn_check_launch_msg("qbnn_conv2d_q8_f32_mc");
}

// W[s][i] = mu[s][i] + eps(s, i) * sigma[s][i] with per-sample mu / sigma (the QAT weight pipeline quantises them with
// per-sample qparams); mu NULL -> the noise term alone (conv_qat.py:45: mul_noise.mul(noise, std)).  Same Philox stream as
// qbnn_sample_weights_f32: ctr = {i >> 2, layer_id, sample_begin + s, 0}, element i & 3; eps_in [S][n] overrides it.
__global__ __launch_bounds__(256) void sample_weights_f32_strided_kernel(const float* __restrict__ mu, int64_t mu_ss, const float* __restrict__ sigma,
                                                                          int64_t sigma_ss, int64_t n, uint32_t seed_lo, uint32_t seed_hi,
                                                                          uint32_t layer_id, uint32_t sample_begin,
                                                                          const float* __restrict__ eps_in, float* __restrict__ w,
                                                                          const uint32_t* __restrict__ nd) {
  if (nd) { seed_lo = nd[0]; seed_hi = nd[1]; sample_begin = nd[2]; }      // captured-graph mode: the seed lives in device memory
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g * 4 >= n) return;
  const int s = blockIdx.y;
  float e[4];
  if (eps_in) {
    for (int j = 0; j < 4; ++j) e[j] = (g * 4 + j < n) ? eps_in[(int64_t)s * n + g * 4 + j] : 0.f;
  } else {
    qbnn::normal4(qbnn::philox4x32_10((uint32_t)g, layer_id, sample_begin + s, 0u, seed_lo, seed_hi), e);
  }
  if ((n & 3) == 0 && ((mu_ss | sigma_ss) & 3) == 0 &&
      ((reinterpret_cast<uintptr_t>(mu) | reinterpret_cast<uintptr_t>(sigma) | reinterpret_cast<uintptr_t>(w)) & 15) == 0) {
    const float4 s4 = reinterpret_cast<const float4*>(sigma + (int64_t)s * sigma_ss)[g];
    const float4 m4 = mu ? reinterpret_cast<const float4*>(mu + (int64_t)s * mu_ss)[g] : float4{0.f, 0.f, 0.f, 0.f};
    float4 o;
    { const float t = e[0] * s4.x; o.x = mu ? m4.x + t : t; }
    { const float t = e[1] * s4.y; o.y = mu ? m4.y + t : t; }
    { const float t = e[2] * s4.z; o.z = mu ? m4.z + t : t; }
    { const float t = e[3] * s4.w; o.w = mu ? m4.w + t : t; }
    reinterpret_cast<float4*>(w + (int64_t)s * n)[g] = o;
    return;
  }
  for (int j = 0; j < 4; ++j) {
    const int64_t i = g * 4 + j;
    if (i < n) {
      const float t = e[j] * sigma[(int64_t)s * sigma_ss + i];
      w[(int64_t)s * n + i] = mu ? mu[(int64_t)s * mu_ss + i] + t : t;
    }
  }
}

QBNN_EXPORT int qbnn_sample_weights_f32_strided(const float* mu, int64_t mu_ss, const float* sigma, int64_t sigma_ss, int64_t n,
                                                uint64_t seed, uint32_t layer_id, uint32_t sample_begin, int32_t n_samples,
                                                const float* eps_in, float* w_out, void* stream) {
  if (!sigma || !w_out || n <= 0 || n_samples <= 0) return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_sample_weights_f32_strided: bad argument");
  const int64_t groups = (n + 3) / 4;
  hipLaunchKernelGGL(sample_weights_f32_strided_kernel, dim3((unsigned)((groups + 255) / 256), n_samples), dim3(256), 0, (hipStream_t)stream,
                     mu, mu_ss, sigma, sigma_ss, n, (uint32_t)seed, (uint32_t)(seed >> 32), layer_id, sample_begin, eps_in, w_out, qbnn_noise_dev());
  return qbnn_check_launch_msg("qbnn_sample_weights_f32_strided");
}

// As qbnn_sample_weights_f32_strided for a conv weight, but written in [Cout][KH][KW][Cin] order (the K axis of the implicit
// GEMM contiguous) while the noise index stays the reference's [Cout][Cin][KH][KW] element index.
__global__ __launch_bounds__(256) void sample_weights_f32_ohwi_kernel(const float* __restrict__ mu, int64_t mu_ss, const float* __restrict__ sigma,
                                                                       int64_t sigma_ss, int Cout, int Cin, int KS, uint32_t seed_lo,
                                                                       uint32_t seed_hi, uint32_t layer_id, uint32_t sample_begin,
                                                                       const float* __restrict__ eps_in, float* __restrict__ w,
                                                                       const uint32_t* __restrict__ nd) {
  if (nd) { seed_lo = nd[0]; seed_hi = nd[1]; sample_begin = nd[2]; }      // captured-graph mode: the seed lives in device memory
  const int64_t n = (int64_t)Cout * Cin * KS * KS;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g * 4 >= n) return;
  const int s = blockIdx.y;
  float e[4];
  if (eps_in) {
    for (int j = 0; j < 4; ++j) e[j] = (g * 4 + j < n) ? eps_in[(int64_t)s * n + g * 4 + j] : 0.f;
  } else {
    qbnn::normal4(qbnn::philox4x32_10((uint32_t)g, layer_id, sample_begin + s, 0u, seed_lo, seed_hi), e);
  }
  for (int j = 0; j < 4; ++j) {
    const int64_t i = g * 4 + j;
    if (i < n) {
      const int kw = (int)(i % KS);
      int64_t r = i / KS;
      const int kh = (int)(r % KS); r /= KS;
      const int c = (int)(r % Cin);
      const int o = (int)(r / Cin);
      const float t = e[j] * sigma[(int64_t)s * sigma_ss + i];
      w[(int64_t)s * n + (((int64_t)o * KS + kh) * KS + kw) * Cin + c] = mu ? mu[(int64_t)s * mu_ss + i] + t : t;
    }
  }
}

QBNN_EXPORT int qbnn_sample_weights_f32_ohwi(const float* mu, int64_t mu_ss, const float* sigma, int64_t sigma_ss, int32_t Cout, int32_t Cin,
                                             int32_t ksize, uint64_t seed, uint32_t layer_id, uint32_t sample_begin, int32_t n_samples,
                                             const float* eps_in, float* w_out, void* stream) {
  if (!sigma || !w_out || Cout <= 0 || Cin <= 0 || ksize <= 0 || n_samples <= 0)
    return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_sample_weights_f32_ohwi: bad argument");
  const int64_t groups = ((int64_t)Cout * Cin * ksize * ksize + 3) / 4;
  hipLaunchKernelGGL(sample_weights_f32_ohwi_kernel, dim3((unsigned)((groups + 255) / 256), n_samples), dim3(256), 0, (hipStream_t)stream,
                     mu, mu_ss, sigma, sigma_ss, Cout, Cin, ksize, (uint32_t)seed, (uint32_t)(seed >> 32), layer_id, sample_begin, eps_in, w_out, qbnn_noise_dev());
  return qbnn_check_launch_msg("qbnn_sample_weights_f32_ohwi");
}

// =====================================================================================
// The QAT weight pipelines of ALL stochastic layers in four launches (round 6).
//
// conv_qat.py:26-49 / linear_qat.py:18-41 in eval, per layer:  w = FQ_w(mu c), s = FQ_s(softplus(rho) c), t = FQ_m(eps * s), W = FQ_a(w + t), every
// FQ a live MovingAverageMinMax observer + fake_quantize.  Layer by layer that is ~15 launches of a few microseconds each (min / max, observer scan,
// fake-quantise x 4; the noise draw; the sum; the int8 view) -- ~300 per forward of the ResNet, 1.6 of the 4.9 ms a 10-sample pass spent on the GPU.
// Nothing here depends on an activation, and the only cross-element dependences are the observers' (min, max): so
//   stage 0  draws eps, fake-quantises sigma, leaves per-workgroup (min, max) of t_pre = eps * s                      -> partials `pm`
//   stage 1  recomputes t_pre, t = FQ_m(t_pre) with FQ_m's per-sample qparams, w = FQ_w(mu), (min, max) of w + t      -> partials `pa`
//   stage 2  recomputes w + t, W = FQ_a(w + t): writes W (fp32), its raw integers (int8) and FQ_a's per-sample (scale, zero point)
//   commit   advances the four observers' states (the only writer of a state; stages 0 - 2 only read them)
// for every layer and MC sample at once.  Each workgroup derives the qparams it needs itself: the EMA recurrence over samples 0..s is a few flops --
// from the constant (min, max) of mu c / sigma c for FQ_w / FQ_s (they see the same tensor S times), from the <= 64 partial pairs per sample of the
// previous stage for FQ_m / FQ_a.  Recomputing eps costs three Philox + Box-Muller evaluations per weight instead of three tensor round trips.
// Same arithmetic, step for step, as minmax_f32_kernel / observer_scan_kernel / fake_quant_f32_kernel / sample_weights_f32_(ohwi|strided)_kernel /
// affine_f32 / grid_to_i8: the results are bit-identical (tests: QBNN_QAT_WBATCH=0 switch).
// =====================================================================================
struct QatWLayer {            // mirrors qbnn_qat_wlayer (include/qbnn.h)
  const float* mu; const float* sg;
  float* st_w; float* st_s; float* st_m; float* st_a;
  float cmm[4];
  int32_t n, Cout, Cin, KS;
  uint32_t layer_id;
  int32_t qmin, qmax;
  int32_t blk0, nblk;
  float* pm; float* pa;
  float* W; int8_t* q8; float* scale; int32_t* zp;
};
static_assert(sizeof(QatWLayer) == sizeof(qbnn_qat_wlayer), "qbnn_qat_wlayer layout");

struct ObsQ { float sc, inv, z; };
__device__ __forceinline__ void obs_step(float& mn_s, float& mx_s, bool& init, float mn, float mx, float c) {
  if (!init) { mn_s = mn; mx_s = mx; init = true; }
  else { mn_s = mn_s + c * (mn - mn_s); mx_s = mx_s + c * (mx - mx_s); }
}
__device__ __forceinline__ ObsQ obs_qparams(float mn_s, float mx_s, int qmin, int qmax) {      // observer_scan_kernel's calculate_qparams
  const float lo = fminf(mn_s, 0.f), hi = fmaxf(mx_s, 0.f);
  float sc = (hi - lo) / (float)(qmax - qmin);
  sc = fmaxf(sc, 1.1920928955078125e-07f);
  float z = (float)qmin - rintf(lo / sc);
  z = fminf(fmaxf(z, (float)qmin), (float)qmax);
  return ObsQ{sc, 1.0f / sc, z};
}
// the observer's qparams for sample s after having seen the SAME (mn, mx) s + 1 times, starting from `state`
__device__ __forceinline__ ObsQ obs_const(const float* state, float mn, float mx, int s, float c, int qmin, int qmax) {
  float a = state[0], b = state[1];
  bool init = state[2] != 0.f;
  for (int i = 0; i <= s; ++i) obs_step(a, b, init, mn, mx, c);
  return obs_qparams(a, b, qmin, qmax);
}
// ... after having seen samples 0..s whose (min, max) come from [S][nblk][2] partials; `red` = 2 * 64 floats of LDS.  All threads return the same value.
__device__ __forceinline__ ObsQ obs_partials(const float* state, const float* partials, int nblk, int s, float c, int qmin, int qmax, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();                                  // `red` may still be read from a previous call
  for (int i = wave; i <= s; i += 4) {
    float mn = INFINITY, mx = -INFINITY;
    for (int k = lane; k < nblk; k += 64) { mn = fminf(mn, partials[((int64_t)i * nblk + k) * 2]); mx = fmaxf(mx, partials[((int64_t)i * nblk + k) * 2 + 1]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    if (lane == 0) { red[i] = mn; red[64 + i] = mx; }
  }
  __syncthreads();
  float a = state[0], b = state[1];
  bool init = state[2] != 0.f;
  for (int i = 0; i <= s; ++i) obs_step(a, b, init, red[i], red[64 + i], c);
  return obs_qparams(a, b, qmin, qmax);
}
__device__ __forceinline__ float fq_q(float x, const ObsQ& q, float lo, float hi) { return fminf(fmaxf(rintf(x * q.inv) + q.z, lo), hi); }

template <int STAGE>
__global__ __launch_bounds__(256) void qat_weights_kernel(const QatWLayer* __restrict__ layers, int n_layers, float avg_const, uint32_t seed_lo, uint32_t seed_hi,
                                                          uint32_t sample_begin, const uint32_t* __restrict__ nd) {
  __shared__ float red[128];
  __shared__ float bred[8];
  if (nd) { seed_lo = nd[0]; seed_hi = nd[1]; sample_begin = nd[2]; }
  int li = 0;
  for (int i = 1; i < n_layers; ++i) li = (int)blockIdx.x >= layers[i].blk0 ? i : li;
  const QatWLayer L = layers[li];
  const int b = blockIdx.x - L.blk0, s = blockIdx.y, tid = threadIdx.x;
  const float lo = (float)L.qmin, hi = (float)L.qmax;
  const ObsQ qs = obs_const(L.st_s, L.cmm[2], L.cmm[3], s, avg_const, L.qmin, L.qmax);
  ObsQ qw{}, qm{}, qa{};
  if (STAGE >= 1) {
    qw = obs_const(L.st_w, L.cmm[0], L.cmm[1], s, avg_const, L.qmin, L.qmax);
    qm = obs_partials(L.st_m, L.pm, L.nblk, s, avg_const, L.qmin, L.qmax, red);
  }
  if (STAGE >= 2) qa = obs_partials(L.st_a, L.pa, L.nblk, s, avg_const, L.qmin, L.qmax, red);
  const int64_t n = L.n, groups = (n + 3) / 4;
  const int64_t per = (groups + L.nblk - 1) / L.nblk, g0 = (int64_t)b * per, g1 = g0 + per < groups ? g0 + per : groups;
  float vmin = INFINITY, vmax = -INFINITY;
  for (int64_t g = g0 + tid; g < g1; g += 256) {
    float e[4];
    qbnn::normal4(qbnn::philox4x32_10((uint32_t)g, L.layer_id, sample_begin + s, 0u, seed_lo, seed_hi), e);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = g * 4 + j;
      if (i >= n) break;
      const float sgq = (fq_q(L.sg[i], qs, lo, hi) - qs.z) * qs.sc;       // s = FQ_s(sigma c)
      const float tp = e[j] * sgq;                                          // eps * s
      if (STAGE == 0) { vmin = fminf(vmin, tp); vmax = fmaxf(vmax, tp); continue; }
      uint32_t o = (uint32_t)i;                                             // the output element: [Cout][kh][kw][Cin] for a conv, the reference's order for a linear
      if (L.KS > 0) {
        // i = ((oc Cin + c) KK + t), t = kh KS + kw  ->  o = (oc KK + t) Cin + c.  32-bit unsigned divisions (n < 2^31: the launcher checks); the 64-bit
        // forms of sample_weights_f32_ohwi_kernel cost ~150 instructions each and were three quarters of this kernel
        const uint32_t kk = (uint32_t)(L.KS * L.KS), ckk = (uint32_t)L.Cin * kk, iu = (uint32_t)i;
        const uint32_t oc = iu / ckk, rem = iu - oc * ckk, c = rem / kk, t = rem - c * kk;
        o = (oc * kk + t) * (uint32_t)L.Cin + c;
      }
      const float t = (fq_q(tp, qm, lo, hi) - qm.z) * qm.sc;               // t = FQ_m(eps * s)
      const float wq = (fq_q(L.mu[o], qw, lo, hi) - qw.z) * qw.sc;         // w = FQ_w(mu c)
      const float sum = wq + t;
      if (STAGE == 1) { vmin = fminf(vmin, sum); vmax = fmaxf(vmax, sum); continue; }
      const float q = fq_q(sum, qa, lo, hi);
      L.W[(int64_t)s * n + o] = (q - qa.z) * qa.sc;
      L.q8[(int64_t)s * n + o] = (int8_t)(int)fminf(fmaxf(q, -128.f), 127.f);
    }
  }
  if (STAGE <= 1) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { vmin = fminf(vmin, __shfl_xor(vmin, o)); vmax = fmaxf(vmax, __shfl_xor(vmax, o)); }
    if ((tid & 63) == 0) { bred[2 * (tid >> 6)] = vmin; bred[2 * (tid >> 6) + 1] = vmax; }
    __syncthreads();
    if (tid == 0) {
      float* p = (STAGE == 0 ? L.pm : L.pa) + ((int64_t)s * L.nblk + b) * 2;
      p[0] = fminf(fminf(bred[0], bred[2]), fminf(bred[4], bred[6]));
      p[1] = fmaxf(fmaxf(bred[1], bred[3]), fmaxf(bred[5], bred[7]));
    }
  } else if (b == 0 && tid == 0) {
    L.scale[s] = qa.sc; L.zp[s] = (int)qa.z;
  }
}

// one workgroup per layer: the four observers after all S samples
__global__ __launch_bounds__(256) void qat_weights_commit_kernel(const QatWLayer* __restrict__ layers, int n_samples, float avg_const) {
  __shared__ float red[128];
  const QatWLayer L = layers[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float fin[4][2];
  for (int k = 0; k < 2; ++k) {             // mul_noise, add_weight: per-sample (min, max) over the partials, then the recurrence
    const float* partials = k == 0 ? L.pm : L.pa;
    const float* st = k == 0 ? L.st_m : L.st_a;
    __syncthreads();
    for (int i = wave; i < n_samples; i += 4) {
      float mn = INFINITY, mx = -INFINITY;
      for (int j = lane; j < L.nblk; j += 64) { mn = fminf(mn, partials[((int64_t)i * L.nblk + j) * 2]); mx = fmaxf(mx, partials[((int64_t)i * L.nblk + j) * 2 + 1]); }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
      if (lane == 0) { red[i] = mn; red[64 + i] = mx; }
    }
    __syncthreads();
    float a = st[0], b = st[1];
    bool init = st[2] != 0.f;
    for (int i = 0; i < n_samples; ++i) obs_step(a, b, init, red[i], red[64 + i], avg_const);
    fin[2 + k][0] = a; fin[2 + k][1] = b;
  }
  for (int k = 0; k < 2; ++k) {             // weight_fake_quant, std_fake_quant: the same tensor S times
    const float* st = k == 0 ? L.st_w : L.st_s;
    float a = st[0], b = st[1];
    bool init = st[2] != 0.f;
    for (int i = 0; i < n_samples; ++i) obs_step(a, b, init, L.cmm[2 * k], L.cmm[2 * k + 1], avg_const);
    fin[k][0] = a; fin[k][1] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* sts[4] = {L.st_w, L.st_s, L.st_m, L.st_a};
    for (int k = 0; k < 4; ++k) { sts[k][0] = fin[k][0]; sts[k][1] = fin[k][1]; sts[k][2] = 1.f; }
  }
}

QBNN_EXPORT int qbnn_qat_weights_mc(const qbnn_qat_wlayer* dev_layers, int32_t n_layers, int32_t total_blocks, float avg_const, uint64_t seed,
                                    uint32_t sample_begin, int32_t n_samples, void* stream) {
  if (!dev_layers || n_layers <= 0 || total_blocks <= 0 || n_samples <= 0 || n_samples > 64)
    return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_qat_weights_mc: bad argument (1 to 64 samples per call)");
  const QatWLayer* L = reinterpret_cast<const QatWLayer*>(dev_layers);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(total_blocks, n_samples);
  const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
  hipLaunchKernelGGL(qat_weights_kernel<0>, grid, dim3(256), 0, st, L, n_layers, avg_const, lo, hi, sample_begin, qbnn_noise_dev());
  hipLaunchKernelGGL(qat_weights_kernel<1>, grid, dim3(256), 0, st, L, n_layers, avg_const, lo, hi, sample_begin, qbnn_noise_dev());
  hipLaunchKernelGGL(qat_weights_kernel<2>, grid, dim3(256), 0, st, L, n_layers, avg_const, lo, hi, sample_begin, qbnn_noise_dev());
  hipLaunchKernelGGL(qat_weights_commit_kernel, dim3(n_layers), dim3(256), 0, st, L, n_samples, avg_const);
  return qbnn_check_launch_msg("qbnn_qat_weights_mc");
}

// The float Bayes-by-backprop draw W_s = mu + eps_s * softplus(rho) (bbb/conv.py:33-39, bbb/linear.py:42-50) of ALL layers and samples in one launch
// (round 6): qbnn_sample_weights_f32_ohwi / qbnn_sample_weights_f32 layer by layer are 21 launches of 5 - 18 us for the ResNet -- 6 % of a 10-sample pass.
// Same Philox stream (ctr = {i >> 2, layer_id, sample_begin + s, 0}, element i & 3 of the reference's element order), same two roundings (t = eps * sigma;
// mu + t), a conv's result written in [Cout][kh][kw][Cin] order: bit-identical to the per-layer calls.
struct F32WLayer {            // mirrors qbnn_f32_wlayer (include/qbnn.h)
  const float* mu; const float* sigma; float* w;
  int32_t n, Cout, Cin, KS;
  uint32_t layer_id;
  int32_t blk0, nblk;
};
static_assert(sizeof(F32WLayer) == sizeof(qbnn_f32_wlayer), "qbnn_f32_wlayer layout");

__global__ __launch_bounds__(256) void sample_weights_f32_batch_kernel(const F32WLayer* __restrict__ layers, int n_layers, uint32_t seed_lo, uint32_t seed_hi,
                                                                       uint32_t sample_begin, const uint32_t* __restrict__ nd) {
  if (nd) { seed_lo = nd[0]; seed_hi = nd[1]; sample_begin = nd[2]; }
  int li = 0;
  for (int i = 1; i < n_layers; ++i) li = (int)blockIdx.x >= layers[i].blk0 ? i : li;
  const F32WLayer L = layers[li];
  const int b = blockIdx.x - L.blk0, s = blockIdx.y;
  const uint32_t n = (uint32_t)L.n, groups = (n + 3) / 4;
  const uint32_t kk = (uint32_t)(L.KS * L.KS), ckk = (uint32_t)L.Cin * kk;
  for (uint32_t g = (uint32_t)b * 256 + threadIdx.x; g < groups; g += (uint32_t)L.nblk * 256) {
    float e[4];
    qbnn::normal4(qbnn::philox4x32_10(g, L.layer_id, sample_begin + s, 0u, seed_lo, seed_hi), e);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t i = g * 4 + j;
      if (i >= n) break;
      uint32_t o = i;
      if (L.KS > 0) {        // i = (oc Cin + c) KK + t  ->  o = (oc KK + t) Cin + c
        const uint32_t oc = i / ckk, rem = i - oc * ckk, c = rem / kk, t = rem - c * kk;
        o = (oc * kk + t) * (uint32_t)L.Cin + c;
      }
      const float t = e[j] * L.sigma[i];
      L.w[(int64_t)s * n + o] = L.mu[i] + t;
    }
  }
}

QBNN_EXPORT int qbnn_sample_weights_f32_batch(const qbnn_f32_wlayer* dev_layers, int32_t n_layers, int32_t total_blocks, uint64_t seed, uint32_t sample_begin,
                                              int32_t n_samples, void* stream) {
  if (!dev_layers || n_layers <= 0 || total_blocks <= 0 || n_samples <= 0) return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_sample_weights_f32_batch: bad argument");
  hipLaunchKernelGGL(sample_weights_f32_batch_kernel, dim3(total_blocks, n_samples), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const F32WLayer*>(dev_layers),
                     n_layers, (uint32_t)seed, (uint32_t)(seed >> 32), sample_begin, qbnn_noise_dev());
  return qbnn_check_launch_msg("qbnn_sample_weights_f32_batch");
}

QBNN_EXPORT int qbnn_conv2d_f32_mc(const float* x, int64_t x_ss, const float* w, int64_t w_ss, const float* bias, float* y,
                                   int64_t y_ss, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize,
                                   int32_t stride, int32_t pad, int32_t relu, int32_t n_samples, void* stream) {
  return qbnn_conv2d_f32_fused_mc(x, x_ss, w, w_ss, nullptr, bias, nullptr, nullptr, nullptr, 0, y, y_ss, B, H, W, Cin, Cout, ksize, stride, pad, relu,
                                  n_samples, nullptr, stream);
}

// =====================================================================================
// Fused fp32 Bayes-by-backprop MLP (BASELINE config 0: reference models_bbb.LinearNetwork, :32-78, eval branch of
// bbb/linear.py:42-50): 3 x (Linear + ReLU), heads mu and log_var -> (mu, exp(log_var)).
// The layer-by-layer path is ~25 launches of a few microseconds for a network of 21.5 k weights -- launches, not arithmetic.
// Here: ONE launch draws the S sampled weight sets of all five layers (the Philox normal stream of qbnn_sample_weights_f32:
// ctr = {i >> 2, layer, sample, 0}, W = mu + (eps * sigma), two roundings), ONE launch runs the whole network for a tile of
// 32 rows of one MC sample: the layer's sampled weights and the activations sit in LDS, thread (row, part) owns every 8th
// neuron of its row and accumulates  acc = fma(h[k], W[n][k], acc)  for k ascending, then + bias, ReLU -- the reference's
// x @ W^T + b in fp32.
// =====================================================================================
#define QBNN_MLP_LAYERS 5
#define QBNN_MLP_MAXW 128          // widest layer (inputs or outputs)
struct MlpArgs {
  const float* mu[QBNN_MLP_LAYERS]; const float* sigma[QBNN_MLP_LAYERS]; const float* bias[QBNN_MLP_LAYERS];
  int out[QBNN_MLP_LAYERS], in[QBNN_MLP_LAYERS];
  uint32_t layer_id[QBNN_MLP_LAYERS];
  int64_t woff[QBNN_MLP_LAYERS + 1];      // element offset of each layer inside one sample's weight block
  int goff[QBNN_MLP_LAYERS + 1];          // the same in 4-weight groups (a group never straddles layers)
  const float* x; int B;
  float* w; float* mu_out; float* var_out;
};

__global__ __launch_bounds__(256) void mlp_sample_weights_kernel(const MlpArgs a, uint32_t seed_lo, uint32_t seed_hi, uint32_t sample_begin,
                                                                 const uint32_t* __restrict__ nd) {
  if (nd) { seed_lo = nd[0]; seed_hi = nd[1]; sample_begin = nd[2]; }
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= a.goff[QBNN_MLP_LAYERS]) return;
  int l = 0;
#pragma unroll
  for (int i = 1; i < QBNN_MLP_LAYERS; ++i) l = g >= a.goff[i] ? i : l;
  const int s = blockIdx.y;
  const int64_t gl = g - a.goff[l], n = (int64_t)a.out[l] * a.in[l];
  float e[4];
  qbnn::normal4(qbnn::philox4x32_10((uint32_t)gl, a.layer_id[l], sample_begin + s, 0u, seed_lo, seed_hi), e);
  float* w = a.w + (int64_t)s * a.woff[QBNN_MLP_LAYERS] + a.woff[l];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t i = gl * 4 + j;
    if (i < n) { const float t = e[j] * a.sigma[l][i]; w[i] = a.mu[l][i] + t; }
  }
}

// J = neurons per thread (every 8th): ceil(widest hidden layer / 8).  The layer's weight rows are zero-padded to 8 J rows in LDS, so
// the inner loop carries no per-neuron condition (a lane-dependent `n < N` test per accumulator made it 16 divergent branches per step).
// LDS is sized by the launch for the network at hand (wmax = widest padded input row, HP = activation pitch): the 4 x 100 MLP takes
// 68 KB, so two workgroups share a CU -- with the compile-time maximum (87 KB) the 320 workgroups of config 0 ran in two rounds.
template <int J>
__global__ __launch_bounds__(256) void mlp_forward_kernel(const MlpArgs a, const int wmax, const int HP) {
  constexpr int ROWS = 32;
  extern __shared__ __attribute__((aligned(16))) float mlp_smem[];
  float* wl = mlp_smem;                                     // [8 J][KP] weights of the current layer
  float* hbuf0 = wl + 8 * J * wmax;                         // two activation buffers [ROWS][HP]
  float* hbuf[2] = {hbuf0, hbuf0 + ROWS * HP};
  const int tid = threadIdx.x, r = tid >> 3, o = tid & 7;
  const int s = blockIdx.y, row0 = blockIdx.x * ROWS;
  const float* ws = a.w + (int64_t)s * a.woff[QBNN_MLP_LAYERS];
  // input rows (zero beyond B and beyond in_dim, up to the next multiple of 4)
  const int K0 = a.in[0], K0P = (K0 + 3) & ~3;
  for (int i = tid; i < ROWS * K0P; i += 256) {
    const int rr = i / K0P, k = i - rr * K0P;
    hbuf[0][rr * HP + k] = (row0 + rr < a.B && k < K0) ? a.x[(int64_t)(row0 + rr) * K0 + k] : 0.f;
  }
  int cur = 0;
#pragma unroll 1
  for (int l = 0; l < 3; ++l) {
    const int K = a.in[l], KP = (K + 3) & ~3, N = a.out[l];
    __syncthreads();                                        // previous layer done with wl; its outputs complete
    const float* wsl = ws + a.woff[l];
    if ((K & 3) == 0) {              // whole float4s (layer offsets and rows are 16-byte aligned): several loads in flight per thread
      const int Q = KP / 4;
#pragma unroll 4
      for (int i = tid; i < 8 * J * Q; i += 256) {
        const int n = i / Q, q4 = i - n * Q;
        const float4 v = n < N ? *reinterpret_cast<const float4*>(wsl + (int64_t)n * K + 4 * q4) : float4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<float4*>(wl + n * KP + 4 * q4) = v;
      }
    } else {
#pragma unroll 4
      for (int i = tid; i < 8 * J * KP; i += 256) {
        const int n = i / KP, k = i - n * KP;
        wl[i] = (n < N && k < K) ? wsl[(int64_t)n * K + k] : 0.f;
      }
    }
    __syncthreads();
    const float* h = hbuf[cur] + r * HP;
    float* ho = hbuf[cur ^ 1] + r * HP;
    const float* wr = wl + o * KP;
    float acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j] = 0.f;
    for (int k4 = 0; k4 < KP; k4 += 4) {
      const float4 hv = *reinterpret_cast<const float4*>(h + k4);
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const float4 wv = *reinterpret_cast<const float4*>(wr + 8 * j * KP + k4);
        float v = acc[j];
        v = __builtin_fmaf(hv.x, wv.x, v); v = __builtin_fmaf(hv.y, wv.y, v); v = __builtin_fmaf(hv.z, wv.z, v); v = __builtin_fmaf(hv.w, wv.w, v);
        acc[j] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int n = o + 8 * j;
      if (n < N) {
        float v = acc[j];
        if (a.bias[l]) v = v + a.bias[l][n];
        ho[n] = fmaxf(v, 0.f);
      }
    }
    if (o == 0) for (int n = N; n < ((N + 3) & ~3); ++n) ho[n] = 0.f;       // zero the k padding of the next layer
    cur ^= 1;
  }
  __syncthreads();
  // heads: mu (layer 3) and log_var (layer 4), one output each; var = exp(log_var) (models_bbb.py:78)
  if (o < 2 && row0 + r < a.B) {
    const int l = 3 + o, K = a.in[l];
    const float* wv = ws + a.woff[l];
    const float* h = hbuf[cur] + r * HP;
    float v = 0.f;
    for (int k = 0; k < K; ++k) v = __builtin_fmaf(h[k], wv[k], v);
    if (a.bias[l]) v = v + a.bias[l][0];
    if (o == 0) a.mu_out[(int64_t)s * a.B + row0 + r] = v;
    else a.var_out[(int64_t)s * a.B + row0 + r] = expf(v);
  }
}

QBNN_EXPORT int qbnn_mlp_bbb_f32_mc(const float* x, int32_t B, const qbnn_mlp_layer* layers, uint64_t seed, uint32_t sample_begin,
                                    int32_t n_samples, float* w_workspace, float* mu_out, float* var_out, void* stream) {
  if (!x || !layers || !w_workspace || !mu_out || !var_out || B <= 0 || n_samples <= 0)
    return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_mlp_bbb_f32_mc: bad argument");
  MlpArgs a;
  memset(&a, 0, sizeof(a));
  int64_t off = 0;
  int goff = 0;
  for (int l = 0; l < QBNN_MLP_LAYERS; ++l) {
    const qbnn_mlp_layer& q = layers[l];
    if (!q.mu || !q.sigma || q.out_features <= 0 || q.in_features <= 0 || q.in_features > QBNN_MLP_MAXW || q.out_features > QBNN_MLP_MAXW)
      return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_mlp_bbb_f32_mc: layer widths must be in [1, 128]");
    if (l >= 3 && q.out_features != 1) return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_mlp_bbb_f32_mc: layers 3 and 4 are the mu / log_var heads (one output each)");
    if (l > 0 && l < 3 && q.in_features != layers[l - 1].out_features) return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_mlp_bbb_f32_mc: layer widths do not chain");
    if (l >= 3 && q.in_features != layers[2].out_features) return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_mlp_bbb_f32_mc: the heads read the last hidden layer");
    a.mu[l] = q.mu; a.sigma[l] = q.sigma; a.bias[l] = q.bias; a.out[l] = q.out_features; a.in[l] = q.in_features; a.layer_id[l] = q.layer_id;
    a.woff[l] = off; a.goff[l] = goff;
    const int64_t n = (int64_t)q.out_features * q.in_features;
    off += (n + 3) / 4 * 4;                 // every layer starts on a group boundary
    goff += (int)((n + 3) / 4);
  }
  a.woff[QBNN_MLP_LAYERS] = off; a.goff[QBNN_MLP_LAYERS] = goff;
  a.x = x; a.B = B; a.w = w_workspace; a.mu_out = mu_out; a.var_out = var_out;
  hipLaunchKernelGGL(mlp_sample_weights_kernel, dim3((goff + 255) / 256, n_samples), dim3(256), 0, (hipStream_t)stream, a, (uint32_t)seed,
                     (uint32_t)(seed >> 32), sample_begin, qbnn_noise_dev());
  int widest = 0, wmax = 0;
  for (int l = 0; l < 3; ++l) {
    widest = layers[l].out_features > widest ? layers[l].out_features : widest;
    const int kp = (layers[l].in_features + 3) & ~3;
    wmax = kp > wmax ? kp : wmax;
  }
  const int hp = ((widest > wmax ? widest : wmax) + 3) / 4 * 4 + 4;
  const dim3 grid((B + 31) / 32, n_samples);
  if (widest <= 104) {
    const int lds = (8 * 13 * wmax + 2 * 32 * hp) * 4;
    static std::atomic<uint64_t> attr13{0};
    if (int rc = qbnn_ensure_dyn_lds((const void*)mlp_forward_kernel<13>, &attr13, lds)) return rc;
    hipLaunchKernelGGL(mlp_forward_kernel<13>, grid, dim3(256), lds, (hipStream_t)stream, a, wmax, hp);
  } else {
    const int lds = (8 * 16 * wmax + 2 * 32 * hp) * 4;
    static std::atomic<uint64_t> attr16{0};
    if (int rc = qbnn_ensure_dyn_lds((const void*)mlp_forward_kernel<16>, &attr16, lds)) return rc;
    hipLaunchKernelGGL(mlp_forward_kernel<16>, grid, dim3(256), lds, (hipStream_t)stream, a, wmax, hp);
  }
  return qbnn_check_launch_msg("qbnn_mlp_bbb_f32_mc");
}

QBNN_EXPORT int64_t qbnn_mlp_bbb_f32_workspace_floats(const qbnn_mlp_layer* layers) {
  int64_t off = 0;
  for (int l = 0; l < QBNN_MLP_LAYERS; ++l) off += ((int64_t)layers[l].out_features * layers[l].in_features + 3) / 4 * 4;
  return off;
}
