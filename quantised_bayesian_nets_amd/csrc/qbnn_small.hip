// libqbnn_hip.so, third translation unit: the small networks' layers on the matrix pipe (BASELINE configs 1-2: LeNet,
// MLP; reference mcdropout/models_mc.py:75-111, bbb/models_bbb.py LeNet / linear graphs in their converted int8 form).
//
//   qbnn_conv_pool_drop_i8_mc : k x k / stride 1 conv on a small map (the whole image lives in LDS), then optionally
//                               MaxPool2d(2,2) and the quantised BernoulliDropout, written as flattened NHWC rows with a
//                               16-byte-multiple pitch -- conv -> pool -> dropout -> Flatten of LeNet in ONE launch.
//   qbnn_linear_i8_mc         : quantised Linear / LinearReLU as an LDS-tiled int8 GEMM (128 x 128 tiles), optionally with the
//                               per-element BernoulliDropout of a 2-D activation in its epilogue.
// Both take weights in the QBNN_LAYOUT_MFMA32 fragment layout (qbnn_pack_weights_host / the sampler), so fixed (MC-Dropout,
// sample stride 0) and sampled (Bayes-by-backprop) weights run through the same code.  Arithmetic contract = the any-geometry
// kernels of qbnn_kernels.hip (qbnn_conv2d_i8_generic_mc, qbnn_maxpool2_q_mc, qbnn_dropout_q_mc): the tests compare them bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <atomic>

#include "../../include/qbnn.h"
#include "qbnn_common.h"
#include "qbnn_rng.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));

namespace {

int failf(int code, const char* msg) { return qbnn_fail_msg(code, msg); }

__device__ __forceinline__ int rne_sat(float v) {
  v = fminf(fmaxf(v, -1.0e9f), 1.0e9f);
  return __float2int_rn(v);
}

// bytes x - z (x, z in [0,127]) of a dword, each as a signed byte
__device__ __forceinline__ uint32_t centre4(uint32_t v, uint32_t z4) {
  return ((v | 0x80808080u) - z4) ^ 0x80808080u;
}

#define QBNN_MAGIC 12582912.0f     // 1.5 * 2^23: (v + MAGIC) has rne(v) in its low mantissa bits for |v| < 2^22
__device__ __forceinline__ float med3f(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
// low bytes of four fp32 bit patterns -> one dword (element 0 in byte 0)
__device__ __forceinline__ uint32_t pack_low_bytes(float t0, float t1, float t2, float t3) {
  const uint32_t p01 = __builtin_amdgcn_perm(__float_as_uint(t1), __float_as_uint(t0), 0x0c0c0400u);
  const uint32_t p23 = __builtin_amdgcn_perm(__float_as_uint(t3), __float_as_uint(t2), 0x04000c0cu);
  return p01 | p23;
}
// per-byte max of two dwords of bytes < 128
__device__ __forceinline__ uint32_t max4_u7(uint32_t a, uint32_t b) {
  const uint32_t ge = (((a | 0x80808080u) - b) & 0x80808080u) >> 7;      // 1 where a >= b
  const uint32_t mask = (ge << 8) - ge;                                   // 0xff there
  return (a & mask) | (b & ~mask);
}
__device__ __forceinline__ float ubyte_f(uint32_t w, int j) { return (float)((w >> (8 * j)) & 0xffu); }     // v_cvt_f32_ubyteN

struct Requant {                    // FBGEMM requantisation of one conv / linear layer (qconv.cpp / qlinear.cpp)
  int z_w, z_y, lo, hi; float rcp, mult; const float* bias;
};
__device__ __forceinline__ int requant(int a, int n, const Requant& q) {
  float xf = (float)a;
  if (q.bias) xf = __builtin_fmaf(q.bias[n], q.rcp, xf);
  return min(max(q.z_y + rne_sat(xf * q.mult), q.lo), q.hi);
}

struct Drop {                       // quantised BernoulliDropout behind the layer (mcdropout/dropout.py:15-40); on = 0: none
  int on; float keep, inv_sm, mult; int z_x, z_m, hi;
  float zxf, zmf, dlo, dhi;         // float form: r' = rne(clamp(((x - z_x) mq) mult, -z_m, min(255, hi) - z_m)), byte = r' + z_m
  uint32_t seed_lo, seed_hi, layer_id, sample_begin; const float* mask_in; const uint32_t* nd;
};
__device__ __forceinline__ int mask_from(float m, const Drop& d) { return min(max(d.z_m + rne_sat(m * d.inv_sm), 0), 255) - d.z_m; }
__device__ __forceinline__ int drop_apply(int xb, int mq, const Drop& d) {
  const int q = min(max(d.z_m + rne_sat((float)((xb - d.z_x) * mq) * d.mult), 0), 255);
  return min(q, d.hi);
}
// the same map on floats ((x - z_x) mq is an exact small integer): the CENTRED result r' before rounding
__device__ __forceinline__ float drop_centred(float xf, float mqf, const Drop& d) { return med3f(((xf - d.zxf) * mqf) * d.mult, d.dlo, d.dhi); }
__device__ __forceinline__ float keep_draw(uint32_t rv, float keep) { return ((float)(rv >> 8) * 5.9604644775390625e-8f) < keep ? 1.0f : 0.0f; }

// =====================================================================================
// conv (+ pool + dropout) on a small map
// =====================================================================================
struct ConvSmallArgs {
  const uint8_t* x; int64_t x_ss;      // [S|1][B][HIN][HIN][CIN] quint8
  const int8_t* w; int64_t w_ss;       // MFMA32 fragments (cout = COUT, k = KSZ KSZ CIN, krow = KSZ CIN)
  uint8_t* y; int64_t y_ss; int ldy;   // [S][B][ldy]: flattened NHWC map (pooled if POOL), bytes beyond the map 0
  int B, n_samples, z_x;
  Requant q; Drop d;
  Drop din;                            // optional dropout applied to x on the way into LDS (one draw per (sample, image, input channel));
                                       // z_x is then the zero point of ITS output
  float vlo, vhi;                      // requantisation clamp around z_y: lo - z_y, hi - z_y
};

template <int HIN_, int CIN_, int KSZ_, int COUT_, int G_>
struct SmallCfg {
  static constexpr int HIN = HIN_, CIN = CIN_, KSZ = KSZ_, COUT = COUT_, G = G_;
  static constexpr int PAD = KSZ / 2, TW = HIN + 2 * PAD, PITCH = TW * CIN;
  static constexpr int TILE = TW * TW * CIN;
  static constexpr int RB = KSZ * CIN, RBP = (RB + 31) / 32 * 32, SPR = RBP / 32, KS = KSZ * SPR;
  static constexpr int NT = (COUT + 32) / 32;              // + the ones row (COUT % 32 != 0 is required)
  static constexpr int M = G * HIN * HIN, MT = (M + 31) / 32;
  static constexpr int OUTP = (COUT + 3) / 4 * 4;          // staging bytes per pixel
  static constexpr int TILES_BYTES = (G * TILE + 64 + 15) / 16 * 16;      // + over-read slack of the last kernel row
  static constexpr int A_BYTES = NT * KS * 1024;
  static constexpr int OUT_BYTES = (MT * 32 * OUTP + 15) / 16 * 16;
  static constexpr int LDS = TILES_BYTES + OUT_BYTES + G * COUT * 4 + NT * 32 * 4 + G * CIN * 4;
  static_assert(COUT % 2 == 0, "rows are written as 16-bit pieces");
  static constexpr int ONES_TILE = COUT / 32, ONES_LOCAL = COUT % 32;
  static_assert(COUT % 32 != 0, "the window sum comes from the packed layout's ones row");
  static_assert(CIN % 4 == 0, "pixels are moved as dwords");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

struct __attribute__((packed, aligned(4))) frag16 { v4i v; };     // a 16-byte B fragment at a 4-byte-aligned LDS address

#ifndef QBNN_CS_THREADS
#define QBNN_CS_THREADS 512
#endif
constexpr int CS_THREADS = QBNN_CS_THREADS;

template <class C, bool POOL>
__global__ __launch_bounds__(CS_THREADS) void conv_pool_drop_i8_kernel(const ConvSmallArgs a) {
  // Persistent workgroup of CS_THREADS / 64 waves (two workgroups of 4 share a CU, so one's load / store phases run under the other's MFMAs); work item = (MC sample, G images), contiguous item ranges per workgroup.  Every wave keeps the
  // layer's WHOLE weight (NT x KS fragments = 160 VGPRs for LeNet's 20 -> 50 5x5) in registers -- loaded once per kernel for fixed
  // weights, once per sample change for sampled ones -- so the K loop reads only the pixel fragments from LDS (one 16-byte fragment per
  // NT MFMAs).  The next item's pixels are fetched into registers under the current item's MFMAs.
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* tiles = smem;
  uint8_t* outb = smem + C::TILES_BYTES;
  float* mq = reinterpret_cast<float*>(outb + C::OUT_BYTES);
  float* bias_lds = mq + C::G * C::COUT;
  float* mqin = bias_lds + C::NT * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int HW = C::HIN * C::HIN, DPP = C::CIN / 4;          // dwords per pixel
  constexpr int NDW = C::G * HW * DPP, PER_T = (NDW + CS_THREADS - 1) / CS_THREADS;
  const int groups = (a.B + C::G - 1) / C::G;
  const int items = a.n_samples * groups;
  const int per = (items + gridDim.x - 1) / gridDim.x;
  const int it0 = blockIdx.x * per, it1 = min(items, it0 + per);
  Drop d = a.d, din = a.din;
  if (d.on && d.nd) { d.seed_lo = d.nd[0]; d.seed_hi = d.nd[1]; d.sample_begin = d.nd[2]; }
  if (din.on && din.nd) { din.seed_lo = din.nd[0]; din.seed_hi = din.nd[1]; din.sample_begin = din.nd[2]; }
  auto draw_masks = [&](const Drop& dd, float* dst, int CH, int s, int img0) {       // dst[g][c] = mask value - z_m of (sample s, image, channel)
    for (int i = tid; i < C::G * CH; i += CS_THREADS) {
      const int b = img0 + i / CH, c = i % CH;
      float m = 0.f;
      if (b < a.B) {
        const int slot = b * CH + c;
        if (dd.mask_in) m = dd.mask_in[(int64_t)s * a.B * CH + slot];
        else {
          const qbnn::u32x4 r = qbnn::philox4x32_10((uint32_t)(slot >> 2), dd.layer_id, dd.sample_begin + s, 1u, dd.seed_lo, dd.seed_hi);
          m = keep_draw((slot & 3) == 0 ? r.x : ((slot & 3) == 1 ? r.y : ((slot & 3) == 2 ? r.z : r.w)), dd.keep);
        }
      }
      dst[i] = (float)mask_from(m, dd);
    }
  };
  for (int i = tid; i < C::TILES_BYTES / 4; i += CS_THREADS) reinterpret_cast<uint32_t*>(tiles)[i] = 0u;     // halo = centred 0
  for (int i = tid; i < C::NT * 32; i += CS_THREADS) bias_lds[i] = (a.q.bias && i < C::COUT) ? a.q.bias[i] : 0.f;
  const uint32_t z4 = (uint32_t)a.z_x * 0x01010101u;
  // this thread's dwords of an item's G images: (LDS offset, global offset from the item's first image) -- the same for every item
  int t_lds[PER_T], t_src[PER_T], t_mq[PER_T];
#pragma unroll
  for (int u = 0; u < PER_T; ++u) {
    const int i = tid + CS_THREADS * u;
    const int g = i / (HW * DPP), r = i - g * (HW * DPP), px = r / DPP, c4 = r - px * DPP;
    const int oh = px / C::HIN, ow = px - oh * C::HIN;
    t_lds[u] = i < NDW ? g * C::TILE + ((oh + C::PAD) * C::TW + ow + C::PAD) * C::CIN + 4 * c4 : -1;
    t_src[u] = (g * HW + px) * C::CIN + 4 * c4;
    t_mq[u] = g * C::CIN + 4 * c4;
  }
  uint32_t pre[PER_T];
  auto fetch = [&](int it) {
    const int s = it / groups, img0 = (it - s * groups) * C::G;
    const uint8_t* xs = a.x + (int64_t)s * a.x_ss + (int64_t)img0 * HW * C::CIN;
#pragma unroll
    for (int u = 0; u < PER_T; ++u) {
      pre[u] = din.on ? (uint32_t)din.z_x * 0x01010101u : z4;          // images beyond the batch: centred 0 either way
      if (t_lds[u] >= 0 && img0 + (tid + CS_THREADS * u) / (HW * DPP) < a.B) pre[u] = *reinterpret_cast<const uint32_t*>(xs + t_src[u]);
    }
  };
  v4i areg[C::NT][C::KS];
  int s_loaded = -1;
  if (it0 < it1) fetch(it0);
  for (int it = it0; it < it1; ++it) {
    const int s = it / groups, img0 = (it - s * groups) * C::G;
    if (s != s_loaded && (s_loaded < 0 || a.w_ss != 0)) {
      const int8_t* ws = a.w + (int64_t)s * a.w_ss;
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) areg[nt][ks] = *reinterpret_cast<const v4i*>(ws + ((int64_t)(nt * C::KS + ks) * 64 + lane) * 16);
    }
    s_loaded = s;
    __syncthreads();                                  // previous item's store pass has read outb / mq; first trip: halo zeroed
    if (din.on) {                                     // dropout in front of the conv, applied while the pixels enter LDS
      draw_masks(din, mqin, C::CIN, s, img0);
      __syncthreads();
#pragma unroll
      for (int u = 0; u < PER_T; ++u)
        if (t_lds[u] >= 0) {
          const float4 m4 = *reinterpret_cast<const float4*>(mqin + t_mq[u]);
          const uint32_t w = pre[u];
          // centred on the dropout's output zero point (= the conv's z_x): exactly what the tile holds
          *reinterpret_cast<uint32_t*>(tiles + t_lds[u]) =
              pack_low_bytes(drop_centred(ubyte_f(w, 0), m4.x, din) + QBNN_MAGIC, drop_centred(ubyte_f(w, 1), m4.y, din) + QBNN_MAGIC,
                             drop_centred(ubyte_f(w, 2), m4.z, din) + QBNN_MAGIC, drop_centred(ubyte_f(w, 3), m4.w, din) + QBNN_MAGIC);
        }
    } else {
#pragma unroll
      for (int u = 0; u < PER_T; ++u)
        if (t_lds[u] >= 0) *reinterpret_cast<uint32_t*>(tiles + t_lds[u]) = centre4(pre[u], z4);
    }
    if (d.on) draw_masks(d, mq, C::COUT, s, img0);
    __syncthreads();
    if (it + 1 < it1) fetch(it + 1);                  // in flight under the MFMAs below
    // ---- implicit GEMM: M-tile = 32 pixels (lane & 31), both k-halves (lane >> 5); all NT channel tiles per B fragment
    for (int mt = wave; mt < C::MT; mt += CS_THREADS / 64) {
      const int p = mt * 32 + (lane & 31);
      const bool valid = p < C::M;
      const int g = valid ? p / HW : 0, r = valid ? p - g * HW : 0, oh = r / C::HIN, ow = r - oh * C::HIN;
      const uint8_t* bp = tiles + g * C::TILE + (oh * C::TW + ow) * C::CIN + 16 * (lane >> 5);
      v16i acc[C::NT];
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0;
      v4i bv = reinterpret_cast<const frag16*>(bp)->v;
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        v4i bn = bv;
        if (ks + 1 < C::KS) bn = reinterpret_cast<const frag16*>(bp + ((ks + 1) / C::SPR) * C::PITCH + ((ks + 1) % C::SPR) * 32)->v;
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(areg[nt][ks], bv, acc[nt], 0, 0, 0);
        bv = bn;
      }
      // window sum R = the ones row (output channel COUT): register of local channel ONES_LOCAL on the lanes of its k-half
      constexpr int OG = C::ONES_LOCAL / 8, OH = (C::ONES_LOCAL % 8) / 4, OI = C::ONES_LOCAL % 4;
      const int R = __shfl(acc[C::ONES_TILE][4 * OG + OI], (lane & 31) + 32 * OH);
      const int zr = a.q.z_w * R;
      const int h = lane >> 5;
      const float zyf = (float)a.q.z_y;
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          if (nt * 32 + 8 * g4 >= C::COUT) continue;                 // compile-time: channel groups beyond COUT
          const int n0 = nt * 32 + 8 * g4 + 4 * h;
          if (n0 >= C::COUT) continue;
          const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + n0);
          const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
          float t[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float xf = (float)(acc[nt][4 * g4 + i] - zr);
            if (a.q.bias) xf = __builtin_fmaf(bb[i], a.q.rcp, xf);
            // z_y + rne(clamp(v, lo - z_y, hi - z_y)) == clamp(z_y + rne(v), lo, hi): round with the even magic constant, add z_y exactly
            t[i] = (med3f(xf * a.q.mult, a.vlo, a.vhi) + QBNN_MAGIC) + zyf;
          }
          const uint32_t o = pack_low_bytes(t[0], t[1], t[2], t[3]);     // channels >= COUT of the last group: never read back
          *reinterpret_cast<uint32_t*>(outb + p * C::OUTP + n0) = o;
        }
    }
    __syncthreads();
    // ---- store pass: (max-pool) -> (dropout) -> flattened NHWC rows; one group of 4 channels of one output pixel per thread trip
    constexpr int HO = POOL ? C::HIN / 2 : C::HIN;
    constexpr int ROW = HO * HO * C::COUT, CG = C::OUTP / 4;
    uint8_t* ys = a.y + (int64_t)s * a.y_ss;
    for (int i = tid; i < C::G * HO * HO * CG; i += CS_THREADS) {
      const int g = i / (HO * HO * CG), r = i - g * (HO * HO * CG), px = r / CG, c = (r - px * CG) * 4;
      if (img0 + g >= a.B) continue;
      uint32_t w;
      if (POOL) {
        const int ph = px / HO, pw = px - ph * HO;
        const uint8_t* q0 = outb + (g * HW + (2 * ph) * C::HIN + 2 * pw) * C::OUTP + c;
        w = max4_u7(max4_u7(*reinterpret_cast<const uint32_t*>(q0), *reinterpret_cast<const uint32_t*>(q0 + C::OUTP)),
                    max4_u7(*reinterpret_cast<const uint32_t*>(q0 + C::HIN * C::OUTP), *reinterpret_cast<const uint32_t*>(q0 + (C::HIN + 1) * C::OUTP)));
      } else {
        w = *reinterpret_cast<const uint32_t*>(outb + (g * HW + px) * C::OUTP + c);
      }
      if (d.on) {
        const float* mp = mq + g * C::COUT + c;                 // (the last group's channels beyond COUT read the next image's masks: unused)
        w = pack_low_bytes((drop_centred(ubyte_f(w, 0), mp[0], d) + QBNN_MAGIC) + d.zmf, (drop_centred(ubyte_f(w, 1), mp[1], d) + QBNN_MAGIC) + d.zmf,
                           (drop_centred(ubyte_f(w, 2), mp[2], d) + QBNN_MAGIC) + d.zmf, (drop_centred(ubyte_f(w, 3), mp[3], d) + QBNN_MAGIC) + d.zmf);
      }
      uint16_t* dst = reinterpret_cast<uint16_t*>(ys + (int64_t)(img0 + g) * a.ldy + px * C::COUT + c);
      dst[0] = (uint16_t)w;
      if (c + 2 < C::COUT) dst[1] = (uint16_t)(w >> 16);
    }
    for (int i = tid; i < C::G * (a.ldy - ROW); i += CS_THREADS) {       // bytes beyond the map
      const int g = i / (a.ldy - ROW);
      if (img0 + g < a.B) ys[(int64_t)(img0 + g) * a.ldy + ROW + (i - g * (a.ldy - ROW))] = 0;
    }
  }
}

template <class C>
int launch_conv_small(const ConvSmallArgs& a, bool pool, hipStream_t st) {
  const int items = a.n_samples * ((a.B + C::G - 1) / C::G);
  const int cap = 256 * (1024 / CS_THREADS) / 2;                 // workgroups resident at once: 512 registers per lane and SIMD / 256 per wave
  const int grid = items < cap ? items : cap;
  if (pool) {
    static std::atomic<uint64_t> done{0};
    if (int rc = qbnn_ensure_dyn_lds((const void*)conv_pool_drop_i8_kernel<C, true>, &done, C::LDS)) return rc;
    hipLaunchKernelGGL((conv_pool_drop_i8_kernel<C, true>), dim3(grid), dim3(CS_THREADS), C::LDS, st, a);
  } else {
    static std::atomic<uint64_t> done{0};
    if (int rc = qbnn_ensure_dyn_lds((const void*)conv_pool_drop_i8_kernel<C, false>, &done, C::LDS)) return rc;
    hipLaunchKernelGGL((conv_pool_drop_i8_kernel<C, false>), dim3(grid), dim3(CS_THREADS), C::LDS, st, a);
  }
  return qbnn_check_launch_msg("qbnn_conv_pool_drop_i8_mc");
}

int fill_requant(Requant& q, const qbnn_conv_desc* d, const float* bias) {
  if (d->z_x < 0 || d->z_x > 127 || d->a_hi > 127 || d->a_hi < 1 || d->z_y < 0 || d->z_y > 127)
    return failf(QBNN_E_INVALID, "qbnn small-layer kernels: activations must be <= 7 bit with zero points in [0,127] (reference quant_utils.py:120)");
  const float atw = d->s_x * d->s_w;
  q.rcp = 1.0f / atw; q.mult = atw / d->s_y;
  q.z_w = d->z_w; q.z_y = d->z_y; q.lo = d->relu ? d->z_y : 0; q.hi = d->a_hi < 255 ? d->a_hi : 255;
  q.bias = d->has_bias ? bias : nullptr;
  return QBNN_OK;
}

// dropout on a tensor with qparams (s_x, z_x)
int fill_drop(Drop& o, float s_x, int z_x, int a_hi, const qbnn_dropout_desc* p, const float* mask_in, uint64_t seed, uint32_t sample_begin) {
  memset(&o, 0, sizeof(o));
  if (!p) return QBNN_OK;
  if (!(p->s_m > 0.f) || p->z_m < 0 || p->z_m > 127 || z_x < 0 || z_x > 127)
    return failf(QBNN_E_INVALID, "qbnn small-layer kernels: bad mask qparams (zero points in [0,127])");
  o.on = 1; o.keep = p->keep_prob; o.inv_sm = 1.0f / p->s_m; o.z_m = p->z_m; o.z_x = z_x;
  o.mult = (float)((double)s_x * (double)p->s_m / (double)p->s_m);      // ATen qmul: self_scale * other_scale / out_scale
  o.hi = a_hi < 255 ? a_hi : 255;
  o.zxf = (float)o.z_x; o.zmf = (float)o.z_m; o.dlo = (float)(-o.z_m); o.dhi = (float)(o.hi - o.z_m);
  o.seed_lo = (uint32_t)seed; o.seed_hi = (uint32_t)(seed >> 32); o.layer_id = p->layer_id; o.sample_begin = sample_begin;
  o.mask_in = mask_in; o.nd = qbnn_noise_dev();
  return QBNN_OK;
}

// =====================================================================================
// Linear as an int8 GEMM: Y[b][n] = requant( sum_k (x[b][k] - z_x)(w[n][k] - z_w) ).  Workgroup = 4 waves, tile 128 rows x 128
// outputs (wave = 64 x 64 = 2 x 2 MFMA tiles), K in chunks of 64 bytes staged through LDS: x rows centred on the way in
// (pitch 80: conflict-free fragment reads), weights as whole 1 KiB fragments of the MFMA32 layout; the next chunk's global loads
// are in flight under the current chunk's MFMAs.  The row sums R (for the weights' zero point) are v_dot4 over the fragments.
// =====================================================================================
struct LinearArgs {
  const uint8_t* x; int64_t x_ss; int ldx;     // [S|1][B][ldx] quint8 rows, ldx % 16 == 0, ldx >= K
  const int8_t* w; int64_t w_ss;               // MFMA32 fragments (cout = N, k = krow = K)
  uint8_t* y; int64_t y_ss; int ldy;           // [S][B][ldy], ldy >= N; bytes N..ldy-1 are written 0
  int B, K, N, KS, z_x;
  Requant q; Drop d;
};

constexpr int LIN_BP = 80;                      // LDS pitch of a 64-byte x-row chunk

__global__ __launch_bounds__(256) void linear_i8_kernel(const LinearArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t Bs[2][128 * LIN_BP];
  __shared__ __attribute__((aligned(16))) uint8_t As[2][8 * 1024];          // [n-tile 0..3][k-step 0..1] fragments
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int s = blockIdx.z, m0 = blockIdx.x * 128, nt0 = blockIdx.y * 4;
  const int NT = (a.N + 31) / 32;                                            // packed n-tiles (the layout's ones row, if any, is ignored here)
  const uint8_t* xs = a.x + (int64_t)s * a.x_ss;
  const int8_t* ws = a.w + (int64_t)s * a.w_ss;
  const uint32_t z4 = (uint32_t)a.z_x * 0x01010101u;
  const int nchunk = (a.KS + 1) / 2;
  // staging roles: x -- thread t moves 16 bytes of row t / 4 (+ 64), piece t % 4;  w -- 16 bytes of fragment t / 64 (+ 4)
  const int xr = tid >> 2, xp = tid & 3;
  v4i xv[2], wv[2];
  auto fetch = [&](int ch) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = m0 + xr + 64 * u, kb = ch * 64 + 16 * xp;
      v4i v = v4i{0, 0, 0, 0};
      if (row < a.B && kb < a.K) {
        v = *reinterpret_cast<const v4i*>(xs + (int64_t)row * a.ldx + kb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          uint32_t c = centre4((uint32_t)v[i], z4);
          const int left = a.K - (kb + 4 * i);                               // bytes of this dword inside K
          if (left < 4) c = left <= 0 ? 0u : (c & (0xffffffffu >> (8 * (4 - left))));
          v[i] = (int)c;
        }
      }
      xv[u] = v;
      const int f = (tid >> 6) + 4 * u;                                      // fragment f = nt_local * 2 + ks_local
      const int nt = nt0 + (f >> 1), ks = ch * 2 + (f & 1);
      wv[u] = (nt < NT && ks < a.KS) ? *reinterpret_cast<const v4i*>(ws + (((int64_t)nt * a.KS + ks) * 64 + lane) * 16) : v4i{0, 0, 0, 0};
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      *reinterpret_cast<v4i*>(&Bs[buf][(xr + 64 * u) * LIN_BP + 16 * xp]) = xv[u];
      *reinterpret_cast<v4i*>(&As[buf][(((tid >> 6) + 4 * u) * 64 + lane) * 16]) = wv[u];
    }
  };
  v16i acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
  int rs[2] = {0, 0};
  fetch(0);
  stage(0);
  __syncthreads();
  for (int ch = 0; ch < nchunk; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunk) fetch(ch + 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      v4i bv[2], av[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        bv[i] = *reinterpret_cast<const v4i*>(&Bs[buf][(wm * 64 + i * 32 + (lane & 31)) * LIN_BP + kk * 32 + 16 * (lane >> 5)]);
        av[i] = *reinterpret_cast<const v4i*>(&As[buf][(((wn * 2 + i) * 2 + kk) * 64 + lane) * 16]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[j], bv[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) rs[i] = __builtin_amdgcn_sdot4(bv[i][r], 0x01010101, rs[i], false);
      }
    }
    if (ch + 1 < nchunk) stage(buf ^ 1);
    __syncthreads();
  }
  Drop d = a.d;
  if (d.on && d.nd) { d.seed_lo = d.nd[0]; d.seed_hi = d.nd[1]; d.sample_begin = d.nd[2]; }
  const int h = lane >> 5;
  uint8_t* ys = a.y + (int64_t)s * a.y_ss;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int R = rs[i] + __shfl_xor(rs[i], 32);
    const int row = m0 + wm * 64 + i * 32 + (lane & 31);
    if (row >= a.B) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int n0 = (nt0 + wn * 2 + j) * 32 + 8 * g4 + 4 * h;
        if (n0 >= a.ldy || ((a.ldy & 3) && n0 >= a.N)) continue;
        uint32_t o = 0;
        uint32_t rnd[4] = {0, 0, 0, 0};
        if (d.on && !d.mask_in && n0 < a.N) {
          if ((a.N & 3) == 0) {                  // the four outputs share one Philox block: slot = row * N + n, N % 4 == 0
            const qbnn::u32x4 r4 = qbnn::philox4x32_10((uint32_t)((row * a.N + n0) >> 2), d.layer_id, d.sample_begin + s, 1u, d.seed_lo, d.seed_hi);
            rnd[0] = r4.x; rnd[1] = r4.y; rnd[2] = r4.z; rnd[3] = r4.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int slot = row * a.N + n0 + e;
              const qbnn::u32x4 r4 = qbnn::philox4x32_10((uint32_t)(slot >> 2), d.layer_id, d.sample_begin + s, 1u, d.seed_lo, d.seed_hi);
              rnd[e] = (slot & 3) == 0 ? r4.x : ((slot & 3) == 1 ? r4.y : ((slot & 3) == 2 ? r4.z : r4.w));
            }
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = n0 + e;
          if (n >= a.N) continue;
          int v = requant(acc[i][j][4 * g4 + e] - a.q.z_w * R, n, a.q);
          if (d.on) {
            const float m = d.mask_in ? d.mask_in[((int64_t)s * a.B + row) * a.N + n] : keep_draw(rnd[e], d.keep);
            v = drop_apply(v, mask_from(m, d), d);
          }
          o |= (uint32_t)v << (8 * e);
        }
        if ((a.ldy & 3) == 0) {
          *reinterpret_cast<uint32_t*>(ys + (int64_t)row * a.ldy + n0) = o;
        } else {                                  // dense rows of a width that is no dword multiple (ldy == N): byte stores
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n0 + e < a.N) ys[(int64_t)row * a.ldy + n0 + e] = (uint8_t)(o >> (8 * e));
        }
      }
  }
}

// ---- k x k conv on a ONE-channel image + clamp + MaxPool2d(2,2): LeNet's first conv with per-sample (sampled) weights -----------------
// (bbb/models_bbb.py:120-133 layers.0 -> layers.1.)  With one input channel the whole receptive field of a pixel is k * k = 25 bytes:
// the centred patches [B][H * W][32] are built once per batch (qbnn_im2col5x5_c1; the network input is shared by the MC samples) and a
// 32-pixel tile of the conv is ONE MFMA against the sample's single weight fragment (<= 31 output channels + the ones row of the
// QBNN_LAYOUT_MFMA32 layout = the window sum).  A tile is 2 image rows x W / 2 columns, so the 2 x 2 pooling window of a pixel lives in
// lanes l, l ^ 1, l +- W / 2 of the same wave: the conv output never reaches memory, only the pooled map [S][B][H/2][W/2][COUT] does.
struct Conv1Args {
  const int8_t* patches; int64_t p_ss;      // [S | 1][B][H * W][32]
  const int8_t* w; int64_t w_ss;            // [S | 1][1024]: the layer's fragment tile
  uint8_t* y; int64_t y_ss;
  int B; Requant q; float vlo, vhi;
};

__device__ __forceinline__ int half_sum32(int v) {             // lane l gets v(l & 31) + v((l & 31) + 32)
  const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  return (int)r[0] + (int)r[1];
}

template <int W, int COUT>
__global__ __launch_bounds__(256) void conv_c1_pool_kernel(const Conv1Args a) {
  constexpr int TW = W / 2, NTILE = W;                         // W / 2 row pairs x 2 column halves
  static_assert(2 * TW <= 32 && (TW % 2) == 0 && (COUT % 4) == 0 && COUT < 32, "tile = 2 rows x W/2 columns of one wave half; channel groups of 4");
  constexpr int OJ = 4 * (COUT / 8) + (COUT & 3), OH = (COUT >> 2) & 1;      // accumulator register / lane half of the ones row (row COUT)
  const int s = blockIdx.y, b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const bool act = r < 2 * TW;
  const int pr = r / TW, pc = r - pr * TW;
  const v4i wf = *reinterpret_cast<const v4i*>(a.w + (int64_t)s * a.w_ss + lane * 16);
  const int8_t* pb = a.patches + (int64_t)s * a.p_ss + (int64_t)b * W * W * 32 + 16 * h;
  uint8_t* yb = a.y + (int64_t)s * a.y_ss + (int64_t)b * (W / 2) * (W / 2) * COUT + 4 * h;
  const float zy = (float)a.q.z_y;
  const int partner = pr == 0 ? lane + TW : lane - TW;         // the pixel one row down / up
  for (int t = wave; t < NTILE; t += 4) {
    const int i = t >> 1, c = t & 1;
    v4i xf = {0, 0, 0, 0};
    if (act) xf = *reinterpret_cast<const v4i*>(pb + ((2 * i + pr) * W + c * TW + pc) * 32);
    const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, xf, zero16, 0, 0, 0);
    const int zwr = a.q.z_w * half_sum32(h == OH ? acc[OJ] : 0);
    uint8_t* yo = yb + ((i * (W / 2)) + c * (TW / 2) + (pc >> 1)) * COUT;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      if (8 * g4 >= COUT) continue;                            // (a group beyond COUT on one half only: computed, never stored)
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float xv = (float)(acc[4 * g4 + k] - zwr);
        if (a.q.bias) xv = __builtin_fmaf(a.q.bias[min(8 * g4 + 4 * h + k, COUT - 1)], a.q.rcp, xv);
        v[k] = (med3f(xv * a.q.mult, a.vlo, a.vhi) + QBNN_MAGIC) + zy;
      }
      uint32_t d = pack_low_bytes(v[0], v[1], v[2], v[3]);
      d = max4_u7(d, (uint32_t)__shfl_xor((int)d, 1));
      d = max4_u7(d, (uint32_t)__shfl((int)d, partner));
      if (act && pr == 0 && (pc & 1) == 0 && 8 * g4 + 4 * h < COUT) *reinterpret_cast<uint32_t*>(yo + 8 * g4) = d;
    }
  }
}

__global__ __launch_bounds__(256) void im2col5x5_c1_kernel(const uint8_t* __restrict__ x, int B, int H, int W, int z_x, int8_t* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * H * W) return;
  const int b = (int)(idx / (H * W)), p = (int)(idx - (int64_t)b * H * W), oh = p / W, ow = p - oh * W;
  const uint8_t* xb = x + (int64_t)b * H * W;
  uint32_t wd[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
  for (int k = 0; k < 25; ++k) {
    const int ih = oh + k / 5 - 2, iw = ow + k % 5 - 2;
    const int v = ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) ? (int)xb[ih * W + iw] - z_x : 0;
    wd[k >> 2] |= (uint32_t)(v & 0xff) << (8 * (k & 3));
  }
  v4i* o = reinterpret_cast<v4i*>(out + idx * 32);
  o[0] = v4i{(int)wd[0], (int)wd[1], (int)wd[2], (int)wd[3]};
  o[1] = v4i{(int)wd[4], (int)wd[5], (int)wd[6], (int)wd[7]};
}

}  // namespace

QBNN_EXPORT int qbnn_im2col5x5_c1(const uint8_t* x, int32_t B, int32_t H, int32_t W, int32_t z_x, int8_t* out, void* stream) {
  if (!x || !out || B <= 0 || H <= 0 || W <= 0 || z_x < 0 || z_x > 127) return failf(QBNN_E_INVALID, "qbnn_im2col5x5_c1: bad argument");
  const int64_t n = (int64_t)B * H * W;
  hipLaunchKernelGGL(im2col5x5_c1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, B, H, W, z_x, out);
  return qbnn_check_launch_msg("qbnn_im2col5x5_c1");
}

QBNN_EXPORT int qbnn_conv_c1_pool_i8_mc(const int8_t* patches, int64_t patches_ss, const int8_t* w_packed, int64_t w_ss, const float* bias,
                                        uint8_t* y, int64_t y_ss, int32_t n_samples, const qbnn_conv_desc* d, void* stream) {
  if (!patches || !w_packed || !y || !d || n_samples <= 0 || d->B <= 0) return failf(QBNN_E_INVALID, "qbnn_conv_c1_pool_i8_mc: bad argument");
  if (d->has_bias && !bias) return failf(QBNN_E_INVALID, "qbnn_conv_c1_pool_i8_mc: has_bias set but bias is NULL");
  if ((patches_ss & 15) || (w_ss & 15) || (reinterpret_cast<uintptr_t>(patches) & 15) || (reinterpret_cast<uintptr_t>(w_packed) & 15) ||
      (reinterpret_cast<uintptr_t>(y) & 3) || (y_ss & 3))
    return failf(QBNN_E_INVALID, "qbnn_conv_c1_pool_i8_mc: operands must be 16-byte (y: 4-byte) aligned");
  if (d->relu) return failf(QBNN_E_INVALID, "qbnn_conv_c1_pool_i8_mc: plain conv only (the LeNet graphs have no ReLU behind their convs)");
  Conv1Args a;
  memset(&a, 0, sizeof(a));
  a.patches = patches; a.p_ss = patches_ss; a.w = w_packed; a.w_ss = w_ss; a.y = y; a.y_ss = y_ss; a.B = d->B;
  if (int rc = fill_requant(a.q, d, bias)) return rc;
  a.vlo = (float)(a.q.lo - a.q.z_y); a.vhi = (float)(a.q.hi - a.q.z_y);
  if (d->H == 28 && d->W == 28 && d->Cin == 1 && d->Cout == 20 && d->ksize == 5 && d->stride == 1 && d->pad == 2) {
    hipLaunchKernelGGL((conv_c1_pool_kernel<28, 20>), dim3((unsigned)d->B, (unsigned)n_samples), dim3(256), 0, (hipStream_t)stream, a);
    return qbnn_check_launch_msg("qbnn_conv_c1_pool_i8_mc");
  }
  return failf(QBNN_E_INVALID, "qbnn_conv_c1_pool_i8_mc: unsupported geometry (built: 28x28, 1 -> 20, 5x5, pad 2)");
}

QBNN_EXPORT int qbnn_conv_pool_drop_i8_mc(const uint8_t* x, int64_t x_ss, const int8_t* w_packed, int64_t w_ss, const float* bias,
                                          uint8_t* y, int64_t y_ss, int32_t ldy, int32_t n_samples, const qbnn_conv_desc* d,
                                          int32_t pool, const qbnn_dropout_desc* drop, const float* mask_in,
                                          const qbnn_dropout_desc* drop_in, const float* mask_in_in, float s_in, int32_t z_in,
                                          uint64_t seed, uint32_t sample_begin, void* stream) {
  if (!x || !w_packed || !y || !d || n_samples <= 0 || d->B <= 0) return failf(QBNN_E_INVALID, "qbnn_conv_pool_drop_i8_mc: bad argument");
  if (d->has_bias && !bias) return failf(QBNN_E_INVALID, "qbnn_conv_pool_drop_i8_mc: has_bias set but bias is NULL");
  if (d->has_res || d->x_is_centered_im2col) return failf(QBNN_E_INVALID, "qbnn_conv_pool_drop_i8_mc: no residual / im2col form");
  if ((x_ss & 3) || (reinterpret_cast<uintptr_t>(x) & 3) || (w_ss & 15) || (reinterpret_cast<uintptr_t>(w_packed) & 15) ||
      (ldy & 15) || (y_ss & 3) || (reinterpret_cast<uintptr_t>(y) & 3))
    return failf(QBNN_E_INVALID, "qbnn_conv_pool_drop_i8_mc: alignment (x 4, packed weights 16, output pitch 16 bytes)");
  ConvSmallArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.x_ss = x_ss; a.w = w_packed; a.w_ss = w_ss; a.y = y; a.y_ss = y_ss; a.ldy = ldy;
  a.B = d->B; a.n_samples = n_samples; a.z_x = d->z_x;
  int rc = fill_requant(a.q, d, bias);
  if (rc) return rc;
  a.vlo = (float)(a.q.lo - a.q.z_y); a.vhi = (float)(a.q.hi - a.q.z_y);
  if ((rc = fill_drop(a.d, d->s_y, d->z_y, d->a_hi, drop, mask_in, seed, sample_begin))) return rc;
  if ((rc = fill_drop(a.din, s_in, z_in, d->a_hi, drop_in, mask_in_in, seed, sample_begin))) return rc;
  if (drop_in && drop_in->z_m != d->z_x) return failf(QBNN_E_INVALID, "qbnn_conv_pool_drop_i8_mc: with drop_in, z_x must be the input dropout's zero point");
  const int ho = pool ? d->H / 2 : d->H;
  if (ldy < ho * ho * d->Cout) return failf(QBNN_E_INVALID, "qbnn_conv_pool_drop_i8_mc: ldy smaller than the flattened map");
  using LeNet2 = SmallCfg<14, 20, 5, 50, 4>;           // conv_lenet_*: layers.3 (20 -> 50, 5x5, pad 2) on the pooled 14 x 14 map
  if (d->H == 14 && d->W == 14 && d->Cin == 20 && d->Cout == 50 && d->ksize == 5 && d->stride == 1 && d->pad == 2)
    return launch_conv_small<LeNet2>(a, pool != 0, (hipStream_t)stream);
  return failf(QBNN_E_INVALID, "qbnn_conv_pool_drop_i8_mc: unsupported geometry (built: 14x14, 20 -> 50, 5x5, pad 2)");
}

QBNN_EXPORT int qbnn_linear_i8_mc(const uint8_t* x, int64_t x_ss, int32_t ldx, const int8_t* w_packed, int64_t w_ss, const float* bias,
                                  uint8_t* y, int64_t y_ss, int32_t ldy, int32_t n_samples, const qbnn_conv_desc* d,
                                  const qbnn_dropout_desc* drop, const float* mask_in, uint64_t seed, uint32_t sample_begin,
                                  void* stream) {
  if (!x || !w_packed || !y || !d || n_samples <= 0 || d->B <= 0 || d->Cin <= 0 || d->Cout <= 0)
    return failf(QBNN_E_INVALID, "qbnn_linear_i8_mc: bad argument");
  if (d->has_bias && !bias) return failf(QBNN_E_INVALID, "qbnn_linear_i8_mc: has_bias set but bias is NULL");
  if ((ldx & 15) || ldx < d->Cin || (x_ss & 15) || (reinterpret_cast<uintptr_t>(x) & 15) || (w_ss & 15) ||
      (reinterpret_cast<uintptr_t>(w_packed) & 15) || ldy < d->Cout ||
      ((ldy & 3) ? ldy != d->Cout : ((y_ss & 3) || (reinterpret_cast<uintptr_t>(y) & 3))))
    return failf(QBNN_E_INVALID, "qbnn_linear_i8_mc: alignment (x rows and packed weights 16 bytes; output pitch 4, or exactly N) or pitch < width");
  if ((int64_t)d->Cin > (1 << 16)) return failf(QBNN_E_INVALID, "qbnn_linear_i8_mc: K too large for int32 accumulation");
  LinearArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.x_ss = x_ss; a.ldx = ldx; a.w = w_packed; a.w_ss = w_ss; a.y = y; a.y_ss = y_ss; a.ldy = ldy;
  a.B = d->B; a.K = d->Cin; a.N = d->Cout; a.KS = (d->Cin + 31) / 32; a.z_x = d->z_x;
  int rc = fill_requant(a.q, d, bias);
  if (rc) return rc;
  if ((rc = fill_drop(a.d, d->s_y, d->z_y, d->a_hi, drop, mask_in, seed, sample_begin))) return rc;
  const dim3 grid((unsigned)((d->B + 127) / 128), (unsigned)((ldy + 127) / 128), (unsigned)n_samples);
  hipLaunchKernelGGL(linear_i8_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  return qbnn_check_launch_msg("qbnn_linear_i8_mc");
}
