// libqbnn_hip.so -- hand-written gfx950 (CDNA4) kernels + C ABI (include/qbnn.h).
// Compiled with:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC
// (-ffp-contract=off: bit-exactness with the reference's ATen/FBGEMM arithmetic depends on
//  where a multiply-add is fused and where it is not; every fma below is explicit.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/qbnn.h"
#include "qbnn_rng.cuh"

#define QBNN_EXPORT extern "C" __attribute__((visibility("default")))

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, const char* a = "", long b = 0, long c = 0) {
  snprintf(g_err, sizeof(g_err), fmt, a, b, c);
  return code;
}

static int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return QBNN_E_LAUNCH;
  }
  return QBNN_OK;
}

QBNN_EXPORT const char* qbnn_last_error(void) { return g_err; }
QBNN_EXPORT int qbnn_version(void) { return 1; }

// =====================================================================================
// Packed weight layout (QBNN_LAYOUT_MFMA32): the B operand of v_mfma_i32_32x32x32_i8.
//   tile (nt, ks) covers output channels [32 nt, 32 nt + 32) and k in [32 ks, 32 ks + 32)
//   lane l of the wave holds, as 16 consecutive bytes, W[n = 32 nt + (l & 31)][k = 32 ks + 16 (l >> 5) + j]
//   byte offset = ((nt * KS + ks) * 64 + l) * 16 + j ; pad entries (n >= cout or k >= K) are 0.
// =====================================================================================
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

QBNN_EXPORT size_t qbnn_packed_weight_bytes(int32_t cout, int32_t k, int32_t layout) {
  if (layout == QBNN_LAYOUT_ROWMAJOR) return ((size_t)cout * k + 15) / 16 * 16;
  return (size_t)ceil_div(cout, 32) * ceil_div(k, 32) * 1024;
}

QBNN_EXPORT int qbnn_pack_weights_host(const int8_t* src, int32_t cout, int32_t k, int32_t layout, int8_t* dst) {
  if (!src || !dst || cout <= 0 || k <= 0) return fail(QBNN_E_INVALID, "qbnn_pack_weights_host: bad argument%s");
  memset(dst, 0, qbnn_packed_weight_bytes(cout, k, layout));
  if (layout == QBNN_LAYOUT_ROWMAJOR) { memcpy(dst, src, (size_t)cout * k); return QBNN_OK; }
  const int KS = ceil_div(k, 32);
  for (int n = 0; n < cout; ++n)
    for (int kk = 0; kk < k; ++kk) {
      const int nt = n >> 5, col = n & 31, ks = kk >> 5, half = (kk >> 4) & 1, j = kk & 15;
      dst[(((size_t)nt * KS + ks) * 64 + half * 32 + col) * 16 + j] = src[(size_t)n * k + kk];
    }
  return QBNN_OK;
}

// =====================================================================================
// Weight sampler: one thread = one 16-byte chunk of the packed layout for one MC sample.
// HBM traffic: reads 2 B/weight (mu_q, sigma_q; L2-resident across samples), writes 1 B/weight/sample.
// =====================================================================================
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// rne of an fp32 that may be far outside the int range (clamp first: monotone, so the later
// integer clamp gives the same result as the reference's saturating conversion)
__device__ __forceinline__ int rne_sat(float v) {
  v = fminf(fmaxf(v, -1.0e9f), 1.0e9f);
  return __float2int_rn(v);
}

__device__ __forceinline__ int sample_one(int mu_q, int sigma_q, float eps, const qbnn_sample_params& p) {
  const int eps_q = clampi(rne_sat(eps * p.inv_noise_scale), -128, 127);
  const int prod = (sigma_q - p.z_sigma) * eps_q;
  const int t_q = clampi(p.z_mul + rne_sat((float)prod * p.mul_multiplier), -128, 127);
  const float dw = __builtin_fmaf(p.s_w, (float)mu_q, p.nzs_w);
  const float dt = __builtin_fmaf(p.s_mul, (float)t_q, p.nzs_mul);
  const int w_q = clampi(p.z_add + rne_sat((dw + dt) * p.inv_s_add), -128, 127);
  return clampi(w_q, p.w_lo, p.w_hi);
}

__global__ __launch_bounds__(256) void sample_weights_i8_kernel(
    const v4i* __restrict__ mu, const v4i* __restrict__ sigma, int cout, int K, int layout, int n_chunks,
    qbnn_sample_params p, uint32_t seed_lo, uint32_t seed_hi, uint32_t layer_id, uint32_t sample_begin,
    const float* __restrict__ eps_in, int8_t* __restrict__ w_out, int64_t w_sample_stride) {
  const int chunk = blockIdx.x * 256 + threadIdx.x;
  if (chunk >= n_chunks) return;
  const int s = blockIdx.y;
  int n, k0;
  if (layout == QBNN_LAYOUT_MFMA32) {
    const int KS = (K + 31) >> 5;
    const int lane = chunk & 63, tile = chunk >> 6;
    const int nt = tile / KS, ks = tile - nt * KS;
    n = nt * 32 + (lane & 31);
    k0 = ks * 32 + (lane >> 5) * 16;
  } else {
    const int flat = chunk * 16;           // row-major [cout][K] flat index
    n = flat / K; k0 = flat - n * K;       // may straddle rows: handled per element below
  }
  const v4i m4 = mu[chunk], s4 = sigma[chunk];
  int mw[4] = {m4.x, m4.y, m4.z, m4.w}, sw[4] = {s4.x, s4.y, s4.z, s4.w};
  uint32_t ow[4] = {0u, 0u, 0u, 0u};
  const int64_t total = (int64_t)cout * K;
  uint32_t cur_blk = 0xffffffffu;
  float nrm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    int64_t idx;
    bool valid;
    if (layout == QBNN_LAYOUT_MFMA32) {
      valid = (n < cout) && (k0 + j < K);
      idx = (int64_t)n * K + k0 + j;
    } else {
      idx = (int64_t)chunk * 16 + j;
      valid = idx < total;
    }
    if (valid) {
      float eps;
      if (eps_in) {
        eps = eps_in[(int64_t)s * total + idx];
      } else {
        const uint32_t blk = (uint32_t)(idx >> 2);
        if (blk != cur_blk) {
          cur_blk = blk;
          qbnn::normal4(qbnn::philox4x32_10(blk, layer_id, sample_begin + s, 0u, seed_lo, seed_hi), nrm);
        }
        const int l = (int)(idx & 3);
        eps = l == 0 ? nrm[0] : (l == 1 ? nrm[1] : (l == 2 ? nrm[2] : nrm[3]));
      }
      const int mu_q = (mw[j >> 2] << (24 - 8 * (j & 3))) >> 24;       // sign-extended byte j
      const int sg_q = (sw[j >> 2] << (24 - 8 * (j & 3))) >> 24;
      const int wq = sample_one(mu_q, sg_q, eps, p);
      ow[j >> 2] |= ((uint32_t)wq & 0xffu) << (8 * (j & 3));
    }
  }
  const v4i o4 = {(int)ow[0], (int)ow[1], (int)ow[2], (int)ow[3]};
  reinterpret_cast<v4i*>(w_out + (int64_t)s * w_sample_stride)[chunk] = o4;
}

QBNN_EXPORT int qbnn_sample_weights_i8(const int8_t* mu_packed, const int8_t* sigma_packed, int32_t cout, int32_t k,
                                       int32_t layout, const qbnn_sample_params* hp, uint64_t seed, uint32_t layer_id,
                                       uint32_t sample_begin, int32_t n_samples, const float* eps_in, int8_t* w_out,
                                       int64_t w_sample_stride, void* stream) {
  if (!mu_packed || !sigma_packed || !hp || !w_out || cout <= 0 || k <= 0 || n_samples <= 0)
    return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8: bad argument%s");
  if (layout != QBNN_LAYOUT_MFMA32 && layout != QBNN_LAYOUT_ROWMAJOR)
    return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8: unknown layout%s");
  const size_t bytes = qbnn_packed_weight_bytes(cout, k, layout);
  if ((size_t)w_sample_stride < bytes || (w_sample_stride & 15))
    return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8: w_sample_stride too small or not 16-byte aligned%s");
  const int n_chunks = (int)(bytes / 16);
  dim3 grid(ceil_div(n_chunks, 256), n_samples);
  hipLaunchKernelGGL(sample_weights_i8_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                     (const v4i*)mu_packed, (const v4i*)sigma_packed, cout, k, layout, n_chunks, *hp,
                     (uint32_t)seed, (uint32_t)(seed >> 32), layer_id, sample_begin, eps_in, w_out, w_sample_stride);
  return check_launch("qbnn_sample_weights_i8");
}

// =====================================================================================
// int8 implicit-GEMM convolution on v_mfma_i32_32x32x32_i8, whole images resident in LDS.
//
//   GEMM view: M = G images * HO*HO output pixels, N = COUT, K = KSZ*KSZ*CIN in (kh,kw,c) order.
//   A (activations): the workgroup stages G centred images x' = x_q - z_x (int8, zero halo) in LDS;
//       because (kw,c) is contiguous in NHWC, the K axis of one output pixel is KSZ runs of
//       KSZ*CIN bytes; a lane's 16-byte A fragment is two 8-byte pieces addressed independently.
//   B (weights): pre-packed fragments streamed from L2 (qbnn_sample_weights_i8 wrote them).
//   acc = sum x' * W_q ;  sum x' (W_q - z_w) = acc - z_w * R,  R = window sum of x' (dot4 on the A
//       fragments the wave already holds).
//   Epilogue: FBGEMM requantisation + clamp_activation (+ residual quantized::add + ReLU), written
//       to an LDS staging tile and stored to HBM as full 16-byte lines.
// =====================================================================================
struct ConvArgs {
  const uint8_t* x; int64_t x_ss;
  const int8_t* w; int64_t w_ss;
  const float* bias;
  const uint8_t* res; int64_t res_ss;
  uint8_t* y; int64_t y_ss;
  int B;
  int z_x, z_w, z_y, lo, a_hi;
  float rcp, mult;
  // residual add
  float s_y, nzs_y, s_r, nzs_r, inv_s_o;
  int z_o;
};

template <int CIN_, int COUT_, int KSZ_, int STRIDE_, int HIN_, int HALO_, int G_, int MB_, int NB_, bool PRESUB_>
struct ConvCfg {
  static constexpr int CIN = CIN_, COUT = COUT_, KSZ = KSZ_, STRIDE = STRIDE_, HIN = HIN_, HALO = HALO_;
  static constexpr int G = G_, MB = MB_, NB = NB_;
  static constexpr bool PRESUB = PRESUB_;
  static constexpr int PAD = (KSZ - 1) / 2;
  static constexpr int OFF0 = HALO - PAD;
  static constexpr int HO = HIN / STRIDE;
  static constexpr int TW = HIN + 2 * HALO;
  static constexpr int PITCH = TW * CIN;
  static constexpr int TILE_BYTES = (TW * TW * CIN + 15) / 16 * 16;
  static constexpr int ROWB = HIN * CIN;                  // bytes of one image row in HBM
  static constexpr int K = KSZ * KSZ * CIN;
  static constexpr int PIECES = K / 8;
  static constexpr int PPR = KSZ * CIN / 8;               // 8-byte pieces per kernel row
  static constexpr int KS = (K + 31) / 32;
  static constexpr int NT = (COUT + 31) / 32;
  static constexpr int M = G * HO * HO;
  static constexpr int MT = M / 32;
  static constexpr int MBLKS = MT / MB, NBLKS = NT / NB;
  static constexpr int NPASS = MBLKS * NBLKS;
  static constexpr int OUT_BYTES = (M * COUT + 15) / 16 * 16;
  static constexpr int RSCR_BYTES = 4 * MB * 32 * 4;
  static constexpr int LDS_BYTES = G * TILE_BYTES + OUT_BYTES + RSCR_BYTES;
  static_assert(K % 8 == 0, "K must be a multiple of 8");
  static_assert(ROWB % 16 == 0, "image rows must be 16-byte multiples");
  static_assert(M % 32 == 0 && MT % MB == 0 && NT % NB == 0, "tile blocking must divide the problem");
  static_assert((HO & (HO - 1)) == 0, "HO must be a power of two");
  static_assert((CIN % 8) == 0, "CIN must be a multiple of 8");
  static constexpr int piece_off(int p) { return p < PIECES ? (p / PPR) * PITCH + (p % PPR) * 8 : 0; }
  static constexpr bool piece_valid(int p) { return p < PIECES; }
};

// (x_q bytes) - z  for four packed bytes, no cross-byte borrow: x in [0,127], z in [0,127]
__device__ __forceinline__ uint32_t sub_bytes(uint32_t x, uint32_t z4) {
  return ((x | 0x80808080u) - z4) ^ 0x80808080u;
}

template <class C, bool HAS_RES>
__global__ __launch_bounds__(256) void conv_i8_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* tile = smem;
  uint8_t* outb = smem + C::G * C::TILE_BYTES;
  int* rscr = reinterpret_cast<int*>(outb + C::OUT_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.y;
  const int img0 = blockIdx.x * C::G;

  // ---- 1. zero the halo'd tiles (halo must read as x' = 0)
  if (C::HALO > 0) {
    v4i z = {0, 0, 0, 0};
    for (int i = tid; i < C::G * C::TILE_BYTES / 16; i += 256) reinterpret_cast<v4i*>(tile)[i] = z;
    __syncthreads();
  }
  // ---- 2. stage G images: HBM (16 B / lane, coalesced) -> centre -> LDS (2 x 8 B)
  {
    const uint8_t* xs = a.x + (int64_t)s * a.x_ss;
    constexpr int CPR = C::ROWB / 16;                     // 16-byte chunks per image row
    constexpr int CPI = C::HIN * CPR;                     // per image
    const uint32_t z4 = (uint32_t)a.z_x * 0x01010101u;
    for (int i = tid; i < C::G * CPI; i += 256) {
      const int g = i / CPI, rem = i - g * CPI;
      const int row = rem / CPR, within = rem - row * CPR;
      if (img0 + g < a.B) {
        v4i v = *reinterpret_cast<const v4i*>(xs + ((int64_t)(img0 + g) * C::HIN + row) * C::ROWB + within * 16);
        if (!C::PRESUB) {
          v.x = sub_bytes(v.x, z4); v.y = sub_bytes(v.y, z4); v.z = sub_bytes(v.z, z4); v.w = sub_bytes(v.w, z4);
        }
        uint8_t* d = tile + g * C::TILE_BYTES + (row + C::HALO) * C::PITCH + C::HALO * C::CIN + within * 16;
        *reinterpret_cast<v2i*>(d) = v2i{v.x, v.y};
        *reinterpret_cast<v2i*>(d + 8) = v2i{v.z, v.w};
      }
    }
  }
  // ---- 3. residual operand -> output staging tile (the epilogue updates it in place)
  if (HAS_RES) {
    const uint8_t* rs = a.res + (int64_t)s * a.res_ss;
    constexpr int IMG_OUT = C::HO * C::HO * C::COUT;
    for (int i = tid; i < C::M * C::COUT / 16; i += 256) {
      const int g = (i * 16) / IMG_OUT;
      if (img0 + g < a.B)
        reinterpret_cast<v4i*>(outb)[i] = *reinterpret_cast<const v4i*>(rs + (int64_t)img0 * IMG_OUT + (int64_t)i * 16);
    }
  }
  __syncthreads();

  // ---- 4. MFMA passes
  const int r = lane & 31, h = lane >> 5;
  const int8_t* wq = a.w + (int64_t)s * a.w_ss;
  int* myr = rscr + wave * C::MB * 32;
  for (int pass = wave; pass < C::NPASS; pass += 4) {
    const int mblk = pass / C::NBLKS, nblk = pass - mblk * C::NBLKS;
    int abase[C::MB];
#pragma unroll
    for (int mb = 0; mb < C::MB; ++mb) {
      const int m = (mblk * C::MB + mb) * 32 + r;
      const int g = m / (C::HO * C::HO), rem = m % (C::HO * C::HO);
      const int oh = rem / C::HO, ow = rem % C::HO;
      abase[mb] = g * C::TILE_BYTES + ((oh * C::STRIDE + C::OFF0) * C::TW + ow * C::STRIDE + C::OFF0) * C::CIN;
    }
    v16i acc[C::MB][C::NB];
    int rsum[C::MB];
#pragma unroll
    for (int mb = 0; mb < C::MB; ++mb) {
      rsum[mb] = 0;
#pragma unroll
      for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0;
    }
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      v4i bf[C::NB];
#pragma unroll
      for (int nb = 0; nb < C::NB; ++nb)
        bf[nb] = *reinterpret_cast<const v4i*>(wq + ((int64_t)((nblk * C::NB + nb) * C::KS + ks) * 64 + lane) * 16);
      const int o0 = h ? C::piece_off(4 * ks + 2) : C::piece_off(4 * ks + 0);
      const int o1 = h ? C::piece_off(4 * ks + 3) : C::piece_off(4 * ks + 1);
      // dot4 masks: pad pieces (k >= K) hold unrelated bytes and must not enter R
      const int m0 = h ? (C::piece_valid(4 * ks + 2) ? 0x01010101 : 0) : (C::piece_valid(4 * ks + 0) ? 0x01010101 : 0);
      const int m1 = h ? (C::piece_valid(4 * ks + 3) ? 0x01010101 : 0) : (C::piece_valid(4 * ks + 1) ? 0x01010101 : 0);
#pragma unroll
      for (int mb = 0; mb < C::MB; ++mb) {
        const v2i lo = *reinterpret_cast<const v2i*>(tile + abase[mb] + o0);
        const v2i hi = *reinterpret_cast<const v2i*>(tile + abase[mb] + o1);
        const v4i af = {lo.x, lo.y, hi.x, hi.y};
        int rs_ = rsum[mb];
        rs_ = __builtin_amdgcn_sdot4(lo.x, m0, rs_, false);
        rs_ = __builtin_amdgcn_sdot4(lo.y, m0, rs_, false);
        rs_ = __builtin_amdgcn_sdot4(hi.x, m1, rs_, false);
        rs_ = __builtin_amdgcn_sdot4(hi.y, m1, rs_, false);
        rsum[mb] = rs_;
#pragma unroll
        for (int nb = 0; nb < C::NB; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf[nb], acc[mb][nb], 0, 0, 0);
      }
    }
    // window sums: both k-halves, then transpose lane-row -> register-row through LDS
#pragma unroll
    for (int mb = 0; mb < C::MB; ++mb) {
      const int tot = rsum[mb] + __shfl_xor(rsum[mb], 32);
      if (h == 0) myr[mb * 32 + r] = tot;
    }
    // ---- epilogue
#pragma unroll
    for (int mb = 0; mb < C::MB; ++mb) {
      v4i rr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) rr[i] = *reinterpret_cast<const v4i*>(myr + mb * 32 + 8 * i + 4 * h);
#pragma unroll
      for (int nb = 0; nb < C::NB; ++nb) {
        const int n = (nblk * C::NB + nb) * 32 + r;
        const bool nvalid = n < C::COUT;
        const float bias = (a.bias && nvalid) ? a.bias[n] : 0.0f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
          const int m = (mblk * C::MB + mb) * 32 + row;
          const int av = acc[mb][nb][reg] - a.z_w * rr[reg >> 2][reg & 3];
          float xf = (float)av;
          if (a.bias) xf = __builtin_fmaf(bias, a.rcp, xf);
          int q = a.z_y + rne_sat(xf * a.mult);
          q = min(max(q, a.lo), 255);
          q = min(q, a.a_hi);
          if (nvalid) {
            uint8_t* o = outb + m * C::COUT + n;
            if (HAS_RES) {
              const float da = __builtin_fmaf(a.s_y, (float)q, a.nzs_y);
              const float db = __builtin_fmaf(a.s_r, (float)(int)(*o), a.nzs_r);
              int q2 = min(max(a.z_o + rne_sat((da + db) * a.inv_s_o), 0), 255);
              q2 = min(q2, a.a_hi);
              q2 = max(q2, a.z_o);
              q = q2;
            }
            *o = (uint8_t)q;
          }
        }
      }
    }
  }
  __syncthreads();
  // ---- 5. LDS staging tile -> HBM, 16 B / lane
  {
    uint8_t* ys = a.y + (int64_t)s * a.y_ss;
    constexpr int IMG_OUT = C::HO * C::HO * C::COUT;
    for (int i = tid; i < C::M * C::COUT / 16; i += 256) {
      const int g = (i * 16) / IMG_OUT;
      if (img0 + g < a.B)
        *reinterpret_cast<v4i*>(ys + (int64_t)img0 * IMG_OUT + (int64_t)i * 16) = reinterpret_cast<const v4i*>(outb)[i];
    }
  }
}

template <class C>
static int launch_conv(const ConvArgs& a, int n_samples, bool has_res, hipStream_t st) {
  static_assert(C::LDS_BYTES <= 160 * 1024, "LDS budget");
  dim3 grid(ceil_div(a.B, C::G), n_samples);
  if (has_res) {
    static bool attr_r = false;
    if (!attr_r) { hipFuncSetAttribute((const void*)conv_i8_kernel<C, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES); attr_r = true; }
    hipLaunchKernelGGL((conv_i8_kernel<C, true>), grid, dim3(256), C::LDS_BYTES, st, a);
  } else {
    static bool attr_n = false;
    if (!attr_n) { hipFuncSetAttribute((const void*)conv_i8_kernel<C, false>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES); attr_n = true; }
    hipLaunchKernelGGL((conv_i8_kernel<C, false>), grid, dim3(256), C::LDS_BYTES, st, a);
  }
  return check_launch("qbnn_conv2d_i8_mc");
}

//                      CIN COUT K  S  HIN HALO G  MB NB PRESUB
using Cfg_c0     = ConvCfg<32, 24, 1, 1, 32, 0, 1, 4, 1, true>;    // layers.0 on the im2col tensor (K = 27 -> 32)
using Cfg_24_24  = ConvCfg<24, 24, 3, 1, 32, 1, 1, 4, 1, false>;   // layers.3.*
using Cfg_24_48s = ConvCfg<24, 48, 3, 2, 32, 1, 1, 2, 2, false>;   // layers.4.0.stem.0
using Cfg_24_48p = ConvCfg<24, 48, 1, 2, 32, 1, 1, 2, 2, false>;   // layers.4.0.shortcut.0
using Cfg_48_48  = ConvCfg<48, 48, 3, 1, 16, 1, 2, 2, 2, false>;   // layers.4.*
using Cfg_48_96s = ConvCfg<48, 96, 3, 2, 16, 1, 2, 1, 3, false>;   // layers.5.0.stem.0
using Cfg_48_96p = ConvCfg<48, 96, 1, 2, 16, 1, 2, 1, 3, false>;   // layers.5.0.shortcut.0
using Cfg_96_96  = ConvCfg<96, 96, 3, 1, 8, 1, 4, 1, 3, false>;    // layers.5.*
using Cfg_96_192s = ConvCfg<96, 192, 3, 2, 8, 1, 4, 1, 3, false>;  // layers.6.0.stem.0
using Cfg_96_192p = ConvCfg<96, 192, 1, 2, 8, 1, 4, 1, 3, false>;  // layers.6.0.shortcut.0
using Cfg_192_192 = ConvCfg<192, 192, 3, 1, 4, 1, 4, 1, 3, false>; // layers.6.*

QBNN_EXPORT int qbnn_conv2d_i8_mc(const uint8_t* x, int64_t x_ss, const int8_t* w_packed, int64_t w_ss, const float* bias,
                                  const uint8_t* res, int64_t res_ss, uint8_t* y, int64_t y_ss, int32_t n_samples,
                                  const qbnn_conv_desc* d, void* stream) {
  if (!x || !w_packed || !y || !d || n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_mc: bad argument%s");
  if (d->has_res && !res) return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_mc: has_res set but res is NULL%s");
  if (d->has_bias && !bias) return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_mc: has_bias set but bias is NULL%s");
  if (d->H != d->W) return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_mc: only square inputs are supported%s");
  if (d->z_x < 0 || d->z_x > 127 || d->a_hi > 127 || d->a_hi < 1)
    return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_mc: activations must be <= 7 bit (reference quant_utils.py:120)%s");
  ConvArgs a;
  a.x = x; a.x_ss = x_ss; a.w = w_packed; a.w_ss = w_ss; a.bias = d->has_bias ? bias : nullptr;
  a.res = res; a.res_ss = res_ss; a.y = y; a.y_ss = y_ss; a.B = d->B;
  a.z_x = d->z_x; a.z_w = d->z_w; a.z_y = d->z_y; a.lo = d->relu ? d->z_y : 0; a.a_hi = d->a_hi;
  const float atw = d->s_x * d->s_w;     // qconv.cpp GetQuantizationParams: float * float
  a.rcp = 1.0f / atw;                    // FBGEMM act_times_w_rcp
  a.mult = atw / d->s_y;                 // output_multiplier_float
  a.s_y = d->s_y; a.nzs_y = (float)(-d->z_y) * d->s_y;
  a.s_r = d->s_r; a.nzs_r = (float)(-d->z_r) * d->s_r;
  a.inv_s_o = d->has_res ? 1.0f / d->s_o : 0.f; a.z_o = d->z_o;
  hipStream_t st = (hipStream_t)stream;
  const bool hr = d->has_res != 0;
#define QBNN_CASE(CFG, cin, cout, ks, sd, hin, im2c)                                                      \
  if (d->Cin == (cin) && d->Cout == (cout) && d->ksize == (ks) && d->stride == (sd) && d->H == (hin) &&   \
      d->pad == ((ks) - 1) / 2 && (d->x_is_centered_im2col != 0) == (im2c))                                \
    return launch_conv<CFG>(a, n_samples, hr, st);
  QBNN_CASE(Cfg_c0, 32, 24, 1, 1, 32, true)
  QBNN_CASE(Cfg_24_24, 24, 24, 3, 1, 32, false)
  QBNN_CASE(Cfg_24_48s, 24, 48, 3, 2, 32, false)
  QBNN_CASE(Cfg_24_48p, 24, 48, 1, 2, 32, false)
  QBNN_CASE(Cfg_48_48, 48, 48, 3, 1, 16, false)
  QBNN_CASE(Cfg_48_96s, 48, 96, 3, 2, 16, false)
  QBNN_CASE(Cfg_48_96p, 48, 96, 1, 2, 16, false)
  QBNN_CASE(Cfg_96_96, 96, 96, 3, 1, 8, false)
  QBNN_CASE(Cfg_96_192s, 96, 192, 3, 2, 8, false)
  QBNN_CASE(Cfg_96_192p, 96, 192, 1, 2, 8, false)
  QBNN_CASE(Cfg_192_192, 192, 192, 3, 1, 4, false)
#undef QBNN_CASE
  return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_mc: unsupported geometry%s Cin=%ld Cout=%ld", "", d->Cin, d->Cout);
}

// =====================================================================================
// Input quantisation, layer-0 im2col, head, MC reduction
// =====================================================================================
__global__ __launch_bounds__(256) void quantize_input_kernel(const float* __restrict__ x, int B, int Cc, int H, int W,
                                                             float inv, int z, int a_hi, uint8_t* __restrict__ out) {
  const int64_t n = (int64_t)B * Cc * H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    // i indexes the NHWC output
    const int c = (int)(i % Cc);
    int64_t t = i / Cc;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int b = (int)(t / H);
    const float v = x[(((int64_t)b * Cc + c) * H + h) * W + w];
    int q = min(max(z + rne_sat(v * inv), 0), 255);
    out[i] = (uint8_t)min(q, a_hi);
  }
}

QBNN_EXPORT int qbnn_quantize_input_nchw(const float* x, int32_t B, int32_t Cc, int32_t H, int32_t W, float scale,
                                         int32_t zp, int32_t a_hi, uint8_t* out, void* stream) {
  if (!x || !out || B <= 0 || Cc <= 0 || H <= 0 || W <= 0) return fail(QBNN_E_INVALID, "qbnn_quantize_input_nchw: bad argument%s");
  const int64_t n = (int64_t)B * Cc * H * W;
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(quantize_input_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, B, Cc, H, W,
                     1.0f / scale, zp, a_hi, out);
  return check_launch("qbnn_quantize_input_nchw");
}

__global__ __launch_bounds__(256) void im2col3x3_c3_kernel(const uint8_t* __restrict__ x, int B, int H, int W, int z_x,
                                                           int8_t* __restrict__ out) {
  const int64_t npix = (int64_t)B * H * W;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
    const int ow = (int)(p % W);
    const int oh = (int)((p / W) % H);
    const int64_t b = p / ((int64_t)W * H);
    uint32_t wds[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int8_t* by = reinterpret_cast<int8_t*>(wds);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = oh + kh - 1, iw = ow + kw - 1;
        const bool in = ih >= 0 && ih < H && iw >= 0 && iw < W;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          by[(kh * 3 + kw) * 3 + c] = in ? (int8_t)((int)x[((b * H + ih) * W + iw) * 3 + c] - z_x) : (int8_t)0;
      }
    v4i* o = reinterpret_cast<v4i*>(out + p * 32);
    o[0] = v4i{(int)wds[0], (int)wds[1], (int)wds[2], (int)wds[3]};
    o[1] = v4i{(int)wds[4], (int)wds[5], (int)wds[6], (int)wds[7]};
  }
}

QBNN_EXPORT int qbnn_im2col3x3_c3(const uint8_t* x, int32_t B, int32_t H, int32_t W, int32_t z_x, int8_t* out, void* stream) {
  if (!x || !out || B <= 0) return fail(QBNN_E_INVALID, "qbnn_im2col3x3_c3: bad argument%s");
  const int64_t n = (int64_t)B * H * W;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(im2col3x3_c3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, B, H, W, z_x, out);
  return check_launch("qbnn_im2col3x3_c3");
}

// head: one wave per (sample, image).  C <= 256 channels, N <= 64 classes.
struct HeadArgs {
  const uint8_t* x; int64_t x_ss;
  const int8_t* w; int64_t w_ss;
  const float* bias;
  float* probs;
  int B, kk, C, N;
  int z_x, z_w, z_y, a_hi;
  float inv_kk, rcp, mult, s_y;
};

__global__ __launch_bounds__(256) void head_i8_kernel(const HeadArgs a) {
  __shared__ int pooled[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  const int s = blockIdx.y;
  if (b >= a.B) return;
  const uint8_t* xs = a.x + (int64_t)s * a.x_ss + (int64_t)b * a.kk * a.C;
  // AvgPool2d(k) on quint8, channels-last: q = clamp(rne((sum - kk z) / kk) + z, 0, 255); then clamp_activation
  for (int c = lane; c < a.C; c += 64) {
    int sum = 0;
    for (int p = 0; p < a.kk; ++p) sum += xs[p * a.C + c];
    int q = min(max(rne_sat((float)(sum - a.kk * a.z_x) * a.inv_kk) + a.z_x, 0), 255);
    q = min(q, a.a_hi);
    pooled[wave][c] = q - a.z_x;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): same-wave LDS write -> read
  __builtin_amdgcn_wave_barrier();
  const int8_t* ws = a.w + (int64_t)s * a.w_ss;
  float logit = -INFINITY;
  if (lane < a.N) {
    int acc = 0;
    for (int c = 0; c < a.C; ++c) acc += pooled[wave][c] * ((int)ws[lane * a.C + c] - a.z_w);
    float xf = (float)acc;
    if (a.bias) xf = __builtin_fmaf(a.bias[lane], a.rcp, xf);
    int q = min(max(a.z_y + rne_sat(xf * a.mult), 0), 255);
    q = min(q, a.a_hi);
    logit = (float)(q - a.z_y) * a.s_y;     // DeQuantStub
  }
  float mx = logit;
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float e = lane < a.N ? expf(logit - mx) : 0.f;
  float sum = e;
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if (lane < a.N) a.probs[((int64_t)s * a.B + b) * a.N + lane] = e / sum;
}

QBNN_EXPORT int qbnn_head_i8_mc(const uint8_t* x, int64_t x_ss, const int8_t* w, int64_t w_ss, const float* bias,
                                float* probs, int32_t n_samples, const qbnn_head_desc* d, void* stream) {
  if (!x || !w || !probs || !d || n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_head_i8_mc: bad argument%s");
  if (d->C > 256 || d->N > 64 || d->C <= 0 || d->N <= 0) return fail(QBNN_E_INVALID, "qbnn_head_i8_mc: C <= 256 and N <= 64 required%s");
  HeadArgs a;
  a.x = x; a.x_ss = x_ss; a.w = w; a.w_ss = w_ss; a.bias = d->has_bias ? bias : nullptr; a.probs = probs;
  a.B = d->B; a.kk = d->k * d->k; a.C = d->C; a.N = d->N;
  a.z_x = d->z_x; a.z_w = d->z_w; a.z_y = d->z_y; a.a_hi = d->a_hi;
  a.inv_kk = 1.0f / (float)(d->k * d->k);
  const float atw = d->s_x * d->s_w;
  a.rcp = 1.0f / atw; a.mult = atw / d->s_y; a.s_y = d->s_y;
  hipLaunchKernelGGL(head_i8_kernel, dim3(ceil_div(d->B, 4), n_samples), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("qbnn_head_i8_mc");
}

__global__ __launch_bounds__(256) void reduce_moments_kernel(const float* __restrict__ probs, int S, int64_t n,
                                                             int accumulate, float* __restrict__ mom) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s1 = accumulate ? mom[i] : 0.f, s2 = accumulate ? mom[n + i] : 0.f;
  for (int s = 0; s < S; ++s) {
    const float p = probs[(int64_t)s * n + i];
    s1 += p;
    s2 += p * p;
  }
  mom[i] = s1; mom[n + i] = s2;
}

QBNN_EXPORT int qbnn_reduce_moments(const float* probs, int32_t S, int64_t n, int32_t accumulate, float* mom, void* stream) {
  if (!probs || !mom || S <= 0 || n <= 0) return fail(QBNN_E_INVALID, "qbnn_reduce_moments: bad argument%s");
  hipLaunchKernelGGL(reduce_moments_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, probs, S, n, accumulate, mom);
  return check_launch("qbnn_reduce_moments");
}
