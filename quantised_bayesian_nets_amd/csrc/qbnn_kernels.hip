// libqbnn_hip.so -- hand-written gfx950 (CDNA4) kernels + C ABI (include/qbnn.h).
// Compiled with:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC
// (-ffp-contract=off: bit-exactness with the reference's ATen/FBGEMM arithmetic depends on
//  where a multiply-add is fused and where it is not; every fma below is explicit.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
#include <atomic>
#include <type_traits>

#include "../../include/qbnn.h"
#include "qbnn_rng.cuh"
#include "qbnn_common.h"
#define QBNN_EPS_TABLE_QUALIFIER __device__ static const
#include "qbnn_eps_table.h"

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

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: `done` keeps one bit per device ordinal
// (set after a successful call), so a process that runs models on several GPUs, or from several threads, sets it on each.
static int ensure_dyn_lds(const void* fn, std::atomic<uint64_t>& done, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) { snprintf(g_err, sizeof(g_err), "hipGetDevice: %s", hipGetErrorString(e)); return QBNN_E_LAUNCH; }
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return QBNN_OK;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d): %s", bytes, hipGetErrorString(e));
    return QBNN_E_LAUNCH;
  }
  done.fetch_or(bit, std::memory_order_release);
  return QBNN_OK;
}

// Device noise source (see include/qbnn.h: qbnn_set_device_noise_source): while non-null on the calling thread, every sampler /
// dropout launch reads (seed_lo, seed_hi, sample_begin) from this device address instead of from its kernel arguments.
static thread_local const unsigned int* g_noise_dev = nullptr;
const unsigned int* qbnn_noise_dev() { return g_noise_dev; }
QBNN_EXPORT int qbnn_set_device_noise_source(const uint32_t* dev_seed3) { g_noise_dev = dev_seed3; return QBNN_OK; }

int qbnn_ensure_dyn_lds(const void* fn, std::atomic<uint64_t>* done, int bytes) { return ensure_dyn_lds(fn, *done, bytes); }
int qbnn_fail_msg(int code, const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return code; }
int qbnn_check_launch_msg(const char* what) { return check_launch(what); }

QBNN_EXPORT const char* qbnn_last_error(void) { return g_err; }
QBNN_EXPORT int qbnn_version(void) { return 1; }

// =====================================================================================
// Packed weight layout (QBNN_LAYOUT_MFMA32): the weight operand of v_mfma_i32_32x32x32_i8.
//   logical matrix [cout][k], k = kh * krow + j, j = kw * Cin + c  (krow = KW * Cin bytes per kernel row)
//   packed K axis: every kernel row is padded to RBP = roundup32(krow) bytes: kp = kh * RBP + j   (pads are 0), so
//       that inside a row the activation bytes of one k-step are 32 contiguous LDS bytes (immediate offsets);
//   tile (nt, ks) covers output channels [32 nt, 32 nt + 32) and kp in [32 ks, 32 ks + 32);
//   lane l of the wave holds, as 16 consecutive bytes, W[n = 32 nt + (l & 31)][kp = 32 ks + 16 (l >> 5) + b];
//   byte offset = ((nt * KS + ks) * 64 + l) * 16 + b.
//   Ragged cout (cout % 32 != 0): row n = cout is the "ones row" (1 at every valid k, 0 at pads); the MFMA then
//       delivers the activation window sum R in that output row for free.
// =====================================================================================
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

struct PackGeom { int cout, k, krow, rows, rbp, kp, KS, NT; };
static inline PackGeom pack_geom(int cout, int k, int krow) {
  PackGeom g;
  g.cout = cout; g.k = k; g.krow = krow; g.rows = k / krow; g.rbp = ceil_div(krow, 32) * 32;
  g.kp = g.rows * g.rbp; g.KS = g.kp / 32; g.NT = ceil_div(cout, 32);
  return g;
}

QBNN_EXPORT size_t qbnn_packed_weight_bytes(int32_t cout, int32_t k, int32_t krow, int32_t layout) {
  if (layout == QBNN_LAYOUT_ROWMAJOR) return ((size_t)cout * k + 15) / 16 * 16;
  if (krow <= 0 || k % krow) return 0;
  const PackGeom g = pack_geom(cout, k, krow);
  return (size_t)g.NT * g.KS * 1024;
}

QBNN_EXPORT int qbnn_pack_weights_host(const int8_t* src, int32_t cout, int32_t k, int32_t krow, int32_t layout, int8_t* dst) {
  if (!src || !dst || cout <= 0 || k <= 0) return fail(QBNN_E_INVALID, "qbnn_pack_weights_host: bad argument%s");
  if (layout == QBNN_LAYOUT_ROWMAJOR) {
    memset(dst, 0, qbnn_packed_weight_bytes(cout, k, krow, layout));
    memcpy(dst, src, (size_t)cout * k);
    return QBNN_OK;
  }
  if (krow <= 0 || k % krow) return fail(QBNN_E_INVALID, "qbnn_pack_weights_host: k must be a multiple of krow%s");
  const PackGeom g = pack_geom(cout, k, krow);
  memset(dst, 0, (size_t)g.NT * g.KS * 1024);
  const bool ones = (cout % 32) != 0;
  for (int n = 0; n < cout + (ones ? 1 : 0); ++n)
    for (int kk = 0; kk < k; ++kk) {
      const int kp = (kk / krow) * g.rbp + kk % krow;
      const int nt = n >> 5, col = n & 31, ks = kp >> 5, half = (kp >> 4) & 1, b = kp & 15;
      dst[(((size_t)nt * g.KS + ks) * 64 + half * 32 + col) * 16 + b] = n < cout ? src[(size_t)n * k + kk] : (int8_t)1;
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

// eps_q drawn directly: one 32-bit Philox word through the alias table of clamp(rne(N(0,1) / s_n), -128, 127)
// (qbnn_eps_table.h; `tab` = the workgroup's LDS copy).  Integer compare only: the same bits as oracle/qbnn_oracle.c.
__device__ __forceinline__ int eps_q_from_u32(uint32_t u, const uint32_t* tab) {
  const uint32_t e = tab[u >> 24];
  return (int)(((u & 0xffffffu) < (e >> 8)) ? (u >> 24) : (e & 0xffu)) - 128;
}
__device__ __forceinline__ void load_eps_table(uint32_t* tab, int tid) {       // 256 threads: one entry each
  tab[tid & 255] = QBNN_EPS_ALIAS[tid & 255];
}

__device__ __forceinline__ int sample_one_q(int mu_q, int sigma_q, int eps_q, const qbnn_sample_params& p);
// injected fp32 eps (parity mode): quantise it as the reference does, then the common chain
__device__ __forceinline__ int sample_one(int mu_q, int sigma_q, float eps, const qbnn_sample_params& p) {
  return sample_one_q(mu_q, sigma_q, clampi(rne_sat(eps * p.inv_noise_scale), -128, 127), p);
}
__device__ __forceinline__ int sample_one_q(int mu_q, int sigma_q, int eps_q, const qbnn_sample_params& p) {
  const int prod = (sigma_q - p.z_sigma) * eps_q;
  const int t_q = clampi(p.z_mul + rne_sat((float)prod * p.mul_multiplier), -128, 127);
  const float dw = __builtin_fmaf(p.s_w, (float)mu_q, p.nzs_w);
  const float dt = __builtin_fmaf(p.s_mul, (float)t_q, p.nzs_mul);
  const int w_q = clampi(p.z_add + rne_sat((dw + dt) * p.inv_s_add), -128, 127);
  return clampi(w_q, p.w_lo, p.w_hi);
}

__global__ __launch_bounds__(256) void sample_weights_i8_kernel(
    const v4i* __restrict__ mu, const v4i* __restrict__ sigma, int cout, int K, int krow, int rbp, int KS, int layout,
    int n_chunks, qbnn_sample_params p, uint32_t seed_lo, uint32_t seed_hi, uint32_t layer_id, uint32_t sample_begin,
    const float* __restrict__ eps_in, int8_t* __restrict__ w_out, int64_t w_sample_stride, const uint32_t* __restrict__ nd) {
  if (nd) { seed_lo = nd[0]; seed_hi = nd[1]; sample_begin = nd[2]; }      // captured-graph mode: the seed lives in device memory
  __shared__ uint32_t eps_tab[256];
  load_eps_table(eps_tab, threadIdx.x);
  __syncthreads();
  const int chunk = blockIdx.x * 256 + threadIdx.x;
  if (chunk >= n_chunks) return;
  const int s = blockIdx.y;
  int n = 0, kh = 0, j0 = 0;       // MFMA32: output row, kernel row, first byte within the (padded) kernel row
  if (layout == QBNN_LAYOUT_MFMA32) {
    const int lane = chunk & 63, tile = chunk >> 6;
    const int nt = tile / KS, ks = tile - nt * KS;
    n = nt * 32 + (lane & 31);
    const int kp0 = ks * 32 + (lane >> 5) * 16;
    kh = kp0 / rbp; j0 = kp0 - kh * rbp;
  }
  const bool ones_row = (layout == QBNN_LAYOUT_MFMA32) && (cout & 31) && n == cout;
  const v4i m4 = mu[chunk], s4 = sigma[chunk];
  int mw[4] = {m4.x, m4.y, m4.z, m4.w}, sw[4] = {s4.x, s4.y, s4.z, s4.w};
  uint32_t ow[4] = {0u, 0u, 0u, 0u};
  const int64_t total = (int64_t)cout * K;
  uint32_t cur_blk = 0xffffffffu;
  qbnn::u32x4 rb = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    int64_t idx;
    bool valid;
    if (layout == QBNN_LAYOUT_MFMA32) {
      valid = (n < cout) && (j0 + j < krow);
      idx = (int64_t)n * K + kh * krow + j0 + j;
      if (ones_row && j0 + j < krow) ow[j >> 2] |= 1u << (8 * (j & 3));
    } else {
      idx = (int64_t)chunk * 16 + j;
      valid = idx < total;
    }
    if (valid) {
      const int mu_q = (mw[j >> 2] << (24 - 8 * (j & 3))) >> 24;       // sign-extended byte j
      const int sg_q = (sw[j >> 2] << (24 - 8 * (j & 3))) >> 24;
      int wq;
      if (eps_in) {
        wq = sample_one(mu_q, sg_q, eps_in[(int64_t)s * total + idx], p);
      } else {
        const uint32_t blk = (uint32_t)(idx >> 2);
        if (blk != cur_blk) {
          cur_blk = blk;
          rb = qbnn::philox4x32_10(blk, layer_id, sample_begin + s, 0u, seed_lo, seed_hi);
        }
        const int l = (int)(idx & 3);
        const uint32_t u = l == 0 ? rb.x : (l == 1 ? rb.y : (l == 2 ? rb.z : rb.w));
        wq = sample_one_q(mu_q, sg_q, eps_q_from_u32(u, eps_tab), p);
      }
      ow[j >> 2] |= ((uint32_t)wq & 0xffu) << (8 * (j & 3));
    }
  }
  const v4i o4 = {(int)ow[0], (int)ow[1], (int)ow[2], (int)ow[3]};
  reinterpret_cast<v4i*>(w_out + (int64_t)s * w_sample_stride)[chunk] = o4;
}

QBNN_EXPORT int qbnn_sample_weights_i8(const int8_t* mu_packed, const int8_t* sigma_packed, int32_t cout, int32_t k,
                                       int32_t krow, int32_t layout, const qbnn_sample_params* hp, uint64_t seed,
                                       uint32_t layer_id, uint32_t sample_begin, int32_t n_samples, const float* eps_in,
                                       int8_t* w_out, int64_t w_sample_stride, void* stream) {
  if (!mu_packed || !sigma_packed || !hp || !w_out || cout <= 0 || k <= 0 || n_samples <= 0)
    return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8: bad argument%s");
  if (layout != QBNN_LAYOUT_MFMA32 && layout != QBNN_LAYOUT_ROWMAJOR)
    return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8: unknown layout%s");
  if (layout == QBNN_LAYOUT_MFMA32 && (krow <= 0 || k % krow))
    return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8: k must be a multiple of krow%s");
  const size_t bytes = qbnn_packed_weight_bytes(cout, k, krow, layout);
  if ((size_t)w_sample_stride < bytes || (w_sample_stride & 15))
    return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8: w_sample_stride too small or not 16-byte aligned%s");
  const int n_chunks = (int)(bytes / 16);
  PackGeom g = pack_geom(cout, k, layout == QBNN_LAYOUT_MFMA32 ? krow : k);
  dim3 grid(ceil_div(n_chunks, 256), n_samples);
  hipLaunchKernelGGL(sample_weights_i8_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                     (const v4i*)mu_packed, (const v4i*)sigma_packed, cout, k, g.krow, g.rbp, g.KS, layout, n_chunks, *hp,
                     (uint32_t)seed, (uint32_t)(seed >> 32), layer_id, sample_begin, eps_in, w_out, w_sample_stride, g_noise_dev);
  return check_launch("qbnn_sample_weights_i8");
}

// =====================================================================================
// int8 implicit-GEMM convolution on v_mfma_i32_32x32x32_i8, whole images resident in LDS.
//
//   GEMM view: D[channel][pixel] = sum_k W[channel][k] * X'[k][pixel]
//       A operand = sampled weights (rows = output channels), pre-packed fragments streamed from L2;
//       B operand = activations: the workgroup stages G centred images x' = x_q - z_x (int8, zero halo) in LDS;
//           K is ordered (kh,kw,c), so with NHWC tiles the K axis of one output pixel is KSZ runs of KSZ*CIN
//           contiguous bytes; a lane's 16-byte fragment is two independently addressed 8-byte pieces.
//   Result layout: lane l owns pixel (l & 31) of the 32-pixel tile and, in registers 4g..4g+3, the four consecutive
//       channels 8g + 4(l>>5) + {0..3}: one dword of NHWC output per register group.
//   sum x' (W_q - z_w) = acc - z_w * R, R = window sum of x' = dot4 over the fragments the lane already holds
//       (+ the other k-half from lane l^32).
//   Epilogue functors implement FBGEMM requantisation + clamp_activation (+ quantized::add + ReLU) and write packed
//       dwords either to a dense quint8 staging buffer (stored to HBM as full 16-byte lines) or, centred, into the
//       halo'd LDS tile that feeds the next conv of a fused block.
// =====================================================================================
struct QConv {             // one conv layer's scalars (by value in kernel arguments)
  const int8_t* w; int64_t w_ss;   // sampled weights: base, per-MC-sample stride
  const float* bias;               // fp32 [COUT] or null
  int z_x, z_w, z_y;
  float rcp, mult;                 // FBGEMM act_times_w_rcp, output multiplier
  float vlo, vhi;                  // clamp of v = xf*mult before rounding: lo - z_y, min(255, a_hi) - z_y
  float s_y, nzs_y;                // output qparams as a quantized::add operand
  // ATen dequantises an add operand as fma(s, (float)q, nzs), nzs = rn(-z * s).  With q = q' + z (q' the centred integer the
  // epilogue holds):  s q + nzs = s q' + (s z + nzs), and dl = s z + nzs is the (negated) rounding error of the product z * s,
  // which is exactly representable: fma(s, q', dl) rounds the same real number once -> the same bits, one add fewer.
  float dl_y;
};
struct QAdd {              // BasicBlock Add + ReLU (models_bbb.py:179-182)
  float s_r, nzs_r; int z_r;       // residual operand qparams
  float dl_r;                      // s_r z_r + nzs_r exactly (see QConv::dl_y): dequantises the CENTRED residual byte directly
  float inv_s_o; int z_o;          // add output qparams
  float vhi;                       // min(255, a_hi) - z_o ; lower bound is 0 (ReLU: q >= z_o)
};

#define QBNN_MAGIC 12582912.0f     // 1.5 * 2^23: (v + MAGIC) has rne(v) in its low mantissa bits for |v| < 2^22

// Diagnostic build only (-DQBNN_STAMP, scratch library): per-phase s_memtime sums of wave 0, written to a debug
// buffer that nothing else reads.  The shipped library contains none of this.
#ifdef QBNN_STAMP
static unsigned long long* g_stamp_buf = nullptr;
__device__ unsigned long long* g_stamp_dev_ptr = nullptr;
#define g_stamp_dev g_stamp_dev_ptr
QBNN_EXPORT void qbnn_debug_stamp_buffer(void* p) { g_stamp_buf = (unsigned long long*)p; hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_dev_ptr), &p, sizeof(p)); }
__device__ unsigned long long g_inner[4];
QBNN_EXPORT void qbnn_debug_read_inner(unsigned long long* host4) {
  hipMemcpyFromSymbol(host4, HIP_SYMBOL(g_inner), 32);
  unsigned long long z[4] = {0, 0, 0, 0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_inner), z, 32);
}
#define QBNN_STAMP_DECL unsigned long long st_prev = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define QBNN_STAMP_START() do { __builtin_amdgcn_sched_barrier(0); st_prev = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_sched_barrier(0); } while (0)
#define QBNN_STAMP_AT(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xc07f); st_acc[i] += t_ - st_prev; st_prev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define QBNN_STAMP_DECL
#define QBNN_STAMP_START() do {} while (0)
#define QBNN_STAMP_AT(i) do {} while (0)
#endif

#ifdef QBNN_STAMP
#define QBNN_INNER_T0() unsigned long long it_ = 0, ia_[4] = {0, 0, 0, 0}; do { __builtin_amdgcn_sched_barrier(0); it_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_sched_barrier(0); } while (0)
#define QBNN_INNER_AT(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xc07f); ia_[i] += t_ - it_; it_ = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define QBNN_INNER_FLUSH() do { if (lane == 0 && wave == QBNN_STAMP_WAVE) for (int q_ = 0; q_ < 4; ++q_) atomicAdd(&g_inner[q_], ia_[q_]); } while (0)
#ifndef QBNN_STAMP_WAVE
#define QBNN_STAMP_WAVE 7
#endif
#else
#define QBNN_INNER_T0() do {} while (0)
#define QBNN_INNER_AT(i) do {} while (0)
#define QBNN_INNER_FLUSH() do {} while (0)
#endif

#ifndef QBNN_WDEPTH
#define QBNN_WDEPTH 5
#endif
template <int CIN_, int COUT_, int KSZ_, int STRIDE_, int HIN_, int HALO_, int G_, int MB_, int NB_, bool RING_ = true, int SLAB_KB_ = 36, int PADB_ = 0>
struct ConvCfg {
  // PADB: pad bytes after every pixel of the LDS input tile.  With 96 / 192 channels the 32 pixels of a B-operand
  // fragment sit 96 / 192 bytes apart = 4- / 8-way bank conflicts on every fragment read; +16 bytes makes the
  // stride 28 / 52 banks (2-way, like the 48-channel tiles).  Taps are then addressed one by one (CIN % 32 == 0).
  static constexpr int PADB = PADB_;
  // depth (k-steps) of the weight register ring of conv_passes_stream; 0 = the chunked double-buffer form
  static constexpr int WDEPTH = (!RING_ && CIN_ >= 48) ? QBNN_WDEPTH : 0;
  // RING: fused kernels stage this conv's weights through the LDS slab ring (conv_lds); false = every wave streams
  // its fragments from L2 (conv_passes) -- better when the conv's weights are far larger than the ring (192 channels)
  static constexpr bool RING = RING_;
  static constexpr int CIN = CIN_, COUT = COUT_, KSZ = KSZ_, STRIDE = STRIDE_, HIN = HIN_, HALO = HALO_;
  static constexpr int G = G_, MB = MB_, NB = NB_;
  static constexpr int PAD = (KSZ - 1) / 2;
  static constexpr int OFF0 = HALO - PAD;
  static constexpr int HO = HIN / STRIDE;
  static constexpr int TW = HIN + 2 * HALO;
  static constexpr int PIXB = CIN + PADB;                 // bytes from one tile pixel to the next
  static constexpr int PITCH = TW * PIXB;
  static constexpr int TILE_BYTES = (TW * TW * PIXB + 15) / 16 * 16;
  static constexpr int ROWB = HIN * CIN;                  // bytes of one image row in HBM
  static constexpr int RB = KSZ * CIN;                    // bytes of one kernel row: (kw, c) contiguous in NHWC
  static constexpr int RBP = (RB + 31) / 32 * 32;         // padded to whole 32-byte k-steps (weights are 0 there)
  static constexpr int SPR = RBP / 32;                    // k-steps per kernel row
  static constexpr int KS = KSZ * SPR;
  // k-steps per unrolled chunk: the largest divisor of a kernel row that keeps <= 12 fragments (weights + pixels) per
  // buffer of the double-buffered K loop (2 x 48 VGPRs) -- more spills the fused kernels
  static constexpr int pick_chunk() { int best = 1; for (int d = 1; d <= SPR; ++d) if (SPR % d == 0 && d * (NB_ + MB_) <= 12) best = d; return best; }
  static constexpr int KCHUNK = pick_chunk();
  static constexpr int NT = (COUT + 31) / 32;
  static constexpr bool USE_ONES = (COUT % 32) != 0;      // window sum from the packed layout's ones row
  static constexpr int ONES_TILE = COUT / 32, ONES_REG = 4 * ((COUT % 32) / 8);
  static constexpr int M = G * HO * HO;
  static constexpr int MT = M / 32;
  static constexpr int MBLKS = MT / MB, NBLKS = NT / NB;
  static constexpr int NPASS = MBLKS * NBLKS;
  static constexpr int OUT_BYTES = (M * COUT + 15) / 16 * 16;
  // 32-pixel-wide maps, 3x3/s1, one n-tile: an M-tile is one output row, so the input-row fragments of a pass
  // are shared by the 3 output rows that touch them and all weights fit in registers
  static constexpr bool ROWREUSE = (HO == 32 && STRIDE == 1 && KSZ == 3 && NT == 1 && NB_ == 1 && KS <= 9 && (32 % MB_) == 0);
  // weight slab of the LDS ring: SLK k-steps x all NT tiles, <= SLAB_KB KiB, SLK | KS
  static constexpr int pick_slab() { int best = 1; for (int d = 1; d <= KS; ++d) if (KS % d == 0 && d * NT <= SLAB_KB_) best = d; return best; }
  static constexpr int SLK = pick_slab();
  static constexpr int NSLAB = KS / SLK;
  static constexpr int SLAB_BYTES = NT * SLK * 1024;
  static constexpr int TILE_SLACK = 32;                   // the last k-step of a row over-reads < 32 bytes
  static_assert(ROWB % 16 == 0, "image rows must be 16-byte multiples");
  static_assert(M % 32 == 0 && MT % MB == 0 && NT % NB == 0, "tile blocking must divide the problem");
  static_assert((HO & (HO - 1)) == 0, "HO must be a power of two");
  static_assert((CIN % 8) == 0 && (COUT % 8) == 0, "channels must be multiples of 8");
  static_assert(COUT % 32 == 0 || NBLKS == 1, "ragged COUT needs all n-tiles (incl. the ones row) in one pass");
  static_assert(SPR % KCHUNK == 0, "k-chunks must not straddle kernel rows");
  // byte offset (from the lane's base) of k-step ks, and validity of the 8-byte piece `i` of k-half `h`
  static constexpr int SPT = CIN / 32;                    // k-steps per tap (padded tiles only)
  static constexpr int step_off(int ks) {
    return PADB == 0 ? (ks / SPR) * PITCH + (ks % SPR) * 32
                     : (ks / SPR) * PITCH + ((ks % SPR) / (SPT > 0 ? SPT : 1)) * PIXB + ((ks % SPR) % (SPT > 0 ? SPT : 1)) * 32;
  }
  // offset of the first k-step of weight slab `slab`; valid when slabs are whole kernel rows or whole taps, so that
  // step_off(slab * SLK + j) == slab_off(slab) + step_off(j)
  static constexpr int slab_off(int slab) { return step_off(slab * SLK); }
  static constexpr bool SLAB_ALIGNED = (SLK % SPR == 0) || (PADB_ > 0 && SPR % SLK == 0 && (CIN_ / 32) % SLK == 0);
  // offset inside a tile row of the 16-byte chunk `within` of an image row (chunks never straddle pixels when padded)
  static constexpr int row_chunk_off(int within) {
    return PADB == 0 ? HALO * CIN + within * 16 : (HALO + (within * 16) / CIN) * PIXB + (within * 16) % CIN;
  }
  static_assert(PADB == 0 || (CIN % 32 == 0 && PADB % 8 == 0 && (CIN / 32) % KCHUNK == 0), "padded tiles: whole k-steps per tap, chunks inside a tap");
  static constexpr bool piece_valid(int ks, int h, int i) { return (ks % SPR) * 32 + 16 * h + 8 * i < RB; }
};

// per-byte (x - z) for x in [0,127], z in [0,127]: no cross-byte borrow
__device__ __forceinline__ uint32_t sub_bytes(uint32_t x, uint32_t z4) {
  return ((x | 0x80808080u) - z4) ^ 0x80808080u;
}
// per-byte (x' + z) for the inverse map (result in [0,127])
__device__ __forceinline__ uint32_t add_bytes(uint32_t a, uint32_t z4) {
  return ((a & 0x7f7f7f7fu) + z4) ^ (a & 0x80808080u);
}
// low bytes of four fp32 bit patterns -> one dword (channel c0 in byte 0)
__device__ __forceinline__ uint32_t pack_low_bytes(float t0, float t1, float t2, float t3) {
  const uint32_t p01 = __builtin_amdgcn_perm(__float_as_uint(t1), __float_as_uint(t0), 0x0c0c0400u);
  const uint32_t p23 = __builtin_amdgcn_perm(__float_as_uint(t3), __float_as_uint(t2), 0x04000c0cu);
  return p01 | p23;
}
__device__ __forceinline__ float med3f(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
// Four non-negative-clamped values -> one dword of bytes: v_cvt_pk_u8_f32 rounds to nearest-even and saturates to
// [0, 255] (probed on gfx950: 0.5 -> 0, 1.5 -> 2, 2.5 -> 2, -0.6 -> 0, 300 -> 255), so  byte = rne(clamp(v, 0, hi))
// costs min + cvt per element instead of med3 + magic-add + the v_perm packing.  Only where the lower clamp bound is 0
// (ReLU-fused outputs stored centred on their zero point).
__device__ __forceinline__ uint32_t pack_rne_u8(float v0, float v1, float v2, float v3, float hi) {
  uint32_t r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fminf(v0, hi), 0u, 0u);
  r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fminf(v1, hi), 1u, r);
  r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fminf(v2, hi), 2u, r);
  return __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fminf(v3, hi), 3u, r);
}

// the 16 bytes a lane contributes to a B-operand (pixel) fragment: one ds_read_b128 where pixels are 16-byte aligned
template <class C>
__device__ __forceinline__ v4i load_xfrag(const uint8_t* p) {
  if constexpr (C::PIXB % 16 == 0) {
    return *reinterpret_cast<const v4i*>(p);
  } else {
    const v2i lo = *reinterpret_cast<const v2i*>(p);
    const v2i hi = *reinterpret_cast<const v2i*>(p + 8);
    return v4i{lo.x, lo.y, hi.x, hi.y};
  }
}

// zero the halo ring of G tiles of geometry (TW x TW x CH), 8-byte stores
template <int TW, int CH, int TILE_BYTES, int G, int NTHR = 256>
__device__ __forceinline__ void zero_halo(uint8_t* tile, int tid) {
  constexpr int PITCH = TW * CH;
  constexpr int ROW8 = PITCH / 8;                 // 8-byte words per full row
  constexpr int COL8 = CH / 8;
  constexpr int PER = 2 * ROW8 + 2 * (TW - 2) * COL8;
  const v2i z = {0, 0};
  for (int i = tid; i < G * PER; i += NTHR) {
    const int g = i / PER;
    int j = i - g * PER;
    int off;
    if (j < ROW8) off = j * 8;
    else if (j < 2 * ROW8) off = (TW - 1) * PITCH + (j - ROW8) * 8;
    else {
      j -= 2 * ROW8;
      const int row = 1 + j / (2 * COL8), q = j % (2 * COL8);
      off = row * PITCH + (q < COL8 ? q * 8 : (TW - 1) * CH + (q - COL8) * 8);
    }
    *reinterpret_cast<v2i*>(tile + g * TILE_BYTES + off) = z;
  }
}

// HBM quint8 NHWC images -> centred int8 halo'd tiles in LDS (16 B / lane loads, 2 x 8 B LDS stores)
template <class C, bool PRESUB>
__device__ __forceinline__ void load_tiles(uint8_t* tile, const uint8_t* xs, int img0, int B, int z_x, int tid) {
  constexpr int CPR = C::ROWB / 16, CPI = C::HIN * CPR;
  const uint32_t z4 = (uint32_t)z_x * 0x01010101u;
  for (int i = tid; i < C::G * CPI; i += 256) {
    const int g = i / CPI, rem = i - g * CPI;
    const int row = rem / CPR, within = rem - row * CPR;
    v4i v = {0, 0, 0, 0};
    if (img0 + g < B) {
      v = *reinterpret_cast<const v4i*>(xs + ((int64_t)(img0 + g) * C::HIN + row) * C::ROWB + within * 16);
      if (!PRESUB) { v.x = sub_bytes(v.x, z4); v.y = sub_bytes(v.y, z4); v.z = sub_bytes(v.z, z4); v.w = sub_bytes(v.w, z4); }
    }
    uint8_t* d = tile + g * C::TILE_BYTES + (row + C::HALO) * C::PITCH + C::row_chunk_off(within);
    *reinterpret_cast<v2i*>(d) = v2i{v.x, v.y};
    *reinterpret_cast<v2i*>(d + 8) = v2i{v.z, v.w};
  }
}

// bias -> LDS (zeros when the layer has none: fma(0, rcp, x) == x exactly)
template <int COUT, int NTHR = 256>
__device__ __forceinline__ void load_bias(float* dst, const float* bias, int tid) {
  for (int i = tid; i < COUT; i += NTHR) dst[i] = bias ? bias[i] : 0.0f;
}

// Row-reuse variant for 32-pixel-wide maps (layer 1): one M-tile = one output row.  A pass of MB consecutive output
// rows needs MB + 2 input rows; each input row's fragments are read from LDS once and feed the (up to) 3 output rows
// that touch it; all KS weight fragments stay in registers.  No load sits between two MFMAs.
template <class C, class Epi, int NWAVES>
__device__ __forceinline__ void conv_passes_rows(const uint8_t* tile, const int8_t* wq, const float* bias_lds, const QConv& p,
                                                 Epi& epi, int wave, int lane) {
  static_assert(C::ROWREUSE && C::USE_ONES, "row-reuse path: 32-wide, 3x3/s1, single ragged n-tile");
  const int r = lane & 31, h = lane >> 5;
  constexpr int NR = C::MB + C::KSZ - 1;
  for (int pass = wave; pass < C::NPASS; pass += NWAVES) {
    const int m0 = pass * C::MB * 32;                         // first pixel of the pass (NBLKS == 1)
    const int g = m0 / (C::HO * C::HO), oh0 = (m0 % (C::HO * C::HO)) / C::HO;
    const uint8_t* base = tile + g * C::TILE_BYTES + ((oh0 + C::OFF0) * C::TW + r + C::OFF0) * C::PIXB + 16 * h;
    QBNN_INNER_T0();
    v4i w[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) w[ks] = *reinterpret_cast<const v4i*>(wq + ((int64_t)ks * 64 + lane) * 16);
    float4 b4[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      if (8 * g4 < C::COUT) b4[g4] = *reinterpret_cast<const float4*>(bias_lds + 8 * g4 + 4 * h);
    v4i x[NR][C::SPR];
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int t = 0; t < C::SPR; ++t) {
        const v2i lo = *reinterpret_cast<const v2i*>(base + j * C::PITCH + t * 32);
        const v2i hi = *reinterpret_cast<const v2i*>(base + j * C::PITCH + t * 32 + 8);
        x[j][t] = v4i{lo.x, lo.y, hi.x, hi.y};
      }
    // Software pipeline over the MB output rows of the pass: the 9 MFMAs of row mb are issued, then the epilogue
    // (VALU + LDS) of row mb-1 -- independent instruction streams inside one basic block, so the matrix pipe works
    // on row mb while the vector pipe requantises row mb-1.
    v16i acc[C::MB];
    const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    QBNN_INNER_AT(0);
#pragma unroll
    for (int mb = 0; mb <= C::MB; ++mb) {
      if (mb < C::MB) {
#pragma unroll
        for (int kh = 0; kh < C::KSZ; ++kh)
#pragma unroll
          for (int t = 0; t < C::SPR; ++t)
            acc[mb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[kh * C::SPR + t], x[mb + kh][t], (kh == 0 && t == 0) ? zero16 : acc[mb], 0, 0, 0);
        QBNN_INNER_AT(1);
      }
      if (mb > 0) {
        const int e = mb - 1;
        const int rv = acc[e][C::ONES_REG];
        const int ro = __shfl_xor(rv, 32);
        const int zwr = p.z_w * (h ? ro : rv);
        const int po = epi.pixel(m0 + e * 32 + r);
        uint32_t pre[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          if (8 * g4 < C::COUT) pre[g4] = epi.load(po, 8 * g4 + 4 * h);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          if (8 * g4 >= C::COUT) continue;
          const float4 bb = b4[g4];
          const float v0 = __builtin_fmaf(bb.x, p.rcp, (float)(acc[e][4 * g4 + 0] - zwr)) * p.mult;
          const float v1 = __builtin_fmaf(bb.y, p.rcp, (float)(acc[e][4 * g4 + 1] - zwr)) * p.mult;
          const float v2 = __builtin_fmaf(bb.z, p.rcp, (float)(acc[e][4 * g4 + 2] - zwr)) * p.mult;
          const float v3 = __builtin_fmaf(bb.w, p.rcp, (float)(acc[e][4 * g4 + 3] - zwr)) * p.mult;
          epi.store(po, 8 * g4 + 4 * h, v0, v1, v2, v3, pre[g4]);
        }
        QBNN_INNER_AT(2);
      }
    }
    QBNN_INNER_FLUSH();
  }
}

template <class C, class Epi, int NWAVES>
__device__ __forceinline__ void conv_passes_stream(const uint8_t* tile, const int8_t* wq, const float* bias_lds, const QConv& p,
                                                   Epi& epi, int wave, int lane);      // defined next to conv_epi_phase

// All MFMA passes of one conv over LDS-resident tiles.
// Epilogue functor interface:  pre = epi.load(m, c0)   (residual dword or 0; issued ahead of the arithmetic)
//                              epi.store(m, c0, v0..v3, pre)
// with, for tile pixel m and the four consecutive output channels c0..c0+3,
//   v = fma(bias, rcp, float(acc - z_w R)) * mult  (un-clamped, un-rounded).
// Software pipeline: the weight and activation fragments of K-chunk kc+1 are in flight (L2 -> VGPR, LDS -> VGPR) while
// the MFMAs of chunk kc issue; inside a chunk every offset is an immediate.
template <class C, class Epi, int NWAVES = 4>
__device__ __forceinline__ void conv_passes(const uint8_t* tile, const int8_t* wq, const float* bias_lds, const QConv& p,
                                            Epi& epi, int wave, int lane) {
  if constexpr (C::ROWREUSE) {
    conv_passes_rows<C, Epi, NWAVES>(tile, wq, bias_lds, p, epi, wave, lane);
    return;
  } else if constexpr (C::WDEPTH > 0) {
    conv_passes_stream<C, Epi, NWAVES>(tile, wq, bias_lds, p, epi, wave, lane);
    return;
  }
  const int r = lane & 31, h = lane >> 5;
  constexpr int U = C::KCHUNK, NCHUNK = C::KS / U;
  struct Frags { v4i w[U][C::NB]; v4i x[U][C::MB]; };
  for (int pass = wave; pass < C::NPASS; pass += NWAVES) {
    const int mblk = pass / C::NBLKS, nblk = pass - mblk * C::NBLKS;
    const uint8_t* ap[C::MB];
#pragma unroll
    for (int mb = 0; mb < C::MB; ++mb) {
      const int m = (mblk * C::MB + mb) * 32 + r;
      const int g = m / (C::HO * C::HO), rem = m % (C::HO * C::HO);
      const int oh = rem / C::HO, ow = rem % C::HO;
      ap[mb] = tile + g * C::TILE_BYTES + ((oh * C::STRIDE + C::OFF0) * C::TW + ow * C::STRIDE + C::OFF0) * C::PIXB + 16 * h;
    }
    const int8_t* wbase = wq + ((int64_t)(nblk * C::NB) * C::KS * 64 + lane) * 16;
    auto load_chunk = [&](Frags& f, int kc) {
      const int ks0 = kc * U;
      const int kh = ks0 / C::SPR, t0 = ks0 - kh * C::SPR;
      int aoff;
      if constexpr (C::PADB == 0) aoff = kh * C::PITCH + t0 * 32;
      else { const int kw = t0 / C::SPT; aoff = kh * C::PITCH + kw * C::PIXB + (t0 - kw * C::SPT) * 32; }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int nb = 0; nb < C::NB; ++nb)
          f.w[u][nb] = *reinterpret_cast<const v4i*>(wbase + ((int64_t)(nb * C::KS + ks0 + u) * 64) * 16);
#pragma unroll
        for (int mb = 0; mb < C::MB; ++mb) {
          f.x[u][mb] = load_xfrag<C>(ap[mb] + aoff + u * 32);
        }
      }
    };
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
    auto mfma_chunk = [&](const Frags& f, int kc) {
      const int t0 = (kc * U) % C::SPR;
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int mb = 0; mb < C::MB; ++mb) {
          if (!C::USE_ONES) {
            // window sum on the fragments the lane holds; pad bytes of a ragged kernel row are unrelated data
            const bool v0ok = (C::RB % 32 == 0) || ((t0 + u) * 32 + 16 * h + 0 < C::RB);
            const bool v1ok = (C::RB % 32 == 0) || ((t0 + u) * 32 + 16 * h + 8 < C::RB);
            const int m0 = v0ok ? 0x01010101 : 0, m1 = v1ok ? 0x01010101 : 0;
            int rs_ = rsum[mb];
            rs_ = __builtin_amdgcn_sdot4(f.x[u][mb].x, m0, rs_, false);
            rs_ = __builtin_amdgcn_sdot4(f.x[u][mb].y, m0, rs_, false);
            rs_ = __builtin_amdgcn_sdot4(f.x[u][mb].z, m1, rs_, false);
            rs_ = __builtin_amdgcn_sdot4(f.x[u][mb].w, m1, rs_, false);
            rsum[mb] = rs_;
          }
#pragma unroll
          for (int nb = 0; nb < C::NB; ++nb) {
            acc[mb][nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.w[u][nb], f.x[u][mb], acc[mb][nb], 0, 0, 0);
          }
        }
      }
    };
    Frags f0, f1;
    load_chunk(f0, 0);
    if (NCHUNK <= 4) {
      // short K (fully unrolled)
#pragma unroll
      for (int kc = 0; kc < NCHUNK; ++kc) {
        Frags& cur = (kc & 1) ? f1 : f0;
        Frags& nxt = (kc & 1) ? f0 : f1;
        if (kc + 1 < NCHUNK) load_chunk(nxt, kc + 1);
        mfma_chunk(cur, kc);
      }
    } else {
      int kc = 0;
#pragma unroll 1
      while (true) {
        if (kc + 1 < NCHUNK) load_chunk(f1, kc + 1);
        mfma_chunk(f0, kc);
        if (++kc >= NCHUNK) break;
        if (kc + 1 < NCHUNK) load_chunk(f0, kc + 1);
        mfma_chunk(f1, kc);
        if (++kc >= NCHUNK) break;
      }
    }
    // bias of this pass's channels -> registers only now: held across the K loop they push the fused kernels into
    // scratch, and a spill reload behind the input prefetch costs a full vmcnt(0) drain
    float4 b4[C::NB][4];
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        if (C::COUT % 32 != 0 && nb * 32 + 8 * g4 >= C::COUT) continue;
        b4[nb][g4] = *reinterpret_cast<const float4*>(bias_lds + (nblk * C::NB + nb) * 32 + 8 * g4 + 4 * h);
      }
#pragma unroll
    for (int mb = 0; mb < C::MB; ++mb) {
      int R;
      if (C::USE_ONES) {                 // output row COUT of the ones tile holds R; it lives in the h == 0 lanes
        const int rv = acc[mb][C::ONES_TILE % C::NB][C::ONES_REG];
        const int ro = __shfl_xor(rv, 32);
        R = h ? ro : rv;
      } else {
        R = rsum[mb] + __shfl_xor(rsum[mb], 32);
      }
      const int zwr = p.z_w * R;
      const int m = epi.pixel((mblk * C::MB + mb) * 32 + r);
#pragma unroll
      for (int nb = 0; nb < C::NB; ++nb) {
        uint32_t pre[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          // ragged COUT (24, 48): NBLKS == 1, so nb is the tile index and this test folds at compile time
          if (C::COUT % 32 != 0 && nb * 32 + 8 * g4 >= C::COUT) continue;
          pre[g4] = epi.load(m, (nblk * C::NB + nb) * 32 + 8 * g4 + 4 * h);
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          if (C::COUT % 32 != 0 && nb * 32 + 8 * g4 >= C::COUT) continue;
          const int c0 = (nblk * C::NB + nb) * 32 + 8 * g4 + 4 * h;
          const float4 bb = b4[nb][g4];
          const float v0 = __builtin_fmaf(bb.x, p.rcp, (float)(acc[mb][nb][4 * g4 + 0] - zwr)) * p.mult;
          const float v1 = __builtin_fmaf(bb.y, p.rcp, (float)(acc[mb][nb][4 * g4 + 1] - zwr)) * p.mult;
          const float v2 = __builtin_fmaf(bb.z, p.rcp, (float)(acc[mb][nb][4 * g4 + 2] - zwr)) * p.mult;
          const float v3 = __builtin_fmaf(bb.w, p.rcp, (float)(acc[mb][nb][4 * g4 + 3] - zwr)) * p.mult;
          epi.store(m, c0, v0, v1, v2, v3, pre[g4]);
        }
      }
    }
  }
}

// ---- all stochastic layers of a model in ONE launch -------------------------------------------------------------
#define QBNN_MAX_SAMPLER_LAYERS 24
struct SamplerLayer {
  const v4i* mu; const v4i* sigma; int8_t* out; int64_t out_ss;
  int cout, K, krow, rbp, KS, layout, n_chunks, chunk_begin;     // chunk_begin: first 256-thread block of this layer
  uint32_t layer_id;
  qbnn_sample_params p;
};
struct SamplerTable { SamplerLayer l[QBNN_MAX_SAMPLER_LAYERS]; int n; };

#define QBNN_SAMPLER_NS 4           // MC samples per thread: the chunk's mu / sigma are loaded, unpacked and dequantised once for all of them
__global__ __launch_bounds__(256) void sample_weights_multi_kernel(const SamplerTable t, uint32_t seed_lo, uint32_t seed_hi,
                                                                   uint32_t sample_begin, int n_samples, const uint32_t* __restrict__ nd) {
  if (nd) { seed_lo = nd[0]; seed_hi = nd[1]; sample_begin = nd[2]; }      // captured-graph mode: the seed lives in device memory
  int li = 0;
#pragma unroll 1
  for (int i = 1; i < t.n; ++i) li = ((int)blockIdx.x >= t.l[i].chunk_begin) ? i : li;
  const SamplerLayer& L = t.l[li];
  __shared__ uint32_t eps_tab[256];
  load_eps_table(eps_tab, threadIdx.x);
  __syncthreads();
  const int chunk = ((int)blockIdx.x - L.chunk_begin) * 256 + threadIdx.x;
  if (chunk >= L.n_chunks) return;
  const int s0 = blockIdx.y * QBNN_SAMPLER_NS;
  int n = 0, kh = 0, j0 = 0;
  if (L.layout == QBNN_LAYOUT_MFMA32) {
    const int lane = chunk & 63, tile = chunk >> 6;
    const int nt = tile / L.KS, ks = tile - nt * L.KS;
    n = nt * 32 + (lane & 31);
    const int kp0 = ks * 32 + (lane >> 5) * 16;
    kh = kp0 / L.rbp; j0 = kp0 - kh * L.rbp;
  }
  const bool ones_row = (L.layout == QBNN_LAYOUT_MFMA32) && (L.cout & 31) && n == L.cout;
  const v4i m4 = L.mu[chunk], s4 = L.sigma[chunk];
  int mw[4] = {m4.x, m4.y, m4.z, m4.w}, sw[4] = {s4.x, s4.y, s4.z, s4.w};
  const int64_t total = (int64_t)L.cout * L.K;
  if (L.layout == QBNN_LAYOUT_MFMA32 && ((L.K | L.krow) & 3) == 0) {
    // Fast path (every conv but layers.0): the chunk's 16 weights are 4 whole Philox blocks -- element index
    // n K + kh krow + j0 + j with all terms multiples of 4 -- and a block is valid or padding as a whole.
    // The quantized::mul / add chain of sample_one_q in fp32 on exact small integers (same bits, a third of the instructions):
    //   prod = (sigma_q - z_sigma) eps_q                       exact product of two small integers
    //   t'   = rne(clamp(prod * m, -128 - z_mul, 127 - z_mul))  = t_q - z_mul  (rne and an integer-bounded clamp commute)
    //   dt   = fma(s_mul, t', dl_mul),  dl_mul = s_mul z_mul + nzs_mul exactly (QConv::dl_y's argument)
    //   w'   = clamp((dw + dt) / s_add, lo - z_add, hi - z_add);  byte = low byte of ((w' + 1.5 * 2^23) + z_add)
    // dw = fma(s_w, mu_q, nzs_w) and sigma_q - z_sigma do not depend on the sample: computed once per QBNN_SAMPLER_NS samples.
    const qbnn_sample_params& P = L.p;
    const float tlo = (float)(-128 - P.z_mul), thi = (float)(127 - P.z_mul);
    const float dl_mul = __builtin_fmaf(P.s_mul, (float)P.z_mul, P.nzs_mul);
    const float wlo = (float)(max(-128, P.w_lo) - P.z_add), whi = (float)(min(127, P.w_hi) - P.z_add), zaf = (float)P.z_add;
    const float zsf = (float)P.z_sigma;
    float dw[16], sg[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      dw[j] = __builtin_fmaf(P.s_w, (float)((mw[j >> 2] << (24 - 8 * (j & 3))) >> 24), P.nzs_w);
      sg[j] = (float)((sw[j >> 2] << (24 - 8 * (j & 3))) >> 24) - zsf;
    }
    const int64_t idx0 = (int64_t)n * L.K + kh * L.krow + j0;
#pragma unroll 1
    for (int ss = 0; ss < QBNN_SAMPLER_NS; ++ss) {
      const int s = s0 + ss;
      if (s >= n_samples) break;
      uint32_t ow[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (ones_row && j0 + 4 * g < L.krow) ow[g] = 0x01010101u;
        if (n < L.cout && j0 + 4 * g < L.krow) {
          const qbnn::u32x4 r4 = qbnn::philox4x32_10((uint32_t)((idx0 >> 2) + g), L.layer_id, sample_begin + s, 0u, seed_lo, seed_hi);
          const uint32_t uu[4] = {r4.x, r4.y, r4.z, r4.w};
          float f[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float ef = (float)eps_q_from_u32(uu[i], eps_tab);
            const float tq = __builtin_rintf(med3f((sg[4 * g + i] * ef) * P.mul_multiplier, tlo, thi));
            const float dt = __builtin_fmaf(P.s_mul, tq, dl_mul);
            f[i] = (med3f((dw[4 * g + i] + dt) * P.inv_s_add, wlo, whi) + QBNN_MAGIC) + zaf;
          }
          ow[g] = pack_low_bytes(f[0], f[1], f[2], f[3]);
        }
      }
      reinterpret_cast<v4i*>(L.out + (int64_t)s * L.out_ss)[chunk] = v4i{(int)ow[0], (int)ow[1], (int)ow[2], (int)ow[3]};
    }
    return;
  }
#pragma unroll 1
  for (int ss = 0; ss < QBNN_SAMPLER_NS; ++ss) {
    const int s = s0 + ss;
    if (s >= n_samples) break;
    uint32_t ow[4] = {0u, 0u, 0u, 0u};
    uint32_t cur_blk = 0xffffffffu;
    qbnn::u32x4 rb = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      int64_t idx;
      bool valid;
      if (L.layout == QBNN_LAYOUT_MFMA32) {
        valid = (n < L.cout) && (j0 + j < L.krow);
        idx = (int64_t)n * L.K + kh * L.krow + j0 + j;
        if (ones_row && j0 + j < L.krow) ow[j >> 2] |= 1u << (8 * (j & 3));
      } else {
        idx = (int64_t)chunk * 16 + j;
        valid = idx < total;
      }
      if (valid) {
        const uint32_t blk = (uint32_t)(idx >> 2);
        if (blk != cur_blk) {
          cur_blk = blk;
          rb = qbnn::philox4x32_10(blk, L.layer_id, sample_begin + s, 0u, seed_lo, seed_hi);
        }
        const int l = (int)(idx & 3);
        const uint32_t u = l == 0 ? rb.x : (l == 1 ? rb.y : (l == 2 ? rb.z : rb.w));
        const int mu_q = (mw[j >> 2] << (24 - 8 * (j & 3))) >> 24;
        const int sg_q = (sw[j >> 2] << (24 - 8 * (j & 3))) >> 24;
        ow[j >> 2] |= ((uint32_t)sample_one_q(mu_q, sg_q, eps_q_from_u32(u, eps_tab), L.p) & 0xffu) << (8 * (j & 3));
      }
    }
    reinterpret_cast<v4i*>(L.out + (int64_t)s * L.out_ss)[chunk] = v4i{(int)ow[0], (int)ow[1], (int)ow[2], (int)ow[3]};
  }
}

QBNN_EXPORT int qbnn_sample_weights_i8_multi(const qbnn_sampler_layer* layers, int32_t n_layers, uint64_t seed,
                                             uint32_t sample_begin, int32_t n_samples, void* stream) {
  if (!layers || n_layers <= 0 || n_layers > QBNN_MAX_SAMPLER_LAYERS || n_samples <= 0)
    return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8_multi: bad argument (at most 24 layers per call)%s");
  SamplerTable t;
  memset(&t, 0, sizeof(t));
  t.n = n_layers;
  int blocks = 0;
  for (int i = 0; i < n_layers; ++i) {
    const qbnn_sampler_layer& q = layers[i];
    if (!q.mu_packed || !q.sigma_packed || !q.w_out || q.cout <= 0 || q.k <= 0)
      return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8_multi: bad layer entry%s");
    if (q.layout == QBNN_LAYOUT_MFMA32 && (q.krow <= 0 || q.k % q.krow))
      return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8_multi: k must be a multiple of krow%s");
    const size_t bytes = qbnn_packed_weight_bytes(q.cout, q.k, q.krow, q.layout);
    if ((size_t)q.w_sample_stride < bytes || (q.w_sample_stride & 15))
      return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8_multi: w_sample_stride too small or unaligned%s");
    const PackGeom g = pack_geom(q.cout, q.k, q.layout == QBNN_LAYOUT_MFMA32 ? q.krow : q.k);
    SamplerLayer& L = t.l[i];
    L.mu = (const v4i*)q.mu_packed; L.sigma = (const v4i*)q.sigma_packed; L.out = q.w_out; L.out_ss = q.w_sample_stride;
    L.cout = q.cout; L.K = q.k; L.krow = g.krow; L.rbp = g.rbp; L.KS = g.KS; L.layout = q.layout;
    L.n_chunks = (int)(bytes / 16); L.chunk_begin = blocks; L.layer_id = q.layer_id; L.p = q.params;
    blocks += ceil_div(L.n_chunks, 256);
  }
  hipLaunchKernelGGL(sample_weights_multi_kernel, dim3(blocks, ceil_div(n_samples, QBNN_SAMPLER_NS)), dim3(256), 0, (hipStream_t)stream, t,
                     (uint32_t)seed, (uint32_t)(seed >> 32), sample_begin, n_samples, g_noise_dev);
  return check_launch("qbnn_sample_weights_i8_multi");
}

// =====================================================================================
// LDS-DMA helpers of the fused kernels: weights reach a CU ONCE per workgroup -- its waves DMA them (global_load_lds, 1 KiB
// fragment tile per wave-instruction, no VGPRs) into LDS; every wave then reads its fragments with ds_read_b128.
// =====================================================================================
// Barrier that publishes LDS-DMA data.  __syncthreads() alone is NOT enough: at workgroup scope the compiler's release
// fence waits for LDS traffic only (lgkmcnt), and global_load_lds completes on the vector-memory counter -- without
// the explicit vmcnt(0) a wave could pass the barrier while its own share of the slab is still in flight.
// s_waitcnt simm16 on gfx9: vmcnt = {[15:14],[3:0]}, expcnt = [6:4], lgkmcnt = [11:8]; 0x0f70 = vmcnt(0) only.
__device__ __forceinline__ void dma_barrier() {
  __builtin_amdgcn_s_waitcnt(0x0f70);
  __syncthreads();
}

template <class C, int NWAVES>
__device__ __forceinline__ void dma_slab(uint8_t* dst, const int8_t* wq, int slab, int wave, int lane) {
  constexpr int NFRAG = C::NT * C::SLK;
  for (int f = wave; f < NFRAG; f += NWAVES) {
    const int nt = f / C::SLK, u = f - nt * C::SLK;
    __builtin_amdgcn_global_load_lds(wq + ((int64_t)(nt * C::KS + slab * C::SLK + u) * 64 + lane) * 16,
                                     (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
  }
}

// Post-ops of the layer kernel for graphs with a dropout behind every conv (mcdropout/models_mc.py:116-160): quantised
// BernoulliDropout on the conv output, then optionally quantized::add with the block's other branch + ReLU -- in the conv's
// epilogue (EpiDenseDrop), on the centred integer it already holds.  Same bits as the stand-alone kernels
// (dropout_q_kernel, add_relu_q_kernel), whose element functions follow.
struct PostArgs {
  float keep, inv_sm, dmult; int z_m;
  float dlo, dhi;                  // clamp of the dropped value before rounding: -z_m, min(255, a_hi) - z_m
  uint32_t seed_lo, seed_hi, layer_id, sample_begin;
  const float* mask_in; const uint32_t* nd;
  float s_a, dl_a;                 // add: first operand = the dropped conv output (s_m / (1 - p), z_m); dl_a = s_a z_m + nzs_a exactly
};

// quantised mask value minus its zero point for slot i = b * C + c of MC sample s (mcdropout/dropout.py:24-33)
__device__ __forceinline__ int drop_mask_q(int i, int s, int64_t n_slots, float keep, float inv_sm, int z_m, uint32_t seed_lo,
                                           uint32_t seed_hi, uint32_t layer_id, uint32_t sample_begin, const float* mask_in) {
  float m;
  if (mask_in) {
    m = mask_in[(int64_t)s * n_slots + i];
  } else {
    const qbnn::u32x4 r = qbnn::philox4x32_10((uint32_t)(i >> 2), layer_id, sample_begin + s, 1u, seed_lo, seed_hi);
    const uint32_t rv = (i & 3) == 0 ? r.x : ((i & 3) == 1 ? r.y : ((i & 3) == 2 ? r.z : r.w));
    m = ((float)(rv >> 8) * 5.9604644775390625e-8f) < keep ? 1.0f : 0.0f;
  }
  return min(max(z_m + rne_sat(m * inv_sm), 0), 255) - z_m;
}
// quantized::mul(x, mask_q) with the mask's qparams as output qparams, then clamp_activation
__device__ __forceinline__ uint32_t drop_one(int xb, int mq, int z_x, int z_m, float mult, int hi) {
  const int q = min(max(z_m + rne_sat((float)((xb - z_x) * mq) * mult), 0), 255);
  return (uint32_t)min(q, hi);
}
// quantized::add + clamp_activation (+ ReLU)
__device__ __forceinline__ uint32_t add_relu_one(uint32_t qa, uint32_t qb, float s_a, float nzs_a, float s_b, float nzs_b, float inv_s_o,
                                                 int z_o, int a_hi, int relu) {
  const float da = __builtin_fmaf(s_a, (float)qa, nzs_a);
  const float db = __builtin_fmaf(s_b, (float)qb, nzs_b);
  int q = min(max(z_o + rne_sat((da + db) * inv_s_o), 0), 255);
  q = min(q, a_hi);
  if (relu) q = max(q, z_o);
  return (uint32_t)q;
}

// ---- epilogue functors -----------------------------------------------------------------------------------------
// (a) quint8 into a dense [M][COUT] staging buffer; optional quantized::add + ReLU against the quint8 residual that
//     already sits at the same address (updated in place).
template <int COUT, bool HAS_RES, int PITCH = COUT>
struct EpiDense {
  static constexpr int VALU_PER_MFMA = HAS_RES ? 22 : 11;   // interleave hint: epilogue VALU ops of one 32-pixel row / 9 MFMAs
  uint8_t* outb; QConv p; QAdd a;
  __device__ __forceinline__ int pixel(int m) const { return m * PITCH; }
  __device__ __forceinline__ uint32_t load(int po, int c0) const {
    return HAS_RES ? *reinterpret_cast<const uint32_t*>(outb + po + c0) : 0u;
  }
  __device__ __forceinline__ void store(int po, int c0, float v0, float v1, float v2, float v3, uint32_t rq) const {
    uint32_t* o = reinterpret_cast<uint32_t*>(outb + po + c0);
    if (!HAS_RES && p.vlo == 0.0f) {      // ReLU-fused conv (workgroup-uniform): non-negative centred bytes, then + z_y bytewise
      *o = pack_rne_u8(v0, v1, v2, v3, p.vhi) + (uint32_t)p.z_y * 0x01010101u;
      return;
    }
    v0 = med3f(v0, p.vlo, p.vhi); v1 = med3f(v1, p.vlo, p.vhi); v2 = med3f(v2, p.vlo, p.vhi); v3 = med3f(v3, p.vlo, p.vhi);
    if (!HAS_RES) {
      const float zy = (float)p.z_y;
      // round with the (even) magic constant first, then add z_y exactly: folding an odd z_y into the constant
      // would flip round-half-even ties
      *o = pack_low_bytes((v0 + QBNN_MAGIC) + zy, (v1 + QBNN_MAGIC) + zy, (v2 + QBNN_MAGIC) + zy, (v3 + QBNN_MAGIC) + zy);
    } else {
      float t[4];
      const float vv[4] = {v0, v1, v2, v3};
      const float rf[4] = {(float)(rq & 0xffu), (float)((rq >> 8) & 0xffu), (float)((rq >> 16) & 0xffu), (float)(rq >> 24)};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float da = __builtin_fmaf(p.s_y, __builtin_rintf(vv[i]), p.dl_y);      // centred conv output integer, exactly
        const float db = __builtin_fmaf(a.s_r, rf[i], a.nzs_r);
        t[i] = (da + db) * a.inv_s_o;
      }
      *o = pack_rne_u8(t[0], t[1], t[2], t[3], a.vhi) + (uint32_t)a.z_o * 0x01010101u;      // bytes <= a_hi - z_o: no carry
    }
  }
};

// (a') EpiDense with a quantised channel dropout between the requantisation and the store / the Add:
//   q' = rne(clamp(v))                       centred conv output (q - z_y), as EpiDense
//   r' = rne(clamp((q' * mq) * dmult))       quantized::mul(x, mask_q): (x - z_x)(mask_q - z_m) is exact in fp32; centred on z_m
//   no Add: byte r' + z_m;   Add: fma(s_a, r', dl_a) dequantises it (QConv::dl_y's argument), the rest is EpiDense's Add + ReLU.
// mq: fp32 [G][COUT] in LDS, the mask value (minus its zero point) of (image, channel) for this MC sample.
template <int COUT, int IMG_PIX, bool HAS_RES>
struct EpiDenseDrop {
  static constexpr int VALU_PER_MFMA = HAS_RES ? 26 : 16;
  uint8_t* outb; QConv p; QAdd a; PostArgs q; const float* mq;
  __device__ __forceinline__ int pixel(int m) const { return m * COUT; }
  __device__ __forceinline__ uint32_t load(int po, int c0) const {
    return HAS_RES ? *reinterpret_cast<const uint32_t*>(outb + po + c0) : 0u;
  }
  __device__ __forceinline__ void store(int po, int c0, float v0, float v1, float v2, float v3, uint32_t rq) const {
    uint32_t* o = reinterpret_cast<uint32_t*>(outb + po + c0);
    const int g = po / (IMG_PIX * COUT);
    const float4 m4 = *reinterpret_cast<const float4*>(mq + g * COUT + c0);
    const float mm[4] = {m4.x, m4.y, m4.z, m4.w};
    const float vv[4] = {v0, v1, v2, v3};
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float qc = __builtin_rintf(med3f(vv[i], p.vlo, p.vhi));
      r[i] = med3f((qc * mm[i]) * q.dmult, q.dlo, q.dhi);
    }
    if (!HAS_RES) {
      const float zm = (float)q.z_m;
      *o = pack_low_bytes((r[0] + QBNN_MAGIC) + zm, (r[1] + QBNN_MAGIC) + zm, (r[2] + QBNN_MAGIC) + zm, (r[3] + QBNN_MAGIC) + zm);
    } else {
      const float rf[4] = {(float)(rq & 0xffu), (float)((rq >> 8) & 0xffu), (float)((rq >> 16) & 0xffu), (float)(rq >> 24)};
      float t[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float da = __builtin_fmaf(q.s_a, __builtin_rintf(r[i]), q.dl_a);
        const float db = __builtin_fmaf(a.s_r, rf[i], a.nzs_r);
        t[i] = (da + db) * a.inv_s_o;
      }
      *o = pack_rne_u8(t[0], t[1], t[2], t[3], a.vhi) + (uint32_t)a.z_o * 0x01010101u;
    }
  }
};

template <int HO, int PIXB, int TILE_BYTES>
__device__ __forceinline__ int tile_px_off(int m, int c0) {
  const int g = m / (HO * HO), rem = m % (HO * HO), oh = rem / HO, ow = rem % HO;
  return g * TILE_BYTES + ((oh + 1) * (HO + 2) + ow + 1) * PIXB + c0;
}

// (b) centred int8 (q - z_y) into the halo'd tile feeding the next conv (geometry HO x HO, PIXB bytes per pixel, halo 1)
template <int HO, int PIXB, int TILE_BYTES>
struct EpiTile {
  static constexpr int VALU_PER_MFMA = 10;
  uint8_t* dst; QConv p;
  __device__ __forceinline__ int pixel(int m) const { return tile_px_off<HO, PIXB, TILE_BYTES>(m, 0); }
  __device__ __forceinline__ uint32_t load(int, int) const { return 0u; }
  __device__ __forceinline__ void store(int po, int c0, float v0, float v1, float v2, float v3, uint32_t) const {
    uint32_t* o = reinterpret_cast<uint32_t*>(dst + po + c0);
    *o = pack_rne_u8(v0, v1, v2, v3, p.vhi);        // stem.0 is ConvReLU2d: p.vlo == 0 (set by the library, fill_qconv relu = 1)
  }
};

// (c) conv -> Add(residual) -> ReLU, residual read as centred int8 (x' = q_r - z_r) from a halo'd tile of the same
//     geometry and overwritten in place with the centred block output (q_o - z_o).
template <int HO, int PIXB, int TILE_BYTES>
struct EpiTileResInPlace {
  static constexpr int VALU_PER_MFMA = 22;
  uint8_t* xt; QConv p; QAdd a;
  __device__ __forceinline__ int pixel(int m) const { return tile_px_off<HO, PIXB, TILE_BYTES>(m, 0); }
  __device__ __forceinline__ uint32_t load(int po, int c0) const {
    return *reinterpret_cast<const uint32_t*>(xt + po + c0);
  }
  __device__ __forceinline__ void store(int po, int c0, float v0, float v1, float v2, float v3, uint32_t rqu) const {
    uint32_t* o = reinterpret_cast<uint32_t*>(xt + po + c0);
    const int rq = (int)rqu;
    const float vv[4] = {v0, v1, v2, v3};
    const float rf[4] = {(float)((rq << 24) >> 24), (float)((rq << 16) >> 24), (float)((rq << 8) >> 24), (float)(rq >> 24)};   // centred r'
    float t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float da = __builtin_fmaf(p.s_y, __builtin_rintf(med3f(vv[i], p.vlo, p.vhi)), p.dl_y);
      const float db = __builtin_fmaf(a.s_r, rf[i], a.dl_r);
      t[i] = (da + db) * a.inv_s_o;
    }
    *o = pack_rne_u8(t[0], t[1], t[2], t[3], a.vhi);
  }
};

// ---- single-conv kernel (layer-level C ABI entry) ---------------------------------------------------------------
struct ConvArgs {
  const uint8_t* x; int64_t x_ss;
  const uint8_t* res; int64_t res_ss;
  uint8_t* y; int64_t y_ss;
  int B;
  QConv p; QAdd a;
  PostArgs post;                   // POST kernels only
};

template <class C, bool HAS_RES, bool PRESUB, bool POST = false>
__global__ __launch_bounds__(256) void conv_i8_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* tile = smem;
  uint8_t* outb = smem + C::G * C::TILE_BYTES + C::TILE_SLACK;
  float* bias_lds = reinterpret_cast<float*>(outb + C::OUT_BYTES);
  float* mq_lds = bias_lds + C::COUT;                                     // POST: [G][COUT] mask values of this workgroup's images
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar control flow and addresses
  const int s = blockIdx.y, img0 = blockIdx.x * C::G;

  if (C::HALO > 0) zero_halo<C::TW, C::PIXB, C::TILE_BYTES, C::G>(tile, tid);
  load_tiles<C, PRESUB>(tile, a.x + (int64_t)s * a.x_ss, img0, a.B, a.p.z_x, tid);
  load_bias<C::COUT>(bias_lds, a.p.bias, tid);
  constexpr int IMG_OUT = C::HO * C::HO * C::COUT;
  if constexpr (POST) {
    uint32_t seed_lo = a.post.seed_lo, seed_hi = a.post.seed_hi, sample_begin = a.post.sample_begin;
    if (a.post.nd) { seed_lo = a.post.nd[0]; seed_hi = a.post.nd[1]; sample_begin = a.post.nd[2]; }
    for (int i = tid; i < C::G * C::COUT; i += 256) {
      const int b = img0 + i / C::COUT;
      mq_lds[i] = b < a.B ? (float)drop_mask_q(b * C::COUT + i % C::COUT, s, (int64_t)a.B * C::COUT, a.post.keep, a.post.inv_sm,
                                               a.post.z_m, seed_lo, seed_hi, a.post.layer_id, sample_begin, a.post.mask_in) : 0.f;
    }
  }
  if (HAS_RES) {
    const uint8_t* rs = a.res + (int64_t)s * a.res_ss;
    for (int i = tid; i < C::M * C::COUT / 16; i += 256)
      if (img0 + (i * 16) / IMG_OUT < a.B)
        reinterpret_cast<v4i*>(outb)[i] = *reinterpret_cast<const v4i*>(rs + (int64_t)img0 * IMG_OUT + (int64_t)i * 16);
  }
  __syncthreads();
  if constexpr (POST) {
    EpiDenseDrop<C::COUT, C::HO * C::HO, HAS_RES> epi{outb, a.p, a.a, a.post, mq_lds};
    conv_passes<C>(tile, a.p.w + (int64_t)s * a.p.w_ss, bias_lds, a.p, epi, wave, lane);
  } else {
    EpiDense<C::COUT, HAS_RES> epi{outb, a.p, a.a};
    conv_passes<C>(tile, a.p.w + (int64_t)s * a.p.w_ss, bias_lds, a.p, epi, wave, lane);
  }
  __syncthreads();
  uint8_t* ys = a.y + (int64_t)s * a.y_ss;
  for (int i = tid; i < C::M * C::COUT / 16; i += 256)
    if (img0 + (i * 16) / IMG_OUT < a.B)
      *reinterpret_cast<v4i*>(ys + (int64_t)img0 * IMG_OUT + (int64_t)i * 16) = reinterpret_cast<const v4i*>(outb)[i];
}

template <class C, bool PRESUB>
static int launch_conv(const ConvArgs& a, int n_samples, bool has_res, hipStream_t st, bool post = false) {
  constexpr int LDS = C::G * C::TILE_BYTES + C::TILE_SLACK + C::OUT_BYTES + C::COUT * 4;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  dim3 grid(ceil_div(a.B, C::G), n_samples);
  if (post) {
    if constexpr (!PRESUB) {
      constexpr int LDSP = LDS + C::G * C::COUT * 4;
      static_assert(LDSP <= 160 * 1024, "LDS budget");
      if (has_res) {
        static std::atomic<uint64_t> attr_pr{0};
        if (int rc_attr = ensure_dyn_lds((const void*)conv_i8_kernel<C, true, false, true>, attr_pr, LDSP)) return rc_attr;
        hipLaunchKernelGGL((conv_i8_kernel<C, true, false, true>), grid, dim3(256), LDSP, st, a);
      } else {
        static std::atomic<uint64_t> attr_p{0};
        if (int rc_attr = ensure_dyn_lds((const void*)conv_i8_kernel<C, false, false, true>, attr_p, LDSP)) return rc_attr;
        hipLaunchKernelGGL((conv_i8_kernel<C, false, false, true>), grid, dim3(256), LDSP, st, a);
      }
      return check_launch("qbnn_conv2d_i8_post_mc");
    } else {
      return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_post_mc: not for the im2col layer%s");
    }
  }
  if (has_res) {
    static std::atomic<uint64_t> attr_r{0};
    if (int rc_attr = ensure_dyn_lds((const void*)conv_i8_kernel<C, true, PRESUB>, attr_r, LDS)) return rc_attr;
    hipLaunchKernelGGL((conv_i8_kernel<C, true, PRESUB>), grid, dim3(256), LDS, st, a);
  } else {
    static std::atomic<uint64_t> attr_n{0};
    if (int rc_attr = ensure_dyn_lds((const void*)conv_i8_kernel<C, false, PRESUB>, attr_n, LDS)) return rc_attr;
    hipLaunchKernelGGL((conv_i8_kernel<C, false, PRESUB>), grid, dim3(256), LDS, st, a);
  }
  return check_launch("qbnn_conv2d_i8_mc");
}

// =====================================================================================
// Fused BasicBlock kernels (models_bbb.py:170-183): persistent workgroups, activations never leave LDS between
// the block's convs.
//
//   identity chain  : NBLK x [ stem.0 (3x3, ReLU) -> stem.3 (3x3) -> Add(x) -> ReLU ]   on one X/T tile pair
//       X tile: centred block input, overwritten in place by the centred block output (residual read + write by
//       the same lane); T tile: centred stem.0 output.
//   Work item = (MC sample s, group of G images).  Each workgroup walks items blockIdx.x, +gridDim.x, ... ; items
//   of one sample are adjacent, so the sample's weight slab stays hot in every XCD's L2.  The next item's input is
//   fetched into registers while the current one computes (issue-early / write-late), so HBM latency is off the
//   critical path.
// =====================================================================================
#ifndef QBNN_BLK_THREADS
#define QBNN_BLK_THREADS 512
#endif
constexpr int BLK_THREADS = QBNN_BLK_THREADS, BLK_WAVES = QBNN_BLK_THREADS / 64;

struct BlockParams { QConv a, b; QAdd add; };       // stem.0, stem.3, add

template <int NBLK>
struct ChainArgs {
  const uint8_t* x; int64_t x_ss;     // block-chain input  [S][B][H][H][C] quint8
  uint8_t* y; int64_t y_ss;           // block-chain output [S][B][H][H][C] quint8
  int B, n_samples;
  int z_in;                           // zero point of x
  unsigned long long* dbg;            // diagnostic builds only
  BlockParams blk[NBLK];
  const int8_t* stem_x;               // fused layer-0 conv (STEM kernels): centred im2col patches [B][32*32][32], shared by the samples
  QConv stem;
};

// Several independent launches of one fused kernel in ONE grid: gridDim.y (head: gridDim.z) picks the argument block.  Used for
// ensemble members (reference sgld/models_sgld.py:277-288): every member has its own tensors, weights AND quantisation
// parameters, so it cannot ride the MC-sample dimension of a launch -- but member m's workgroups can sit next to member
// m+1's.  NM = 1 is the ordinary launch (same code, argument block 0).
#define QBNN_FUSED_CALLS 8            // argument blocks per launch (kernel arguments are limited to 4 KiB)
template <class A, int NM> struct ArgsArr { A m[NM]; };

//                        CIN COUT K  S  HIN HALO G  MB NB
using Cfg_c0      = ConvCfg<32, 24, 1, 1, 32, 0, 1, 4, 1>;    // layers.0 on the im2col tensor (K = 27 -> 32)
using Cfg_24_24   = ConvCfg<24, 24, 3, 1, 32, 1, 1, 4, 1>;    // layers.3.*
using Cfg_24_48s  = ConvCfg<24, 48, 3, 2, 32, 1, 1, 2, 2>;    // layers.4.0.stem.0
using Cfg_24_48p  = ConvCfg<24, 48, 1, 2, 32, 1, 1, 2, 2>;    // layers.4.0.shortcut.0
using Cfg_48_48   = ConvCfg<48, 48, 3, 1, 16, 1, 2, 2, 2>;    // layers.4.*
using Cfg_48_96s  = ConvCfg<48, 96, 3, 2, 16, 1, 2, 1, 3>;    // layers.5.0.stem.0
using Cfg_48_96p  = ConvCfg<48, 96, 1, 2, 16, 1, 2, 1, 3>;    // layers.5.0.shortcut.0
using Cfg_96_96   = ConvCfg<96, 96, 3, 1, 8, 1, 4, 1, 3>;     // layers.5.*
using Cfg_96_192s = ConvCfg<96, 192, 3, 2, 8, 1, 4, 1, 3>;    // layers.6.0.stem.0
using Cfg_96_192p = ConvCfg<96, 192, 1, 2, 8, 1, 4, 1, 3>;    // layers.6.0.shortcut.0
using Cfg_192_192 = ConvCfg<192, 192, 3, 1, 4, 1, 4, 1, 3>;   // layers.6.*

static int fill_qconv(QConv& p, const int8_t* w, int64_t w_ss, const float* bias, const qbnn_conv_desc* d) {
  if (d->z_x < 0 || d->z_x > 127 || d->a_hi > 127 || d->a_hi < 1 || d->z_y < 0 || d->z_y > 127)
    return fail(QBNN_E_INVALID, "qbnn conv: activations must be <= 7 bit with zero points in [0,127] (reference quant_utils.py:120)%s");
  p.w = w; p.w_ss = w_ss; p.bias = d->has_bias ? bias : nullptr;
  p.z_x = d->z_x; p.z_w = d->z_w; p.z_y = d->z_y;
  const float atw = d->s_x * d->s_w;     // qconv.cpp GetQuantizationParams: float * float
  p.rcp = 1.0f / atw;                    // FBGEMM act_times_w_rcp
  p.mult = atw / d->s_y;                 // output_multiplier_float
  const int lo = d->relu ? d->z_y : 0, hi = d->a_hi < 255 ? d->a_hi : 255;
  p.vlo = (float)(lo - d->z_y); p.vhi = (float)(hi - d->z_y);
  p.s_y = d->s_y; p.nzs_y = (float)(-d->z_y) * d->s_y;
  p.dl_y = fmaf(d->s_y, (float)d->z_y, p.nzs_y);
  return QBNN_OK;
}

static int fill_qadd(QAdd& a, const qbnn_conv_desc* d) {
  if (d->z_o < 0 || d->z_o > 127 || d->z_r < 0 || d->z_r > 127)
    return fail(QBNN_E_INVALID, "qbnn conv: add zero points must be in [0,127]%s");
  a.s_r = d->s_r; a.nzs_r = (float)(-d->z_r) * d->s_r; a.z_r = d->z_r;
  a.dl_r = fmaf(d->s_r, (float)d->z_r, a.nzs_r);
  a.inv_s_o = 1.0f / d->s_o; a.z_o = d->z_o;
  a.vhi = (float)((d->a_hi < 255 ? d->a_hi : 255) - d->z_o);
  return QBNN_OK;
}

static int dispatch_conv(const ConvArgs& a, const qbnn_conv_desc* d, int n_samples, bool hr, bool post, hipStream_t st);

QBNN_EXPORT int qbnn_conv2d_i8_mc(const uint8_t* x, int64_t x_ss, const int8_t* w_packed, int64_t w_ss, const float* bias,
                                  const uint8_t* res, int64_t res_ss, uint8_t* y, int64_t y_ss, int32_t n_samples,
                                  const qbnn_conv_desc* d, void* stream) {
  if (!x || !w_packed || !y || !d || n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_mc: bad argument%s");
  if (d->has_res && !res) return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_mc: has_res set but res is NULL%s");
  if (d->has_bias && !bias) return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_mc: has_bias set but bias is NULL%s");
  if (d->H != d->W) return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_mc: only square inputs are supported%s");
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.x_ss = x_ss; a.res = res; a.res_ss = res_ss; a.y = y; a.y_ss = y_ss; a.B = d->B;
  int rc = fill_qconv(a.p, w_packed, w_ss, bias, d);
  if (rc) return rc;
  if (d->has_res && (rc = fill_qadd(a.a, d))) return rc;
  return dispatch_conv(a, d, n_samples, d->has_res != 0, false, (hipStream_t)stream);
}

static int dispatch_conv(const ConvArgs& a, const qbnn_conv_desc* d, int n_samples, bool hr, bool post, hipStream_t st) {
#define QBNN_CASE(CFG, cin, cout, ks, sd, hin, im2c)                                                      \
  if (d->Cin == (cin) && d->Cout == (cout) && d->ksize == (ks) && d->stride == (sd) && d->H == (hin) &&   \
      d->pad == ((ks) - 1) / 2 && (d->x_is_centered_im2col != 0) == (im2c))                                \
    return launch_conv<CFG, im2c>(a, n_samples, hr, st, post);
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

QBNN_EXPORT int qbnn_conv2d_i8_post_mc(const uint8_t* x, int64_t x_ss, const int8_t* w_packed, int64_t w_ss, const float* bias,
                                       uint8_t* y, int64_t y_ss, int32_t n_samples, const qbnn_conv_desc* d,
                                       const qbnn_post_desc* q, const float* mask_in, const uint8_t* other, int64_t other_ss,
                                       uint64_t seed, uint32_t sample_begin, void* stream) {
  if (!x || !w_packed || !y || !d || !q || n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_post_mc: bad argument%s");
  if (d->has_res || d->x_is_centered_im2col) return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_post_mc: has_res / im2col are not combined with post-ops%s");
  if (d->has_bias && !bias) return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_post_mc: has_bias set but bias is NULL%s");
  if (d->H != d->W) return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_post_mc: only square inputs are supported%s");
  if (q->add && (!other || (other_ss & 15) || (reinterpret_cast<uintptr_t>(other) & 15)))
    return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_post_mc: add operand missing or not 16-byte aligned%s");
  if (q->z_m < 0 || q->z_m > 127 || !(q->s_m > 0.f)) return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_post_mc: mask zero point must be in [0,127]%s");
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.x_ss = x_ss; a.y = y; a.y_ss = y_ss; a.B = d->B;
  int rc = fill_qconv(a.p, w_packed, w_ss, bias, d);
  if (rc) return rc;
  PostArgs& o = a.post;
  const int hi = d->a_hi < 255 ? d->a_hi : 255;
  o.keep = q->keep_prob; o.inv_sm = 1.0f / q->s_m; o.z_m = q->z_m;
  o.dmult = (float)((double)d->s_y * (double)q->s_m / (double)q->s_m);     // ATen qmul: self_scale * other_scale / out_scale
  o.dlo = (float)(-q->z_m); o.dhi = (float)(hi - q->z_m);
  o.seed_lo = (uint32_t)seed; o.seed_hi = (uint32_t)(seed >> 32); o.layer_id = q->drop_layer_id; o.sample_begin = sample_begin;
  o.mask_in = mask_in; o.nd = g_noise_dev;
  if (q->add) {
    // the Add's operands: a = the dropped conv output with (s_a, z_m), b = `other` -- the epilogue's residual slot
    qbnn_conv_desc dd = *d;
    dd.s_r = q->s_b; dd.z_r = q->z_b; dd.s_o = q->s_o; dd.z_o = q->z_o;
    if ((rc = fill_qadd(a.a, &dd))) return rc;
    a.res = other; a.res_ss = other_ss;
    o.s_a = q->s_a;
    o.dl_a = fmaf(q->s_a, (float)q->z_m, (float)(-q->z_m) * q->s_a);
  }
  return dispatch_conv(a, d, n_samples, q->add != 0, true, (hipStream_t)stream);
}

// =====================================================================================
// Fused down-sampling BasicBlock (models_bbb.py:146-183 with stride 2): shortcut 1x1/s2 conv, stem.0 3x3/s2 ConvReLU,
// stem.3 3x3 conv, Add, ReLU in one persistent kernel.
//   X tile (Cin, HIN): centred block input      --conv_s-->  SC: dense quint8 [M][COUT] (the residual operand)
//                                               --conv_a-->  T tile (COUT, HO): centred stem.0 output
//   T --conv_b--> + SC --> SC in place (block output, quint8) --> HBM
// =====================================================================================
struct DownArgs {
  const uint8_t* x; int64_t x_ss;
  uint8_t* y; int64_t y_ss;
  int B, n_samples, z_in;
  QConv s, a, b; QAdd add;
};

template <class CA, class CS, class CB, bool LDSW> static int launch_block_down_ws(const DownArgs& a, hipStream_t st);
static bool no_pingpong() {
  static const bool v = [] { const char* e = getenv("QBNN_NO_PINGPONG"); return e && e[0] == '1'; }();
  return v;
}

//                           CIN COUT K  S  HIN HALO G  MB NB
using D24_a = ConvCfg<24, 48, 3, 2, 32, 1, 1, 1, 2>;
using D24_s = ConvCfg<24, 48, 1, 2, 32, 1, 1, 1, 2>;
using D24_b = ConvCfg<48, 48, 3, 1, 16, 1, 1, 1, 2>;
using D48_a = ConvCfg<48, 96, 3, 2, 16, 1, 4, 1, 3, false>;
using D48_s = ConvCfg<48, 96, 1, 2, 16, 1, 4, 1, 3, false>;
using D48_b = ConvCfg<96, 96, 3, 1, 8, 1, 4, 1, 3, false, 36, 8>;
using D96_a = ConvCfg<96, 192, 3, 2, 8, 1, 8, 1, 3, false>;
using D96_s = ConvCfg<96, 192, 1, 2, 8, 1, 8, 1, 3, false>;
using D96_b = ConvCfg<192, 192, 3, 1, 4, 1, 8, 1, 3, false>;

static int build_down_args(DownArgs& a, const uint8_t* x, int64_t x_ss, float s_x, int32_t z_x, int32_t B, int32_t a_hi, const qbnn_down_desc* d,
                           uint8_t* y, int64_t y_ss, int32_t n_samples) {
  memset(&a, 0, sizeof(a));
  a.x = x; a.x_ss = x_ss; a.y = y; a.y_ss = y_ss; a.B = B; a.n_samples = n_samples; a.z_in = z_x;
  qbnn_conv_desc c;
  memset(&c, 0, sizeof(c));
  c.a_hi = a_hi;
  int rc;
  c.s_x = s_x; c.z_x = z_x; c.s_w = d->s_ws; c.z_w = d->z_ws; c.s_y = d->s_s; c.z_y = d->z_s; c.relu = 0; c.has_bias = d->bias_s != nullptr;
  if ((rc = fill_qconv(a.s, d->w_s, d->w_s_sample_stride, d->bias_s, &c))) return rc;
  c.s_w = d->blk.s_wa; c.z_w = d->blk.z_wa; c.s_y = d->blk.s_a; c.z_y = d->blk.z_a; c.relu = 1; c.has_bias = d->blk.bias_a != nullptr;
  if ((rc = fill_qconv(a.a, d->blk.w_a, d->blk.w_a_sample_stride, d->blk.bias_a, &c))) return rc;
  c.s_x = d->blk.s_a; c.z_x = d->blk.z_a; c.s_w = d->blk.s_wb; c.z_w = d->blk.z_wb; c.s_y = d->blk.s_b; c.z_y = d->blk.z_b; c.relu = 0;
  c.has_bias = d->blk.bias_b != nullptr;
  if ((rc = fill_qconv(a.b, d->blk.w_b, d->blk.w_b_sample_stride, d->blk.bias_b, &c))) return rc;
  c.s_r = d->s_s; c.z_r = d->z_s; c.s_o = d->blk.s_o; c.z_o = d->blk.z_o;
  return fill_qadd(a.add, &c);
}

template <class CA, class CS, class CB, bool LDSW> static int launch_block_down_ws_multi(const DownArgs* arr, int n, hipStream_t st);

QBNN_EXPORT int qbnn_block_down_i8_multi(const qbnn_down_call* calls, int32_t n_calls, int32_t B, int32_t H, int32_t Cin, int32_t a_hi, void* stream) {
  if (!calls || n_calls <= 0 || B <= 0) return fail(QBNN_E_INVALID, "qbnn_block_down_i8_multi: bad argument%s");
  hipStream_t st = (hipStream_t)stream;
  for (int c0 = 0; c0 < n_calls;) {
    const int n = n_calls - c0 < QBNN_FUSED_CALLS ? n_calls - c0 : QBNN_FUSED_CALLS;
    DownArgs arr[QBNN_FUSED_CALLS];
    int rc;
    for (int i = 0; i < n; ++i) {
      const qbnn_down_call& k = calls[c0 + i];
      if (!k.x || !k.y || !k.desc || k.n_samples <= 0 || !k.desc->blk.w_a || !k.desc->blk.w_b || !k.desc->w_s)
        return fail(QBNN_E_INVALID, "qbnn_block_down_i8_multi: bad call entry%s");
      if ((rc = build_down_args(arr[i], k.x, k.x_sample_stride, k.s_x, k.z_x, B, a_hi, k.desc, k.y, k.y_sample_stride, k.n_samples))) return rc;
    }
    if (Cin == 24 && H == 32) rc = launch_block_down_ws_multi<D24_a, D24_s, D24_b, true>(arr, n, st);
    else if (Cin == 48 && H == 16) rc = launch_block_down_ws_multi<D48_a, D48_s, D48_b, false>(arr, n, st);
    else if (Cin == 96 && H == 8) rc = launch_block_down_ws_multi<D96_a, D96_s, D96_b, false>(arr, n, st);
    else return fail(QBNN_E_INVALID, "qbnn_block_down_i8_multi: unsupported geometry%s Cin=%ld H=%ld", "", Cin, H);
    if (rc) return rc;
    c0 += n;
  }
  return QBNN_OK;
}

QBNN_EXPORT int qbnn_block_down_i8_mc(const uint8_t* x, int64_t x_ss, float s_x, int32_t z_x, int32_t B, int32_t H, int32_t Cin,
                                      int32_t a_hi, const qbnn_down_desc* d, uint8_t* y, int64_t y_ss, int32_t n_samples,
                                      void* stream) {
  if (!x || !y || !d || n_samples <= 0 || B <= 0 || !d->blk.w_a || !d->blk.w_b || !d->w_s)
    return fail(QBNN_E_INVALID, "qbnn_block_down_i8_mc: bad argument%s");
  DownArgs a;
  if (int rc = build_down_args(a, x, x_ss, s_x, z_x, B, a_hi, d, y, y_ss, n_samples)) return rc;
  hipStream_t st = (hipStream_t)stream;
  // (a ping-pong variant of this block -- phases W / M_a / E_sa / M_b / E_b on two 4-wave groups -- measured 15 % SLOWER
  //  than the weights-stationary kernel: five barrier intervals per image, each as long as the slower group's phase)
  if (Cin == 24 && H == 32) return launch_block_down_ws<D24_a, D24_s, D24_b, true>(a, st);
  if (Cin == 48 && H == 16) return launch_block_down_ws<D48_a, D48_s, D48_b, false>(a, st);
  if (Cin == 96 && H == 8) return launch_block_down_ws<D96_a, D96_s, D96_b, false>(a, st);
  return fail(QBNN_E_INVALID, "qbnn_block_down_i8_mc: unsupported geometry%s Cin=%ld H=%ld", "", Cin, H);
}

// =====================================================================================
// Weights-stationary fused kernels (layers whose block weights fit in LDS next to the tiles: 24 and 48 channels).
// Every workgroup walks a CONTIGUOUS range of work items, so consecutive items belong to the same MC sample and the
// block's sampled weights are copied into LDS (global_load_lds) once per sample change instead of once per conv.
// Nothing in the steady state waits on global memory at a barrier: barriers are LDS-only (lds_barrier), the next
// item's input sits in registers from the moment the current one is written to the tile (a whole item of cover), and
// the output stores are fire-and-forget.
// =====================================================================================
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <class C> struct WConv { static constexpr int BYTES = C::NT * C::KS * 1024; };

// whole packed conv (NT x KS fragment tiles of 1 KiB) -> LDS, verbatim
template <class C, int NWAVES>
__device__ __forceinline__ void dma_conv(uint8_t* dst, const int8_t* wq, int wave, int lane) {
  for (int f = wave; f < C::NT * C::KS; f += NWAVES)
    __builtin_amdgcn_global_load_lds(wq + ((int64_t)f * 64 + lane) * 16, (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
}

// The two halves of one MFMA pass (MB x NB output tiles of 32 pixels x 32 channels), separable so that a wave can
// park its accumulators across a barrier (ping-pong kernels) -- conv_core runs them back to back.
template <int MB, int NB> struct ConvAccMN { v16i acc[MB][NB]; int rsum[MB]; };
template <class C> using ConvAcc = ConvAccMN<C::MB, C::NB>;      // convs with equal blocking can share one accumulator set

template <class C>
__device__ __forceinline__ void conv_mfma_phase(const uint8_t* tile, const uint8_t* wconv, ConvAcc<C>& A, int pass, int lane) {
  const int r = lane & 31, h = lane >> 5;
  constexpr int U = C::KCHUNK, NCHUNK = C::KS / U;
  struct Frags { v4i w[U][C::NB]; v4i x[U][C::MB]; };
  const int mblk = pass / C::NBLKS, nblk = pass - mblk * C::NBLKS;
  const uint8_t* ap[C::MB];
#pragma unroll
  for (int mb = 0; mb < C::MB; ++mb) {
    const int m = (mblk * C::MB + mb) * 32 + r;
    const int g = m / (C::HO * C::HO), rem = m % (C::HO * C::HO);
    const int oh = rem / C::HO, ow = rem % C::HO;
    ap[mb] = tile + g * C::TILE_BYTES + ((oh * C::STRIDE + C::OFF0) * C::TW + ow * C::STRIDE + C::OFF0) * C::PIXB + 16 * h;
  }
  const uint8_t* wl = wconv + ((nblk * C::NB) * C::KS * 64 + lane) * 16;
#pragma unroll
  for (int mb = 0; mb < C::MB; ++mb) {
    A.rsum[mb] = 0;
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) A.acc[mb][nb][i] = 0;
  }
  auto load_chunk = [&](Frags& f, int c) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ks = c * U + u;
#pragma unroll
      for (int nb = 0; nb < C::NB; ++nb) f.w[u][nb] = *reinterpret_cast<const v4i*>(wl + (nb * C::KS + ks) * 1024);
#pragma unroll
      for (int mb = 0; mb < C::MB; ++mb) {
        f.x[u][mb] = load_xfrag<C>(ap[mb] + C::step_off(ks));
      }
    }
  };
  auto mfma_chunk = [&](const Frags& f, int c) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ks = c * U + u;
#pragma unroll
      for (int mb = 0; mb < C::MB; ++mb) {
        if (!C::USE_ONES) {
          const int m0 = (h ? C::piece_valid(ks, 1, 0) : C::piece_valid(ks, 0, 0)) ? 0x01010101 : 0;
          const int m1 = (h ? C::piece_valid(ks, 1, 1) : C::piece_valid(ks, 0, 1)) ? 0x01010101 : 0;
          int rs_ = A.rsum[mb];
          rs_ = __builtin_amdgcn_sdot4(f.x[u][mb].x, m0, rs_, false);
          rs_ = __builtin_amdgcn_sdot4(f.x[u][mb].y, m0, rs_, false);
          rs_ = __builtin_amdgcn_sdot4(f.x[u][mb].z, m1, rs_, false);
          rs_ = __builtin_amdgcn_sdot4(f.x[u][mb].w, m1, rs_, false);
          A.rsum[mb] = rs_;
        }
#pragma unroll
        for (int nb = 0; nb < C::NB; ++nb)
          A.acc[mb][nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.w[u][nb], f.x[u][mb], A.acc[mb][nb], 0, 0, 0);
      }
    }
  };
  Frags f0, f1;
  load_chunk(f0, 0);
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c) {
    Frags& cur = (c & 1) ? f1 : f0;
    Frags& nxt = (c & 1) ? f0 : f1;
    if (c + 1 < NCHUNK) load_chunk(nxt, c + 1);
    mfma_chunk(cur, c);
  }
}

// resfn(mb, nb, g4, po, c0): the residual dword of that output group (default: the functor's own load)
//   ahead(mb): called before the arithmetic of M-tile mb -- the place to request the residual of M-tile mb + 1
template <class C, class Epi, class ResFn, class AheadFn>
__device__ __forceinline__ void conv_epi_phase_with(const float* bias_lds, const QConv& p, Epi& epi, ConvAcc<C>& A, int pass, int lane,
                                                    ResFn resfn, AheadFn ahead) {
  const int r = lane & 31, h = lane >> 5;
  const int mblk = pass / C::NBLKS, nblk = pass - mblk * C::NBLKS;
#pragma unroll
  for (int mb = 0; mb < C::MB; ++mb) {
    int R;
    if (C::USE_ONES) {
      const int rv = A.acc[mb][C::ONES_TILE % C::NB][C::ONES_REG];
      const int ro = __shfl_xor(rv, 32);
      R = h ? ro : rv;
    } else {
      R = A.rsum[mb] + __shfl_xor(A.rsum[mb], 32);
    }
    const int zwr = p.z_w * R;
    const int po = epi.pixel((mblk * C::MB + mb) * 32 + r);
    ahead(mb);
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb) {
      float4 b4[4];             // (per n-tile: the whole table in registers costs 16 NB VGPRs through the epilogue)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        if (C::COUT % 32 != 0 && nb * 32 + 8 * g4 >= C::COUT) continue;
        b4[g4] = *reinterpret_cast<const float4*>(bias_lds + (nblk * C::NB + nb) * 32 + 8 * g4 + 4 * h);
      }
      uint32_t pre[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        if (C::COUT % 32 != 0 && nb * 32 + 8 * g4 >= C::COUT) continue;
        pre[g4] = resfn(mb, nb, g4, po, (nblk * C::NB + nb) * 32 + 8 * g4 + 4 * h);
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        if (C::COUT % 32 != 0 && nb * 32 + 8 * g4 >= C::COUT) continue;
        const int c0 = (nblk * C::NB + nb) * 32 + 8 * g4 + 4 * h;
        const float4 bb = b4[g4];
        const float v0 = __builtin_fmaf(bb.x, p.rcp, (float)(A.acc[mb][nb][4 * g4 + 0] - zwr)) * p.mult;
        const float v1 = __builtin_fmaf(bb.y, p.rcp, (float)(A.acc[mb][nb][4 * g4 + 1] - zwr)) * p.mult;
        const float v2 = __builtin_fmaf(bb.z, p.rcp, (float)(A.acc[mb][nb][4 * g4 + 2] - zwr)) * p.mult;
        const float v3 = __builtin_fmaf(bb.w, p.rcp, (float)(A.acc[mb][nb][4 * g4 + 3] - zwr)) * p.mult;
        epi.store(po, c0, v0, v1, v2, v3, pre[g4]);
      }
    }
  }
}

template <class C, class Epi>
__device__ __forceinline__ void conv_epi_phase(const float* bias_lds, const QConv& p, Epi& epi, ConvAcc<C>& A, int pass, int lane) {
  conv_epi_phase_with<C, Epi>(bias_lds, p, epi, A, pass, lane, [&](int, int, int, int po, int c0) { return epi.load(po, c0); }, [](int) {});
}

// (Measured and not adopted here: taking the window sums from a channel-sum table as the dense wide kernel does. Without
//  the v_dot4 chain in the loop the scheduler sinks every ring refill next to its use -- load, wait, MFMA -- 3x slower; with
//  the schedule pinned by sched_barrier the table version is 5 % slower than this one.)
// Streaming form of conv_passes for weights that come straight from L2 (no LDS staging): the wave's weight fragments
// (NB tiles x 1 KiB per k-step, consecutive in the packed layout) run through a register ring WD k-steps deep -- an L2
// round trip is 500-900 cycles, one k-step of MFMAs 32-200 -- while the pixel fragments come from LDS one k-step ahead.
// Fully unrolled over K; no barrier inside.  (The earlier form double-buffered chunks of <= 3 k-steps and stalled on
// every chunk: 18 stalls per 192-channel conv.)
template <class C, class Epi, int NWAVES>
__device__ __forceinline__ void conv_passes_stream(const uint8_t* tile, const int8_t* wq, const float* bias_lds, const QConv& p,
                                                   Epi& epi, int wave, int lane) {
  constexpr int WD = C::KS < C::WDEPTH ? C::KS : C::WDEPTH;
  const int r = lane & 31, h = lane >> 5;
  for (int pass = wave; pass < C::NPASS; pass += NWAVES) {
    const int mblk = pass / C::NBLKS, nblk = pass - mblk * C::NBLKS;
    const uint8_t* ap[C::MB];
#pragma unroll
    for (int mb = 0; mb < C::MB; ++mb) {
      const int m = (mblk * C::MB + mb) * 32 + r;
      const int g = m / (C::HO * C::HO), rem = m % (C::HO * C::HO);
      const int oh = rem / C::HO, ow = rem % C::HO;
      ap[mb] = tile + g * C::TILE_BYTES + ((oh * C::STRIDE + C::OFF0) * C::TW + ow * C::STRIDE + C::OFF0) * C::PIXB + 16 * h;
    }
    const int8_t* wbase = wq + ((int64_t)(nblk * C::NB) * C::KS * 64 + lane) * 16;
    ConvAcc<C> A;
#pragma unroll
    for (int mb = 0; mb < C::MB; ++mb) {
      A.rsum[mb] = 0;
#pragma unroll
      for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) A.acc[mb][nb][i] = 0;
    }
    v4i wr[WD][C::NB];
#pragma unroll
    for (int k = 0; k < WD; ++k)
#pragma unroll
      for (int nb = 0; nb < C::NB; ++nb) wr[k][nb] = *reinterpret_cast<const v4i*>(wbase + ((int64_t)(nb * C::KS + k) * 64) * 16);
    v4i x0[C::MB], x1[C::MB];
#pragma unroll
    for (int mb = 0; mb < C::MB; ++mb) x0[mb] = load_xfrag<C>(ap[mb] + C::step_off(0));
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      v4i* xc = (ks & 1) ? x1 : x0;
      v4i* xn = (ks & 1) ? x0 : x1;
      if (ks + 1 < C::KS) {
#pragma unroll
        for (int mb = 0; mb < C::MB; ++mb) xn[mb] = load_xfrag<C>(ap[mb] + C::step_off(ks + 1));
      }
      v4i w[C::NB];
#pragma unroll
      for (int nb = 0; nb < C::NB; ++nb) {
        w[nb] = wr[ks % WD][nb];
        if (ks + WD < C::KS) wr[ks % WD][nb] = *reinterpret_cast<const v4i*>(wbase + ((int64_t)(nb * C::KS + ks + WD) * 64) * 16);
      }
#pragma unroll
      for (int mb = 0; mb < C::MB; ++mb) {
        if (!C::USE_ONES) {
          const int m0 = (h ? C::piece_valid(ks, 1, 0) : C::piece_valid(ks, 0, 0)) ? 0x01010101 : 0;
          const int m1 = (h ? C::piece_valid(ks, 1, 1) : C::piece_valid(ks, 0, 1)) ? 0x01010101 : 0;
          int rs_ = A.rsum[mb];
          rs_ = __builtin_amdgcn_sdot4(xc[mb].x, m0, rs_, false);
          rs_ = __builtin_amdgcn_sdot4(xc[mb].y, m0, rs_, false);
          rs_ = __builtin_amdgcn_sdot4(xc[mb].z, m1, rs_, false);
          rs_ = __builtin_amdgcn_sdot4(xc[mb].w, m1, rs_, false);
          A.rsum[mb] = rs_;
        }
#pragma unroll
        for (int nb = 0; nb < C::NB; ++nb)
          A.acc[mb][nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[nb], xc[mb], A.acc[mb][nb], 0, 0, 0);
      }
    }
    conv_epi_phase<C, Epi>(bias_lds, p, epi, A, pass, lane);
  }
}

// conv over an LDS-resident tile with LDS-resident weights; no barrier inside.  Same arithmetic and epilogue
// interface as conv_passes / conv_lds.
template <class C, class Epi, int NWAVES>
__device__ __forceinline__ void conv_core(const uint8_t* tile, const uint8_t* wconv, const float* bias_lds, const QConv& p,
                                          Epi& epi, int wave, int lane) {
  const int r = lane & 31, h = lane >> 5;
  if constexpr (C::ROWREUSE) {
    static_assert(C::USE_ONES && C::NT == 1, "row-reuse path");
    float4 b4[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      if (8 * g4 < C::COUT) b4[g4] = *reinterpret_cast<const float4*>(bias_lds + 8 * g4 + 4 * h);
    constexpr int NR = C::MB + C::KSZ - 1;
    v4i w[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) w[ks] = *reinterpret_cast<const v4i*>(wconv + lane * 16 + ks * 1024);
    for (int pass = wave; pass < C::NPASS; pass += NWAVES) {
      const int m0 = pass * C::MB * 32;
      const int g = m0 / (C::HO * C::HO), oh0 = (m0 % (C::HO * C::HO)) / C::HO;
      const uint8_t* base = tile + g * C::TILE_BYTES + ((oh0 + C::OFF0) * C::TW + r + C::OFF0) * C::PIXB + 16 * h;
      v4i x[NR][C::SPR];
#pragma unroll
      for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int t = 0; t < C::SPR; ++t) {
          const v2i lo = *reinterpret_cast<const v2i*>(base + j * C::PITCH + t * 32);
          const v2i hi = *reinterpret_cast<const v2i*>(base + j * C::PITCH + t * 32 + 8);
          x[j][t] = v4i{lo.x, lo.y, hi.x, hi.y};
        }
      v16i acc[C::MB];
      const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int mb = 0; mb <= C::MB; ++mb) {
        if (mb < C::MB) {
#pragma unroll
          for (int kh = 0; kh < C::KSZ; ++kh)
#pragma unroll
            for (int t = 0; t < C::SPR; ++t)
              acc[mb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[kh * C::SPR + t], x[mb + kh][t], (kh == 0 && t == 0) ? zero16 : acc[mb], 0, 0, 0);
        }
        if (mb > 0) {
          const int e = mb - 1;
          const int rv = acc[e][C::ONES_REG];
          const int ro = __shfl_xor(rv, 32);
          const int zwr = p.z_w * (h ? ro : rv);
          const int po = epi.pixel(m0 + e * 32 + r);
          uint32_t pre[4];
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4)
            if (8 * g4 < C::COUT) pre[g4] = epi.load(po, 8 * g4 + 4 * h);
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            if (8 * g4 >= C::COUT) continue;
            const float4 bb = b4[g4];
            const float v0 = __builtin_fmaf(bb.x, p.rcp, (float)(acc[e][4 * g4 + 0] - zwr)) * p.mult;
            const float v1 = __builtin_fmaf(bb.y, p.rcp, (float)(acc[e][4 * g4 + 1] - zwr)) * p.mult;
            const float v2 = __builtin_fmaf(bb.z, p.rcp, (float)(acc[e][4 * g4 + 2] - zwr)) * p.mult;
            const float v3 = __builtin_fmaf(bb.w, p.rcp, (float)(acc[e][4 * g4 + 3] - zwr)) * p.mult;
            epi.store(po, 8 * g4 + 4 * h, v0, v1, v2, v3, pre[g4]);
          }
        }
        if (mb > 0 && mb < C::MB) {
          // in-order issue: the epilogue of row mb-1 only hides under the MFMAs of row mb if it sits between them
#pragma unroll
          for (int i = 0; i < C::KS; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, Epi::VALU_PER_MFMA, 0);
          }
        }
      }
    }
  } else {
    for (int pass = wave; pass < C::NPASS; pass += NWAVES) {
      ConvAcc<C> A;
      conv_mfma_phase<C>(tile, wconv, A, pass, lane);
      conv_epi_phase<C, Epi>(bias_lds, p, epi, A, pass, lane);
    }
  }
}

// contiguous item range of workgroup b out of nb
__device__ __forceinline__ void item_range(int n_items, int b, int nb, int& begin, int& count) {
  const int q = n_items / nb, rm = n_items - q * nb;
  begin = b * q + (b < rm ? b : rm);
  count = q + (b < rm ? 1 : 0);
}

// Interleaved, XCD-aware walk for the kernels that stream their weights per item (no weights-stationary LDS copy).
// Workgroup b runs on XCD b % 8 (round-robin dispatch, one workgroup per CU), and every XCD has its own 4 MiB L2.  XCD x
// takes the x-th eighth of the sample-major item list, and its 32 workgroups walk that range interleaved: at any time
// they sit on the same one or two MC samples, so a sample's weights (130 - 660 KiB) are filled into one L2 (two where a
// range boundary cuts a sample) once and every later read is an L2 hit.  (Plain `item = b + i * gridDim` spreads each
// sample over all eight L2s: 8x the fill traffic, and each L2 churns through the weights of 8 samples at a time.)
#ifndef QBNN_XCDS
#define QBNN_XCDS 8
#endif
struct ItemWalk {
  int first, per, count;
  __device__ __forceinline__ ItemWalk(int n_items, int b, int nb) {
    if (QBNN_XCDS > 1 && (nb % QBNN_XCDS) == 0) {
      int xb, xn;
      item_range(n_items, b % QBNN_XCDS, QBNN_XCDS, xb, xn);
      const int j = b / QBNN_XCDS;
      per = nb / QBNN_XCDS;
      first = xb + j;
      count = j < xn ? (xn - j + per - 1) / per : 0;
    } else {
      first = b; per = nb;
      count = b < n_items ? (n_items - b + nb - 1) / nb : 0;
    }
  }
  __device__ __forceinline__ int item(int it) const { return first + it * per; }
};

// LDSW = true : weights-stationary as described above (contiguous item ranges).
// LDSW = false: the block's weights are too large for LDS -- every wave streams its fragments from L2 (conv_passes) and
//               the workgroups walk the items interleaved per XCD (ItemWalk), so that the workgroups sharing an L2
//               work on the same MC sample at a time and its weights stay hot there.  Same barrier / prefetch scheme.
// STEM = true (layer 1 only): the network's first conv (3 -> 24 channels, on the pre-gathered 27-tap patches) runs inside
//               the same kernel -- its output never goes to HBM (that tensor is the largest of the network: 629 MB per
//               100-sample step written and read back).  The item's input is then its image's patch block (32 KiB,
//               shared by all samples, L2-resident), staged in a dense LDS tile; conv0's epilogue writes the X tile.
template <class C, int NBLK, bool LDSW, bool STEM = false, int NM = 1>
__global__ __launch_bounds__(BLK_THREADS) void block_chain_ws_kernel(const ArgsArr<ChainArgs<NBLK>, NM> all) {
  const ChainArgs<NBLK>& a = all.m[NM == 1 ? 0 : blockIdx.y];
  using C0 = ConvCfg<32, 24, 1, 1, 32, 0, 1, 4, 1>;      // layer 0 on the patch tensor: K = 27 -> 32, one k-step
  static_assert(!STEM || (LDSW && C::CIN == 24 && C::HIN == 32 && C::G == 1), "the fused stem feeds the 32x32x24 chain");
  static_assert(C::CIN == C::COUT && C::STRIDE == 1 && C::KSZ == 3 && C::HALO == 1, "identity BasicBlock geometry");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr int NTHR = BLK_THREADS, NWV = BLK_WAVES;
  constexpr int TILES = C::G * C::TILE_BYTES + C::TILE_SLACK;
  constexpr int WB = LDSW ? WConv<C>::BYTES : 0;
  uint8_t* xt = smem;
  uint8_t* tt = smem + TILES;
  uint8_t* wl = smem + 2 * TILES;                                            // [NBLK][2] whole convs
  float* bias_lds = reinterpret_cast<float*>(wl + 2 * NBLK * WB);            // [NBLK][2][COUT]
  uint8_t* im = reinterpret_cast<uint8_t*>(bias_lds + NBLK * 2 * C::COUT);   // STEM: patch tile [1024][32], stem weights, stem bias
  uint8_t* wl0 = im + C0::TILE_BYTES;
  float* bias0 = reinterpret_cast<float*>(wl0 + WConv<C0>::BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar control flow and addresses

  constexpr int CPR = C::ROWB / 16, CPI = C::HIN * CPR, NCH = C::G * CPI;   // 16-byte chunks of one item
  constexpr int NCH_IN = STEM ? C0::TILE_BYTES / 16 : NCH;                  // ... of its input (the patch block when STEM)
  constexpr int PER_T = (NCH_IN + NTHR - 1) / NTHR, PER_TO = (NCH + NTHR - 1) / NTHR;
  const int groups = (a.B + C::G - 1) / C::G;
  int begin = 0, count;
  const ItemWalk walk(a.n_samples * groups, blockIdx.x, gridDim.x);
  if (LDSW) item_range(a.n_samples * groups, blockIdx.x, gridDim.x, begin, count);
  else count = walk.count;
  auto item_at = [&](int it) { return LDSW ? begin + it : walk.item(it); };

  zero_halo<C::TW, C::PIXB, C::TILE_BYTES, C::G, NTHR>(xt, tid);
  zero_halo<C::TW, C::PIXB, C::TILE_BYTES, C::G, NTHR>(tt, tid);
#pragma unroll
  for (int k = 0; k < NBLK; ++k) {
    load_bias<C::COUT, NTHR>(bias_lds + (2 * k) * C::COUT, a.blk[k].a.bias, tid);
    load_bias<C::COUT, NTHR>(bias_lds + (2 * k + 1) * C::COUT, a.blk[k].b.bias, tid);
  }
  if (STEM) load_bias<C0::COUT, NTHR>(bias0, a.stem.bias, tid);

  v4i pre[PER_T];
  auto fetch = [&](int item) {
    const int s = item / groups, img0 = (item - s * groups) * C::G;
    if constexpr (STEM) {
      const uint8_t* xs = reinterpret_cast<const uint8_t*>(a.stem_x) + (int64_t)(img0 < a.B ? img0 : 0) * C0::TILE_BYTES;
#pragma unroll
      for (int j = 0; j < PER_T; ++j) pre[j] = *reinterpret_cast<const v4i*>(xs + (int64_t)(tid + j * NTHR) * 16);
      return;
    }
    const uint8_t* xs = a.x + (int64_t)s * a.x_ss;
#pragma unroll
    for (int j = 0; j < PER_T; ++j) {
      const int i = tid + j * NTHR;
      const int g = i / CPI, rem = i - g * CPI;
      const bool ok = (i < NCH) && (img0 + g < a.B);
      const int64_t off = ok ? ((int64_t)(img0 + g) * C::HIN) * C::ROWB + (int64_t)rem * 16 : 0;
      pre[j] = *reinterpret_cast<const v4i*>(xs + off);
    }
  };
  // registers -> centred X tile interior.  Runs right after the same thread has read these very chunks out (end of
  // the previous item), so no barrier separates the two.
  auto write_tile = [&](int item) {
    const int s = item / groups, img0 = (item - s * groups) * C::G;
    if constexpr (STEM) {          // the patches are already centred; conv0 produces the X tile
#pragma unroll
      for (int j = 0; j < PER_T; ++j) *reinterpret_cast<v4i*>(im + (tid + j * NTHR) * 16) = pre[j];
      return;
    }
    const uint32_t z4 = (uint32_t)a.z_in * 0x01010101u;
#pragma unroll
    for (int j = 0; j < PER_T; ++j) {
      const int i = tid + j * NTHR;
      if (i < NCH) {
        const int g = i / CPI, rem = i - g * CPI, row = rem / CPR, within = rem - row * CPR;
        const bool ok = img0 + g < a.B;
        const v4i v = pre[j];
        uint8_t* d = xt + g * C::TILE_BYTES + (row + 1) * C::PITCH + C::row_chunk_off(within);
        *reinterpret_cast<v2i*>(d) = ok ? v2i{(int)sub_bytes(v.x, z4), (int)sub_bytes(v.y, z4)} : v2i{0, 0};
        *reinterpret_cast<v2i*>(d + 8) = ok ? v2i{(int)sub_bytes(v.z, z4), (int)sub_bytes(v.w, z4)} : v2i{0, 0};
      }
    }
  };
  if (count <= 0) return;
  fetch(item_at(0));
  write_tile(item_at(0));
  int cur_s = -1;
  QBNN_STAMP_DECL
  for (int it = 0; it < count; ++it) {
    QBNN_STAMP_START();
    const int item = item_at(it);
    const int s = item / groups, img0 = (item - s * groups) * C::G;
    const bool more = it + 1 < count;
    // the next item's input: in flight for the whole of this item (unconditional, so the wait counts at its use are
    // exact: the last iteration re-reads its own item and drops it)
    fetch(more ? item_at(it + 1) : item);
    if (LDSW && s != cur_s) {    // workgroup-uniform; at most a few times per launch
      __syncthreads();           // every wave is done with the previous sample's weights (and the prologue's LDS writes)
#pragma unroll
      for (int k = 0; k < NBLK; ++k) {
        dma_conv<C, NWV>(wl + (2 * k) * WB, a.blk[k].a.w + (int64_t)s * a.blk[k].a.w_ss, wave, lane);
        dma_conv<C, NWV>(wl + (2 * k + 1) * WB, a.blk[k].b.w + (int64_t)s * a.blk[k].b.w_ss, wave, lane);
      }
      if (STEM) dma_conv<C0, NWV>(wl0, a.stem.w + (int64_t)s * a.stem.w_ss, wave, lane);
      dma_barrier();             // vmcnt(0) + barrier: the weights have landed
      cur_s = s;
    }
    QBNN_STAMP_AT(0);
    lds_barrier();
    QBNN_STAMP_AT(1);
    if constexpr (STEM) {        // layers.0 (ConvReLU2d): patch tile -> X tile, centred on its own zero point (= a.z_in)
      EpiTile<C::HO, C::PIXB, C::TILE_BYTES> epi{xt, a.stem};
      conv_core<C0, decltype(epi), NWV>(im, wl0, bias0, a.stem, epi, wave, lane);
      lds_barrier();
    }
#pragma unroll
    for (int k = 0; k < NBLK; ++k) {
      const BlockParams& bp = a.blk[k];
      {
        EpiTile<C::HO, C::PIXB, C::TILE_BYTES> epi{tt, bp.a};
        if constexpr (LDSW) conv_core<C, decltype(epi), NWV>(xt, wl + (2 * k) * WB, bias_lds + (2 * k) * C::COUT, bp.a, epi, wave, lane);
        else conv_passes<C, decltype(epi), NWV>(xt, bp.a.w + (int64_t)s * bp.a.w_ss, bias_lds + (2 * k) * C::COUT, bp.a, epi, wave, lane);
      }
      QBNN_STAMP_AT(2);
      lds_barrier();
      QBNN_STAMP_AT(3);
      {
        EpiTileResInPlace<C::HO, C::PIXB, C::TILE_BYTES> epi{xt, bp.b, bp.add};
        if constexpr (LDSW) conv_core<C, decltype(epi), NWV>(tt, wl + (2 * k + 1) * WB, bias_lds + (2 * k + 1) * C::COUT, bp.b, epi, wave, lane);
        else conv_passes<C, decltype(epi), NWV>(tt, bp.b.w + (int64_t)s * bp.b.w_ss, bias_lds + (2 * k + 1) * C::COUT, bp.b, epi, wave, lane);
      }
      QBNN_STAMP_AT(4);
      lds_barrier();
      QBNN_STAMP_AT(5);
    }
    // ---- X tile interior (centred by the last add's zero point) -> quint8 registers; next item's input -> X tile;
    //      registers -> HBM.  The stores are issued last so that nothing ever waits on them: the only vmcnt waits
    //      of the loop are for the input loads issued a whole item earlier.
    {
      const uint32_t z4 = (uint32_t)a.blk[NBLK - 1].add.z_o * 0x01010101u;
      uint8_t* ys = a.y + (int64_t)s * a.y_ss;
      v4i outv[PER_TO];
#pragma unroll
      for (int j = 0; j < PER_TO; ++j) {
        const int i = tid + j * NTHR;
        if (i < NCH) {
          const int g = i / CPI, rem = i - g * CPI, row = rem / CPR, within = rem - row * CPR;
          const uint8_t* d = xt + g * C::TILE_BYTES + (row + 1) * C::PITCH + C::row_chunk_off(within);
          const v2i lo = *reinterpret_cast<const v2i*>(d), hi = *reinterpret_cast<const v2i*>(d + 8);
          outv[j] = v4i{(int)add_bytes(lo.x, z4), (int)add_bytes(lo.y, z4), (int)add_bytes(hi.x, z4), (int)add_bytes(hi.y, z4)};
        }
      }
      if (more) write_tile(item_at(it + 1));
#pragma unroll
      for (int j = 0; j < PER_TO; ++j) {
        const int i = tid + j * NTHR;
        if (i < NCH) {
          const int g = i / CPI, rem = i - g * CPI;
          if (img0 + g < a.B)
            *reinterpret_cast<v4i*>(ys + ((int64_t)(img0 + g) * C::HIN) * C::ROWB + (int64_t)rem * 16) = outv[j];
        }
      }
    }
    QBNN_STAMP_AT(6);
  }
#ifdef QBNN_STAMP
  if (a.dbg && (tid & 63) == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(a.dbg + wave * 8 + i, st_acc[i]);
#endif
}

// =====================================================================================
// Ping-pong identity chain (48 channels): the workgroup's 8 waves form two groups of 4 (one wave per SIMD each).
// Each group owns one work item at a time (its own X0 / X1 / T tiles; the block weights in LDS are shared) and walks
// the phase sequence   M_a  E_a  M_b  E_b   (M = the conv's MFMA K loop into parked accumulators, E = its
// requantising epilogue), one phase per barrier interval.  Group 1 runs one interval behind group 0, so in every
// interval each SIMD holds one wave issuing MFMAs and one wave issuing epilogue VALU -- the matrix and vector pipes
// overlap instead of alternating.  Input write / output read-out ride along: the finished item k-1 leaves from
// X[(k-1)&1] during M_a(k); the input of item k+1 enters X[(k+1)&1] during E_a(k).
// =====================================================================================
template <class C, int NBLK>
__global__ __launch_bounds__(512) void block_chain_pp_kernel(const ChainArgs<NBLK> a) {
  static_assert(C::CIN == C::COUT && C::STRIDE == 1 && C::KSZ == 3 && C::HALO == 1, "identity BasicBlock geometry");
  static_assert(!C::ROWREUSE && C::NPASS == 4, "one MFMA pass per wave of a 4-wave group");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr int GT = 256;                                                   // threads per group
  constexpr int TILES = C::G * C::TILE_BYTES + C::TILE_SLACK;
  constexpr int WB = WConv<C>::BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar control flow and addresses
  const int grp = wave >> 2, lw = wave & 3, ltid = tid & (GT - 1);
  uint8_t* xg = smem + grp * 3 * TILES;                                     // X0, X1, T of this group
  uint8_t* tt = xg + 2 * TILES;
  uint8_t* wl = smem + 6 * TILES;                                           // [NBLK][2] whole convs
  float* bias_lds = reinterpret_cast<float*>(wl + 2 * NBLK * WB);           // [NBLK][2][COUT]

  constexpr int CPR = C::ROWB / 16, CPI = C::HIN * CPR, NCH = C::G * CPI;
  constexpr int PER_T = (NCH + GT - 1) / GT;
  const int groups = (a.B + C::G - 1) / C::G;                               // items per sample (host: even)
  int pbegin, pcount;                                                       // contiguous range of item PAIRS
  item_range(a.n_samples * groups / 2, blockIdx.x, gridDim.x, pbegin, pcount);

  zero_halo<C::TW, C::PIXB, C::TILE_BYTES, C::G, GT>(xg, ltid);
  zero_halo<C::TW, C::PIXB, C::TILE_BYTES, C::G, GT>(xg + TILES, ltid);
  zero_halo<C::TW, C::PIXB, C::TILE_BYTES, C::G, GT>(tt, ltid);
#pragma unroll
  for (int k = 0; k < NBLK; ++k) {
    load_bias<C::COUT, 512>(bias_lds + (2 * k) * C::COUT, a.blk[k].a.bias, tid);
    load_bias<C::COUT, 512>(bias_lds + (2 * k + 1) * C::COUT, a.blk[k].b.bias, tid);
  }
  if (pcount <= 0) return;

  auto item_of = [&](int k) { return 2 * (pbegin + k) + grp; };
  v4i pre[PER_T];
  auto fetch = [&](int item) {
    const int s = item / groups, img0 = (item - s * groups) * C::G;
    const uint8_t* xs = a.x + (int64_t)s * a.x_ss;
#pragma unroll
    for (int j = 0; j < PER_T; ++j) {
      const int i = ltid + j * GT;
      const int g = i / CPI, rem = i - g * CPI;
      const bool ok = (i < NCH) && (img0 + g < a.B);
      const int64_t off = ok ? ((int64_t)(img0 + g) * C::HIN) * C::ROWB + (int64_t)rem * 16 : 0;
      pre[j] = *reinterpret_cast<const v4i*>(xs + off);
    }
  };
  auto write_tile = [&](uint8_t* xt, int item) {
    const int s = item / groups, img0 = (item - s * groups) * C::G;
    const uint32_t z4 = (uint32_t)a.z_in * 0x01010101u;
#pragma unroll
    for (int j = 0; j < PER_T; ++j) {
      const int i = ltid + j * GT;
      if (i < NCH) {
        const int g = i / CPI, rem = i - g * CPI, row = rem / CPR, within = rem - row * CPR;
        const bool ok = img0 + g < a.B;
        const v4i v = pre[j];
        uint8_t* d = xt + g * C::TILE_BYTES + (row + 1) * C::PITCH + C::row_chunk_off(within);
        *reinterpret_cast<v2i*>(d) = ok ? v2i{(int)sub_bytes(v.x, z4), (int)sub_bytes(v.y, z4)} : v2i{0, 0};
        *reinterpret_cast<v2i*>(d + 8) = ok ? v2i{(int)sub_bytes(v.z, z4), (int)sub_bytes(v.w, z4)} : v2i{0, 0};
      }
    }
  };
  auto store_tile = [&](const uint8_t* xt, int item) {
    const int s = item / groups, img0 = (item - s * groups) * C::G;
    const uint32_t z4 = (uint32_t)a.blk[NBLK - 1].add.z_o * 0x01010101u;
    uint8_t* ys = a.y + (int64_t)s * a.y_ss;
#pragma unroll
    for (int j = 0; j < PER_T; ++j) {
      const int i = ltid + j * GT;
      if (i < NCH) {
        const int g = i / CPI, rem = i - g * CPI, row = rem / CPR, within = rem - row * CPR;
        if (img0 + g < a.B) {
          const uint8_t* d = xt + g * C::TILE_BYTES + (row + 1) * C::PITCH + C::row_chunk_off(within);
          const v2i lo = *reinterpret_cast<const v2i*>(d), hi = *reinterpret_cast<const v2i*>(d + 8);
          v4i v = {(int)add_bytes(lo.x, z4), (int)add_bytes(lo.y, z4), (int)add_bytes(hi.x, z4), (int)add_bytes(hi.y, z4)};
          *reinterpret_cast<v4i*>(ys + ((int64_t)(img0 + g) * C::HIN) * C::ROWB + (int64_t)rem * 16) = v;
        }
      }
    }
  };

  fetch(item_of(0));
  write_tile(xg, item_of(0));
  fetch(item_of(pcount > 1 ? 1 : 0));
  constexpr int NPH = 4 * NBLK;
  const int n_int = NPH * pcount + 1;                  // group 1 finishes one interval after group 0
  ConvAcc<C> A;
  int cur_s = -1;
#pragma unroll 1
  for (int t = 0; t < n_int; ++t) {
    // ---- interval boundary.  Group 0 enters a new pair every NPH intervals; if that pair belongs to another MC
    // sample the block weights are replaced here -- group 1 is in its last epilogue (no weight reads) meanwhile.
    const int k0 = t / NPH;
    bool reload = false;
    int s0 = cur_s;
    if (t - k0 * NPH == 0 && k0 < pcount) { s0 = (2 * (pbegin + k0)) / groups; reload = s0 != cur_s; }
    if (reload) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NBLK; ++k) {
        dma_conv<C, 8>(wl + (2 * k) * WB, a.blk[k].a.w + (int64_t)s0 * a.blk[k].a.w_ss, wave, lane);
        dma_conv<C, 8>(wl + (2 * k + 1) * WB, a.blk[k].b.w + (int64_t)s0 * a.blk[k].b.w_ss, wave, lane);
      }
      dma_barrier();
      cur_s = s0;
    } else {
      lds_barrier();
    }
    const int lt = t - grp;
    if (lt < 0 || lt >= NPH * pcount) continue;
    const int k = lt / NPH, ph = lt - k * NPH, blk = ph >> 2, q = ph & 3;
    uint8_t* X = xg + (k & 1) * TILES;
    uint8_t* Xo = xg + ((k + 1) & 1) * TILES;
    const BlockParams& bp = a.blk[blk];
    if (q == 0) {
      conv_mfma_phase<C>(X, wl + (2 * blk) * WB, A, lw, lane);
      if (blk == 0 && k > 0) store_tile(Xo, item_of(k - 1));
    } else if (q == 1) {
      EpiTile<C::HO, C::PIXB, C::TILE_BYTES> epi{tt, bp.a};
      conv_epi_phase<C, decltype(epi)>(bias_lds + (2 * blk) * C::COUT, bp.a, epi, A, lw, lane);
      if (blk == 0 && k + 1 < pcount) {
        write_tile(Xo, item_of(k + 1));
        fetch(item_of(k + 2 < pcount ? k + 2 : k + 1));
      }
    } else if (q == 2) {
      conv_mfma_phase<C>(tt, wl + (2 * blk + 1) * WB, A, lw, lane);
    } else {
      EpiTileResInPlace<C::HO, C::PIXB, C::TILE_BYTES> epi{X, bp.b, bp.add};
      conv_epi_phase<C, decltype(epi)>(bias_lds + (2 * blk + 1) * C::COUT, bp.b, epi, A, lw, lane);
    }
  }
  lds_barrier();
  store_tile(xg + ((pcount - 1) & 1) * TILES, item_of(pcount - 1));
}

template <class C, int NBLK> constexpr int chain_pp_lds() {
  return 6 * (C::G * C::TILE_BYTES + C::TILE_SLACK) + 2 * NBLK * WConv<C>::BYTES + NBLK * 2 * C::COUT * 4;
}

template <class C, int NBLK>
static int launch_block_chain_pp(const ChainArgs<NBLK>& a, hipStream_t st) {
  constexpr int LDS = chain_pp_lds<C, NBLK>();
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static std::atomic<uint64_t> attr{0};
  if (int rc_attr = ensure_dyn_lds((const void*)block_chain_pp_kernel<C, NBLK>, attr, LDS)) return rc_attr;
  const int groups = (a.B + C::G - 1) / C::G;
  const int n_pairs = a.n_samples * groups / 2;
  const int grid = n_pairs < 256 ? n_pairs : 256;
  hipLaunchKernelGGL((block_chain_pp_kernel<C, NBLK>), dim3(grid), dim3(512), LDS, st, a);
  return check_launch("qbnn_block_chain_i8_mc");
}

// =====================================================================================
// Wide identity block (96 / 192 channels): the block's weights (162 / 663 KiB per MC sample) neither fit in LDS nor
// can every wave afford to stream its own copy from L2, so they pass ONCE per work item through a two-slab LDS ring
// (global_load_lds) shared by the 8 waves.  To leave room for the ring the stem.0 output T overwrites the input tile X
// IN PLACE: each conv runs as two workgroup-wide phases,
//     M: every wave accumulates its MB x NB output tiles over all weight slabs (reads the tile),
//     E: after a barrier, every wave requantises its accumulators and writes them over the tile,
// and the residual operand of the Add is re-read from global memory (the block input, L2-hot, quint8) instead of
// being kept in LDS.  One pass per wave: C::NPASS == 8.
// The tile is DENSE.  With the 1-pixel halo an 8x8 / 4x4 map costs 1.56x / 2.25x its size
// in LDS; stored dense ([image][oh][ow][C + 16]) twice as many images fit next to the weight ring (8 at 96 channels,
// 16 at 192), which doubles the MFMA work per weight slab (the slab's LDS-DMA latency hides behind it) and halves the
// weight bytes moved per image.  (A halo'd variant with 8 images per item was 15 % slower at 192 channels.)  Zero padding is then a per-lane address choice: a tap that falls outside the map
// reads a line of zeros instead.  The tap's position is a function of the slab / k-step only, so this costs a few
// VALU operations per slab.
// =====================================================================================
template <class C> struct DenseTile {
  static constexpr int IMG = C::HO * C::HO * C::PIXB;
  static constexpr int BYTES = C::G * IMG;                  // followed by the zero line (C::PIXB bytes)
  static constexpr int TPS = C::SLK / C::SPT;               // taps per weight slab
  static_assert(C::PADB > 0 && C::SLK % C::SPT == 0 && C::STRIDE == 1 && C::KSZ == 3, "slabs are whole taps");
};

template <class C, int NWV, class FNext>
__device__ __forceinline__ void conv_ring_mfma_dense(const uint8_t* tile, uint8_t* rbase, int& rcur, const int8_t* wq, ConvAcc<C>& A,
                                                     int wave, int lane, FNext prefetch_next) {
  static_assert(C::NPASS == NWV, "one pass per wave");
  using DT = DenseTile<C>;
  const int r = lane & 31, h = lane >> 5;
  const int mblk = wave / C::NBLKS, nblk = wave - mblk * C::NBLKS;
  int pix0[C::MB], poh[C::MB], pow_[C::MB];               // this lane's pixel per M-tile: byte offset, row, column
#pragma unroll
  for (int mb = 0; mb < C::MB; ++mb) {
    const int m = (mblk * C::MB + mb) * 32 + r;
    const int rem = m % (C::HO * C::HO);
    poh[mb] = rem / C::HO; pow_[mb] = rem % C::HO;
    pix0[mb] = m * C::PIXB + 16 * h;
  }
  const uint8_t* zline = tile + DT::BYTES + 16 * h;
#pragma unroll
  for (int mb = 0; mb < C::MB; ++mb) {
    A.rsum[mb] = 0;
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) A.acc[mb][nb][i] = 0;
  }
  struct Frags { v4i w[C::NB]; v4i x[C::MB]; };              // one k-step per buffer
  QBNN_INNER_T0();
#pragma unroll 1
  for (int slab = 0; slab < C::NSLAB; ++slab) {
    dma_barrier();            // slab landed; everyone is done with the other buffer; slab 0: tile complete
    QBNN_INNER_AT(0);
    uint8_t* other = rbase + (rcur ^ 1) * C::SLAB_BYTES;
    if (slab + 1 < C::NSLAB) dma_slab<C, NWV>(other, wq, slab + 1, wave, lane);
    else prefetch_next(other);
    const uint8_t* wl = rbase + rcur * C::SLAB_BYTES + ((nblk * C::NB) * C::SLK * 64 + lane) * 16;
    rcur ^= 1;
    const uint8_t* tb[C::MB][DT::TPS];
#pragma unroll
    for (int tp = 0; tp < DT::TPS; ++tp) {
      const int tap = slab * DT::TPS + tp, kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
      for (int mb = 0; mb < C::MB; ++mb) {
        const bool ok = (unsigned)(poh[mb] + kh - 1) < (unsigned)C::HO && (unsigned)(pow_[mb] + kw - 1) < (unsigned)C::HO;
        tb[mb][tp] = ok ? tile + pix0[mb] + ((kh - 1) * C::HO + (kw - 1)) * C::PIXB : zline;
      }
    }
    auto load_step = [&](Frags& f, int j) {
#pragma unroll
      for (int nb = 0; nb < C::NB; ++nb) f.w[nb] = *reinterpret_cast<const v4i*>(wl + (nb * C::SLK + j) * 1024);
#pragma unroll
      for (int mb = 0; mb < C::MB; ++mb) f.x[mb] = load_xfrag<C>(tb[mb][j / C::SPT] + (j % C::SPT) * 32);
    };
    auto mfma_step = [&](const Frags& f) {
#pragma unroll
      for (int mb = 0; mb < C::MB; ++mb) {
        // (no window sum here: 4 v_dot4 per fragment cost 11 % of the kernel; the epilogue gathers it from the
        //  per-pixel channel sums kept beside the tile, see window_sum_from_table)
#pragma unroll
        for (int nb = 0; nb < C::NB; ++nb)
          A.acc[mb][nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.w[nb], f.x[mb], A.acc[mb][nb], 0, 0, 0);
      }
    };
    Frags f0, f1;
    load_step(f0, 0);
#pragma unroll
    for (int j = 0; j < C::SLK; ++j) {
      Frags& cur = (j & 1) ? f1 : f0;
      Frags& nxt = (j & 1) ? f0 : f1;
      if (j + 1 < C::SLK) load_step(nxt, j + 1);
      mfma_step(cur);
    }
    QBNN_INNER_AT(1);
  }
  QBNN_INNER_FLUSH();
}

// dense-tile epilogues: (b') centred stem.0 output, (c') Add(residual from global) + ReLU, centred block output
template <int PIXB>
struct EpiDenseTile {
  uint8_t* dst; QConv p;
  mutable int csum;            // sum of the centred bytes this lane has written since the last flush (channel-sum table)
  __device__ __forceinline__ int pixel(int m) const { return m * PIXB; }
  __device__ __forceinline__ uint32_t load(int, int) const { return 0u; }
  __device__ __forceinline__ void store(int po, int c0, float v0, float v1, float v2, float v3, uint32_t) const {
    const uint32_t pk = pack_rne_u8(v0, v1, v2, v3, p.vhi);      // ConvReLU2d: p.vlo == 0
    *reinterpret_cast<uint32_t*>(dst + po + c0) = pk;
    csum = __builtin_amdgcn_sdot4((int)pk, 0x01010101, csum, false);
  }
};

// Window sum R(p) = sum over the 3x3 window and all channels of the centred tile bytes, needed because sampled weights
// have a non-zero zero point (sum x'(W - z_w) = acc - z_w R).  The dense-tile kernel keeps S(p) = channel sum of pixel p in
// a small LDS table, maintained where the tile is written (one v_dot4 per dword written, LDS atomic add), and gathers
// the <= 9 neighbours here -- instead of 4 v_dot4 per pixel fragment inside the MFMA loop (x27 / x54 per conv).
// Leaves R in A.rsum so that conv_epi_phase's (rsum + rsum of lane ^ 32) yields it.
template <class C>
__device__ __forceinline__ void window_sum_from_table(const int* tab, ConvAcc<C>& A, int pass, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const int mblk = pass / C::NBLKS;
#pragma unroll
  for (int mb = 0; mb < C::MB; ++mb) {
    const int m = (mblk * C::MB + mb) * 32 + r;
    const int rem = m % (C::HO * C::HO), oh = rem / C::HO, ow = rem % C::HO;
    int R = 0;
#pragma unroll
    for (int kh = -1; kh <= 1; ++kh)
#pragma unroll
      for (int kw = -1; kw <= 1; ++kw) {
        const bool ok = (unsigned)(oh + kh) < (unsigned)C::HO && (unsigned)(ow + kw) < (unsigned)C::HO;
        R += ok ? tab[m + kh * C::HO + kw] : 0;
      }
    A.rsum[mb] = h ? 0 : R;
  }
}
template <int PIXB, int CCH>
struct EpiDenseTileResGlobal {
  uint8_t* xt; const uint8_t* res; int n_valid_px; QConv p; QAdd a;
  __device__ __forceinline__ int pixel(int m) const { return m * PIXB; }
  __device__ __forceinline__ uint32_t load_px(int m, int c0) const {
    return m < n_valid_px ? *reinterpret_cast<const uint32_t*>(res + (int64_t)m * CCH + c0) : 0u;
  }
  __device__ __forceinline__ uint32_t load(int, int) const { return 0u; }
  __device__ __forceinline__ void store(int po, int c0, float v0, float v1, float v2, float v3, uint32_t rq) const {
    const float vv[4] = {v0, v1, v2, v3};
    const float rf[4] = {(float)(rq & 0xffu), (float)((rq >> 8) & 0xffu), (float)((rq >> 16) & 0xffu), (float)(rq >> 24)};
    float t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float da = __builtin_fmaf(p.s_y, __builtin_rintf(med3f(vv[i], p.vlo, p.vhi)), p.dl_y);
      const float db = __builtin_fmaf(a.s_r, rf[i], a.nzs_r);
      t[i] = (da + db) * a.inv_s_o;
    }
    *reinterpret_cast<uint32_t*>(xt + po + c0) = pack_rne_u8(t[0], t[1], t[2], t[3], a.vhi);
  }
};

// NWV = 8: two waves per SIMD, 256 VGPRs each (MB x NB = 2 x 3 tiles per wave).  Measured alternatives, all slower:
// NWV = 4 (one wave per SIMD, 4 x 3 tiles in the 512-register file: -25 %, the epilogues read accumulators out of AGPRs
// and a lone wave hides no latency); NWV = 12 (4 x 1 tiles, 168 VGPRs: -12 %) and NWV = 16 (1 x 3 tiles, 128 VGPRs:
// -5 %), both of which spill the next item's input prefetch and so put its HBM latency back on the critical path; and two
// independent 4-wave workgroups per CU (4 images each, 9 KiB slabs) whose M and E phases drift apart on their own: equal
// time at 96 channels -- overlapping the phases is not what this kernel lacks.
// Round 2 re-tested that with a full ping-pong kernel (two 4-wave groups in anti-phase sharing ONE weight ring, the E group
// taking its epilogue in slices between the M group's slab barriers; bit-exact, no spills): 0.399 ms at 96 channels and
// 0.454 ms at 192 against 0.341 / 0.298 ms here.  The wall is accumulator capacity: the 8 waves' 48 accumulator tiles ARE the
// item (8 / 16 images); a group that drains its accumulators while the other multiplies halves the images per pass of the
// block's weights (162 / 663 KiB), and the L2 -> LDS weight stream (3.5 TB/s chip-wide here, 4.7 TB/s there) is what the M
// phase waits for.  More images per weight pass needs more accumulator registers, not more LDS.
template <class C, int NWV, int NM = 1>
__global__ __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(NWV / 4, NWV / 4)))
void block_chain_ald_kernel(const ArgsArr<ChainArgs<1>, NM> all) {
  const ChainArgs<1>& a = all.m[NM == 1 ? 0 : blockIdx.y];
  static_assert(C::CIN == C::COUT && C::CIN % 32 == 0, "wide identity BasicBlock");
  using DT = DenseTile<C>;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr int NTHR = 64 * NWV;
  uint8_t* xt = smem;                                                        // dense tile + zero line
  uint8_t* rbase = smem + DT::BYTES + C::PIXB;                               // two weight slabs
  static_assert((DT::BYTES + C::PIXB) % 16 == 0, "ring alignment");
  int rcur = 0;
  float* bias_lds = reinterpret_cast<float*>(rbase + 2 * C::SLAB_BYTES);     // [2][COUT]
  int* sx = reinterpret_cast<int*>(bias_lds + 2 * C::COUT);                  // channel sums of the X tile  [G * HO * HO]
  int* stab = sx + C::G * C::HO * C::HO;                                     // ... of the T tile
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar control flow and addresses
  const BlockParams& bp = a.blk[0];

  constexpr int IMG_PX = C::HO * C::HO;
  constexpr int CPP = C::CIN / 16;                                           // 16-byte chunks per pixel
  constexpr int NCH = C::G * IMG_PX * CPP;
  constexpr int PER_T = (NCH + NTHR - 1) / NTHR;
  const int groups = (a.B + C::G - 1) / C::G;
  const ItemWalk walk(a.n_samples * groups, blockIdx.x, gridDim.x);     // interleaved per XCD: a sample's weights stay in ONE L2
  const int count = walk.count;

  for (int i = tid; i < C::PIXB / 4; i += NTHR) reinterpret_cast<uint32_t*>(xt + DT::BYTES)[i] = 0u;
  for (int i = tid; i < 2 * C::G * C::HO * C::HO; i += NTHR) sx[i] = 0;
  load_bias<C::COUT, NTHR>(bias_lds, bp.a.bias, tid);
  load_bias<C::COUT, NTHR>(bias_lds + C::COUT, bp.b.bias, tid);
  if (count <= 0) return;
  __syncthreads();                                   // tables are zero before the first tile write adds into them
  auto dot16 = [](const v4i& c) {
    int d = __builtin_amdgcn_sdot4(c.x, 0x01010101, 0, false);
    d = __builtin_amdgcn_sdot4(c.y, 0x01010101, d, false);
    d = __builtin_amdgcn_sdot4(c.z, 0x01010101, d, false);
    return __builtin_amdgcn_sdot4(c.w, 0x01010101, d, false);
  };

  // an item's images are contiguous in HBM: chunk i of the item is byte 16 i of that block
  v4i pre[PER_T];
  auto fetch = [&](int item) {
    const int s = item / groups, img0 = (item - s * groups) * C::G;
    const uint8_t* xs = a.x + (int64_t)s * a.x_ss + (int64_t)img0 * IMG_PX * C::CIN;
    const int valid = (a.B - img0 < C::G ? a.B - img0 : C::G) * IMG_PX * CPP;
    int t = tid;
    asm volatile("" : "+v"(t));         // per-thread addresses are recomputed here, not hoisted out of the item loop (spills)
#pragma unroll
    for (int j = 0; j < PER_T; ++j) {
      const int i = t + j * NTHR;
      pre[j] = *reinterpret_cast<const v4i*>(xs + (i < valid ? (int64_t)i * 16 : 0));
    }
  };
  auto write_tile = [&](int item) {
    const int s = item / groups, img0 = (item - s * groups) * C::G;
    const int valid = (a.B - img0 < C::G ? a.B - img0 : C::G) * IMG_PX * CPP;
    const uint32_t z4 = (uint32_t)a.z_in * 0x01010101u;
    int t = tid;
    asm volatile("" : "+v"(t));
#pragma unroll
    for (int j = 0; j < PER_T; ++j) {
      const int i = t + j * NTHR;
      if (i < NCH) {
        const int px = i / CPP, within = i - px * CPP;
        const v4i v = pre[j];
        const v4i c = i < valid ? v4i{(int)sub_bytes(v.x, z4), (int)sub_bytes(v.y, z4), (int)sub_bytes(v.z, z4), (int)sub_bytes(v.w, z4)} : v4i{0, 0, 0, 0};
        *reinterpret_cast<v4i*>(xt + px * C::PIXB + within * 16) = c;
        __hip_atomic_fetch_add(&sx[px], dot16(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  };
  auto wbase = [&](const QConv& q, int item) { return q.w + (int64_t)(item / groups) * q.w_ss; };

  fetch(walk.item(0));
  write_tile(walk.item(0));
  dma_slab<C, NWV>(rbase, wbase(bp.a, walk.item(0)), 0, wave, lane);
  ConvAcc<C> A;
  QBNN_STAMP_DECL
  for (int it = 0; it < count; ++it) {
    QBNN_STAMP_START();
    const int item = walk.item(it);
    const int s = item / groups, img0 = (item - s * groups) * C::G;
    const bool more = it + 1 < count;
    const int next = more ? walk.item(it + 1) : item;
    // ---- stem.0: M over the X tile, then T over it
    conv_ring_mfma_dense<C, NWV>(xt, rbase, rcur, wbase(bp.a, item), A, wave, lane,
                            [&](uint8_t* dst) { dma_slab<C, NWV>(dst, wbase(bp.b, item), 0, wave, lane); });
    QBNN_STAMP_AT(0);
    lds_barrier();                                       // every wave has read its last X fragment
    QBNN_STAMP_AT(1);
    {
      // stem.0 epilogue: window sums from the X table; the T table collects the channel sums of what is written
      window_sum_from_table<C>(sx, A, wave, lane);
      EpiDenseTile<C::PIXB> epi{xt, bp.a, 0};
      auto flush = [&](int mb) {
        const int v = epi.csum + __shfl_xor(epi.csum, 32);
        epi.csum = 0;
        if (lane < 32) __hip_atomic_fetch_add(&stab[((wave / C::NBLKS) * C::MB + mb) * 32 + lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      };
      conv_epi_phase_with<C, decltype(epi)>(bias_lds, bp.a, epi, A, wave, lane, [&](int, int, int, int, int) { return 0u; },
                                            [&](int mb) { if (mb > 0) flush(mb - 1); });
      flush(C::MB - 1);
    }
    QBNN_STAMP_AT(2);
    // ---- stem.3: M over T; residual and next input are requested during the last slab
    const int valid_px = (a.B - img0 < C::G ? a.B - img0 : C::G) * IMG_PX;
    EpiDenseTileResGlobal<C::PIXB, C::COUT> epi_b{xt, a.x + (int64_t)s * a.x_ss + (int64_t)img0 * IMG_PX * C::COUT, valid_px, bp.b, bp.add};
    uint32_t resq[2][C::NB][4];
    auto load_res = [&](int mb) {
      const int mblk = wave / C::NBLKS, nblk = wave - mblk * C::NBLKS;
#pragma unroll
      for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          resq[mb & 1][nb][g4] = epi_b.load_px((mblk * C::MB + mb) * 32 + (lane & 31), (nblk * C::NB + nb) * 32 + 8 * g4 + 4 * (lane >> 5));
    };
    conv_ring_mfma_dense<C, NWV>(xt, rbase, rcur, wbase(bp.b, item), A, wave, lane,
                            [&](uint8_t* dst) { if (more) dma_slab<C, NWV>(dst, wbase(bp.a, next), 0, wave, lane); load_res(0); fetch(next); });
    QBNN_STAMP_AT(3);
    lds_barrier();
    QBNN_STAMP_AT(4);
    for (int i = tid; i < C::G * IMG_PX; i += NTHR) sx[i] = 0;          // X table: last read in the stem.0 epilogue; refilled by the tile write below
    window_sum_from_table<C>(stab, A, wave, lane);
    conv_epi_phase_with<C, decltype(epi_b)>(bias_lds + C::COUT, bp.b, epi_b, A, wave, lane,
                                            [&](int mb, int nb, int g4, int, int) { return resq[mb & 1][nb][g4]; },
                                            [&](int mb) { if (mb + 1 < C::MB) load_res(mb + 1); });
    QBNN_STAMP_AT(5);
    lds_barrier();
    QBNN_STAMP_AT(6);
    for (int i = tid; i < C::G * IMG_PX; i += NTHR) stab[i] = 0;        // T table: every wave has gathered from it
    // ---- per 16-byte chunk: tile -> quint8 register, next item's input -> the same tile bytes, register -> HBM (the
    //      item's output block is contiguous).  The next input is written unconditionally (the last item rewrites
    //      itself): a prefetch left unconsumed on one path makes the compiler guard later reuses with vmcnt(0).
    {
      const uint32_t z4o = (uint32_t)bp.add.z_o * 0x01010101u, z4i = (uint32_t)a.z_in * 0x01010101u;
      uint8_t* ys = a.y + (int64_t)s * a.y_ss + (int64_t)img0 * IMG_PX * C::COUT;
      const int valid = valid_px * CPP;
      const int nimg0 = (next - (next / groups) * groups) * C::G;
      const int nvalid = (a.B - nimg0 < C::G ? a.B - nimg0 : C::G) * IMG_PX * CPP;
      int t = tid;
      asm volatile("" : "+v"(t));
#pragma unroll
      for (int j = 0; j < PER_T; ++j) {
        const int i = t + j * NTHR;
        if (i < NCH) {
          const int px = i / CPP, within = i - px * CPP;
          v4i* cell = reinterpret_cast<v4i*>(xt + px * C::PIXB + within * 16);
          const v4i v = *cell, n = pre[j];
          const v4i c = i < nvalid ? v4i{(int)sub_bytes(n.x, z4i), (int)sub_bytes(n.y, z4i), (int)sub_bytes(n.z, z4i), (int)sub_bytes(n.w, z4i)} : v4i{0, 0, 0, 0};
          *cell = c;
          __hip_atomic_fetch_add(&sx[px], dot16(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (i < valid)
            *reinterpret_cast<v4i*>(ys + (int64_t)i * 16) = v4i{(int)add_bytes(v.x, z4o), (int)add_bytes(v.y, z4o), (int)add_bytes(v.z, z4o), (int)add_bytes(v.w, z4o)};
        }
      }
    }
    QBNN_STAMP_AT(7);
  }
#ifdef QBNN_STAMP
  if (a.dbg && (tid & 63) == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(a.dbg + wave * 8 + i, st_acc[i]);
#endif
}

template <class C, int NWV>
static int launch_block_chain_ald(const ChainArgs<1>& a, hipStream_t st) {
  constexpr int LDS = DenseTile<C>::BYTES + C::PIXB + 2 * C::SLAB_BYTES + 2 * C::COUT * 4 + 2 * C::G * C::HO * C::HO * 4;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static std::atomic<uint64_t> attr{0};
  if (int rc_attr = ensure_dyn_lds((const void*)block_chain_ald_kernel<C, NWV, 1>, attr, LDS)) return rc_attr;
  const int groups = (a.B + C::G - 1) / C::G;
  const int n_items = a.n_samples * groups;
  const int grid = n_items < 256 ? n_items : 256;
  ArgsArr<ChainArgs<1>, 1> one;
  one.m[0] = a;
  hipLaunchKernelGGL((block_chain_ald_kernel<C, NWV, 1>), dim3(grid), dim3(64 * NWV), LDS, st, one);
  return check_launch("qbnn_block_chain_i8_mc");
}

// grid of a fused multi-call launch: every call gets the same number of workgroups (<= its item count), 256 in total
static int fused_grid_x(int max_items, int n_calls) {
  const int per = 256 / n_calls > 0 ? 256 / n_calls : 1;
  return max_items < per ? (max_items > 0 ? max_items : 1) : per;
}

template <class C, int NWV>
static int launch_block_chain_ald_multi(const ChainArgs<1>* arr, int n, hipStream_t st) {
  constexpr int LDS = DenseTile<C>::BYTES + C::PIXB + 2 * C::SLAB_BYTES + 2 * C::COUT * 4 + 2 * C::G * C::HO * C::HO * 4;
  static std::atomic<uint64_t> attr{0};
  if (int rc_attr = ensure_dyn_lds((const void*)block_chain_ald_kernel<C, NWV, QBNN_FUSED_CALLS>, attr, LDS)) return rc_attr;
  ArgsArr<ChainArgs<1>, QBNN_FUSED_CALLS> all;
  memset(&all, 0, sizeof(all));                   // unused blocks: n_samples = 0 -> their workgroups (none launched) would exit at once
  int items = 0;
  for (int i = 0; i < n; ++i) { all.m[i] = arr[i]; const int it = arr[i].n_samples * ((arr[i].B + C::G - 1) / C::G); items = it > items ? it : items; }
  hipLaunchKernelGGL((block_chain_ald_kernel<C, NWV, QBNN_FUSED_CALLS>), dim3(fused_grid_x(items, n), n), dim3(64 * NWV), LDS, st, all);
  return check_launch("qbnn_block_chain_i8_multi");
}


template <class C, int NBLK, bool LDSW = true, bool STEM = false> constexpr int chain_ws_lds() {
  return 2 * (C::G * C::TILE_BYTES + C::TILE_SLACK) + (LDSW ? 2 * NBLK * WConv<C>::BYTES : 0) + NBLK * 2 * C::COUT * 4 +
         (STEM ? 32 * 32 * 32 + 1024 + 24 * 4 : 0);
}

template <class C, int NBLK, bool LDSW = true, bool STEM = false>
static int launch_block_chain_ws(const ChainArgs<NBLK>& a, hipStream_t st) {
  constexpr int LDS = chain_ws_lds<C, NBLK, LDSW, STEM>();
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static std::atomic<uint64_t> attr{0};
  if (int rc_attr = ensure_dyn_lds((const void*)block_chain_ws_kernel<C, NBLK, LDSW, STEM, 1>, attr, LDS)) return rc_attr;
  const int groups = (a.B + C::G - 1) / C::G;
  const int n_items = a.n_samples * groups;
  const int grid = n_items < 256 ? n_items : 256;
  ArgsArr<ChainArgs<NBLK>, 1> one;
  one.m[0] = a;
  hipLaunchKernelGGL((block_chain_ws_kernel<C, NBLK, LDSW, STEM, 1>), dim3(grid), dim3(BLK_THREADS), LDS, st, one);
  return check_launch("qbnn_block_chain_i8_mc");
}

template <class C, int NBLK, bool STEM, int NM>
static int launch_block_chain_ws_multi(const ChainArgs<NBLK>* arr, int n, hipStream_t st) {
  constexpr int LDS = chain_ws_lds<C, NBLK, true, STEM>();
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static_assert(sizeof(ArgsArr<ChainArgs<NBLK>, NM>) <= 3840, "kernel arguments are limited to 4 KiB (incl. the hidden ones)");
  static std::atomic<uint64_t> attr{0};
  if (int rc_attr = ensure_dyn_lds((const void*)block_chain_ws_kernel<C, NBLK, true, STEM, NM>, attr, LDS)) return rc_attr;
  ArgsArr<ChainArgs<NBLK>, NM> all;
  memset(&all, 0, sizeof(all));
  int items = 0;
  for (int i = 0; i < n; ++i) { all.m[i] = arr[i]; const int it = arr[i].n_samples * ((arr[i].B + C::G - 1) / C::G); items = it > items ? it : items; }
  hipLaunchKernelGGL((block_chain_ws_kernel<C, NBLK, true, STEM, NM>), dim3(fused_grid_x(items, n), n), dim3(BLK_THREADS), LDS, st, all);
  return check_launch("qbnn_block_chain_i8_multi");
}

template <class CB> struct DownSC {
  static constexpr int PITCH = CB::COUT + 8;
  static constexpr int BYTES = (CB::M * PITCH + 15) / 16 * 16;
};

template <class CA, class CS, class CB, bool LDSW, int NM = 1>
__global__ __launch_bounds__(BLK_THREADS) void block_down_ws_kernel(const ArgsArr<DownArgs, NM> all) {
  const DownArgs& a = all.m[NM == 1 ? 0 : blockIdx.y];
  static_assert(CA::M == CS::M && CA::M == CB::M && CA::G == CS::G && CA::G == CB::G, "one work item, three convs");
  static_assert(CA::COUT == CB::CIN && CA::COUT == CB::COUT && CS::COUT == CB::COUT && CA::HO == CB::HIN, "block geometry");
  static_assert(CA::TILE_BYTES == CS::TILE_BYTES && CA::CIN == CS::CIN && CA::HIN == CS::HIN, "shared input tile");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr int XB = CA::G * CA::TILE_BYTES + CA::TILE_SLACK;
  constexpr int TB = CB::G * CB::TILE_BYTES + CB::TILE_SLACK;
  constexpr int COUT = CB::COUT;
  // SC: the block's shortcut / output staging buffer, quint8 [M][COUT] with the pixel pitch padded by 8 bytes: the
  // epilogues touch it with one dword per lane at 32 consecutive pixels, and a pitch of 48 / 96 / 192 bytes is a
  // 4- / 8- / 16-way bank conflict (32 banks for 4-byte accesses); 56 / 104 / 200 are 2-way, which is free.
  constexpr int SCP = DownSC<CB>::PITCH, SC_BYTES = DownSC<CB>::BYTES;
  uint8_t* xt = smem;
  uint8_t* tt = smem + XB;
  uint8_t* sc = tt + TB;
  uint8_t* wl_s = sc + SC_BYTES;
  uint8_t* wl_a = wl_s + (LDSW ? WConv<CS>::BYTES : 0);
  uint8_t* wl_b = wl_a + (LDSW ? WConv<CA>::BYTES : 0);
  float* bias_lds = reinterpret_cast<float*>(wl_b + (LDSW ? WConv<CB>::BYTES : 0));       // [3][COUT]: s, a, b
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar control flow and addresses

  constexpr int CPR = CA::ROWB / 16, CPI = CA::HIN * CPR, NCH = CA::G * CPI;
  constexpr int PER_T = (NCH + BLK_THREADS - 1) / BLK_THREADS;
  const int groups = (a.B + CA::G - 1) / CA::G;
  int begin = 0, count;
  const ItemWalk walk(a.n_samples * groups, blockIdx.x, gridDim.x);
  if (LDSW) item_range(a.n_samples * groups, blockIdx.x, gridDim.x, begin, count);
  else count = walk.count;
  auto item_at = [&](int it) { return LDSW ? begin + it : walk.item(it); };

  zero_halo<CA::TW, CA::PIXB, CA::TILE_BYTES, CA::G, BLK_THREADS>(xt, tid);
  zero_halo<CB::TW, CB::PIXB, CB::TILE_BYTES, CB::G, BLK_THREADS>(tt, tid);
  load_bias<COUT, BLK_THREADS>(bias_lds, a.s.bias, tid);
  load_bias<COUT, BLK_THREADS>(bias_lds + COUT, a.a.bias, tid);
  load_bias<COUT, BLK_THREADS>(bias_lds + 2 * COUT, a.b.bias, tid);

  v4i pre[PER_T];
  // (the thread's chunk offsets are recomputed per call from an opaque copy of tid: kept in registers across the item loop they are
  //  what spills at 48 -> 96 channels, and a spill reload is a vmcnt wait -- at the loop top it waited for the previous item's stores)
  auto fetch = [&](int item) {
    const int s = item / groups, img0 = (item - s * groups) * CA::G;
    const uint8_t* xs = a.x + (int64_t)s * a.x_ss;
    int t_ = tid;
    if constexpr (!LDSW) asm volatile("" : "+v"(t_));      // (the weights-stationary 24 -> 48 block has registers to spare and is faster without)
#pragma unroll
    for (int j = 0; j < PER_T; ++j) {
      const int i = t_ + j * BLK_THREADS;
      const int g = i / CPI, rem = i - g * CPI;
      const bool ok = (i < NCH) && (img0 + g < a.B);
      const int64_t off = ok ? ((int64_t)(img0 + g) * CA::HIN) * CA::ROWB + (int64_t)rem * 16 : 0;
      pre[j] = *reinterpret_cast<const v4i*>(xs + off);
    }
  };
  // the X tile is free from the barrier that follows conv_a on
  auto write_tile = [&](int item) {
    const int s = item / groups, img0 = (item - s * groups) * CA::G;
    const uint32_t z4 = (uint32_t)a.z_in * 0x01010101u;
    int t_ = tid;
    if constexpr (!LDSW) asm volatile("" : "+v"(t_));      // (the weights-stationary 24 -> 48 block has registers to spare and is faster without)
#pragma unroll
    for (int j = 0; j < PER_T; ++j) {
      const int i = t_ + j * BLK_THREADS;
      if (i < NCH) {
        const int g = i / CPI, rem = i - g * CPI, row = rem / CPR, within = rem - row * CPR;
        const bool ok = img0 + g < a.B;
        const v4i v = pre[j];
        uint8_t* d = xt + g * CA::TILE_BYTES + (row + 1) * CA::PITCH + CA::row_chunk_off(within);
        *reinterpret_cast<v2i*>(d) = ok ? v2i{(int)sub_bytes(v.x, z4), (int)sub_bytes(v.y, z4)} : v2i{0, 0};
        *reinterpret_cast<v2i*>(d + 8) = ok ? v2i{(int)sub_bytes(v.z, z4), (int)sub_bytes(v.w, z4)} : v2i{0, 0};
      }
    }
  };
  if (count <= 0) return;
  fetch(item_at(0));
  write_tile(item_at(0));
  int cur_s = -1;
  QBNN_STAMP_DECL
  for (int it = 0; it < count; ++it) {
    QBNN_STAMP_START();
    const int item = item_at(it);
    const int s = item / groups, img0 = (item - s * groups) * CA::G;
    const bool more = it + 1 < count;
    fetch(more ? item_at(it + 1) : item);    // unconditional: exact wait counts at its use (see block_chain_ws_kernel)
    if (LDSW && s != cur_s) {
      __syncthreads();
      dma_conv<CS, BLK_WAVES>(wl_s, a.s.w + (int64_t)s * a.s.w_ss, wave, lane);
      dma_conv<CA, BLK_WAVES>(wl_a, a.a.w + (int64_t)s * a.a.w_ss, wave, lane);
      dma_conv<CB, BLK_WAVES>(wl_b, a.b.w + (int64_t)s * a.b.w_ss, wave, lane);
      dma_barrier();
      cur_s = s;
    }
    QBNN_STAMP_AT(0);
    lds_barrier();       // X complete; the previous item's SC has been read out by every thread
    QBNN_STAMP_AT(1);
    {
      EpiDense<COUT, false, SCP> epi{sc, a.s, a.add};
      if constexpr (LDSW) conv_core<CS, decltype(epi), BLK_WAVES>(xt, wl_s, bias_lds, a.s, epi, wave, lane);
      else conv_passes<CS, decltype(epi), BLK_WAVES>(xt, a.s.w + (int64_t)s * a.s.w_ss, bias_lds, a.s, epi, wave, lane);
    }
    {
      EpiTile<CB::HIN, CB::PIXB, CB::TILE_BYTES> epi{tt, a.a};
      if constexpr (LDSW) conv_core<CA, decltype(epi), BLK_WAVES>(xt, wl_a, bias_lds + COUT, a.a, epi, wave, lane);
      else conv_passes<CA, decltype(epi), BLK_WAVES>(xt, a.a.w + (int64_t)s * a.a.w_ss, bias_lds + COUT, a.a, epi, wave, lane);
    }
    QBNN_STAMP_AT(2);
    lds_barrier();       // T and SC complete
    QBNN_STAMP_AT(3);
    {
      EpiDense<COUT, true, SCP> epi{sc, a.b, a.add};
      if constexpr (LDSW) conv_core<CB, decltype(epi), BLK_WAVES>(tt, wl_b, bias_lds + 2 * COUT, a.b, epi, wave, lane);
      else conv_passes<CB, decltype(epi), BLK_WAVES>(tt, a.b.w + (int64_t)s * a.b.w_ss, bias_lds + 2 * COUT, a.b, epi, wave, lane);
    }
    QBNN_STAMP_AT(4);
    lds_barrier();
    QBNN_STAMP_AT(5);
    // read-out of the finished block output.  Weights-stationary form (registers to spare): all LDS reads first (a rolled
    // read -> wait -> store loop pays the LDS latency per trip), then the next X tile, then the stores -- nothing in the
    // next item waits on them.  The streaming forms sit at the register limit and keep the rolled loop.
    constexpr int IMG_OUT = CB::HO * CB::HO * COUT, U8 = COUT / 8;          // 8-byte units (the padded pitch is 8-aligned)
    constexpr int NOUT = (CB::M * U8 + BLK_THREADS - 1) / BLK_THREADS;
    uint8_t* ys = a.y + (int64_t)s * a.y_ss + (int64_t)img0 * IMG_OUT;
    if constexpr (LDSW) {
      v2i outv[NOUT];
#pragma unroll
      for (int j = 0; j < NOUT; ++j) {
        const int i = tid + j * BLK_THREADS;
        const int px = i / U8, within = i - px * U8;
        if (i < CB::M * U8) outv[j] = *reinterpret_cast<const v2i*>(sc + px * SCP + within * 8);
      }
      if (more) write_tile(item_at(it + 1));      // before the stores: its vmcnt wait then covers only the (old) input loads
      QBNN_STAMP_AT(6);
#pragma unroll
      for (int j = 0; j < NOUT; ++j) {
        const int i = tid + j * BLK_THREADS;
        if (i < CB::M * U8 && img0 + (i * 8) / IMG_OUT < a.B) *reinterpret_cast<v2i*>(ys + (int64_t)i * 8) = outv[j];
      }
    } else {
      if (more) write_tile(item_at(it + 1));
      QBNN_STAMP_AT(6);
      for (int i = tid; i < CB::M * U8; i += BLK_THREADS)
        if (img0 + (i * 8) / IMG_OUT < a.B) {
          const int px = i / U8, within = i - px * U8;
          *reinterpret_cast<v2i*>(ys + (int64_t)i * 8) = *reinterpret_cast<const v2i*>(sc + px * SCP + within * 8);
        }
    }
    QBNN_STAMP_AT(7);
  }
#ifdef QBNN_STAMP
  if (g_stamp_dev && (tid & 63) == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(g_stamp_dev + wave * 8 + i, st_acc[i]);
#endif
}

template <class CA, class CS, class CB, bool LDSW>
static int launch_block_down_ws(const DownArgs& a, hipStream_t st) {
  constexpr int LDS = CA::G * CA::TILE_BYTES + CA::TILE_SLACK + CB::G * CB::TILE_BYTES + CB::TILE_SLACK + DownSC<CB>::BYTES +
                      (LDSW ? WConv<CS>::BYTES + WConv<CA>::BYTES + WConv<CB>::BYTES : 0) + 3 * CB::COUT * 4;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static std::atomic<uint64_t> attr{0};
  if (int rc_attr = ensure_dyn_lds((const void*)block_down_ws_kernel<CA, CS, CB, LDSW, 1>, attr, LDS)) return rc_attr;
  const int groups = (a.B + CA::G - 1) / CA::G;
  const int n_items = a.n_samples * groups;
  const int grid = n_items < 256 ? n_items : 256;
  ArgsArr<DownArgs, 1> one;
  one.m[0] = a;
  hipLaunchKernelGGL((block_down_ws_kernel<CA, CS, CB, LDSW, 1>), dim3(grid), dim3(BLK_THREADS), LDS, st, one);
  return check_launch("qbnn_block_down_i8_mc");
}

template <class CA, class CS, class CB, bool LDSW>
static int launch_block_down_ws_multi(const DownArgs* arr, int n, hipStream_t st) {
  constexpr int LDS = CA::G * CA::TILE_BYTES + CA::TILE_SLACK + CB::G * CB::TILE_BYTES + CB::TILE_SLACK + DownSC<CB>::BYTES +
                      (LDSW ? WConv<CS>::BYTES + WConv<CA>::BYTES + WConv<CB>::BYTES : 0) + 3 * CB::COUT * 4;
  static_assert(sizeof(ArgsArr<DownArgs, QBNN_FUSED_CALLS>) <= 3840, "kernel arguments are limited to 4 KiB (incl. the hidden ones)");
  static std::atomic<uint64_t> attr{0};
  if (int rc_attr = ensure_dyn_lds((const void*)block_down_ws_kernel<CA, CS, CB, LDSW, QBNN_FUSED_CALLS>, attr, LDS)) return rc_attr;
  ArgsArr<DownArgs, QBNN_FUSED_CALLS> all;
  memset(&all, 0, sizeof(all));
  int items = 0;
  for (int i = 0; i < n; ++i) { all.m[i] = arr[i]; const int it = arr[i].n_samples * ((arr[i].B + CA::G - 1) / CA::G); items = it > items ? it : items; }
  hipLaunchKernelGGL((block_down_ws_kernel<CA, CS, CB, LDSW, QBNN_FUSED_CALLS>), dim3(fused_grid_x(items, n), n), dim3(BLK_THREADS), LDS, st, all);
  return check_launch("qbnn_block_down_i8_multi");
}

//                          CIN COUT K  S  HIN HALO G  MB NB
using Blk_24  = ConvCfg<24, 24, 3, 1, 32, 1, 1, 4, 1>;
using Blk_48  = ConvCfg<48, 48, 3, 1, 16, 1, 2, 2, 2>;
using ALD_96  = ConvCfg<96, 96, 3, 1, 8, 1, 8, 2, 3, true, 36, 16>;      // dense aliased-tile ring kernel
using ALD_192 = ConvCfg<192, 192, 3, 1, 4, 1, 16, 2, 3, true, 36, 16>;
using PP_48   = ConvCfg<48, 48, 3, 1, 16, 1, 1, 2, 2>;          // per wave group of the ping-pong kernel

template <int NBLK>
static int build_chain_args(ChainArgs<NBLK>& a, const uint8_t* x, int64_t x_ss, float s_x, int32_t z_x, int32_t B, int32_t a_hi,
                            const qbnn_block_desc* blk, uint8_t* y, int64_t y_ss, int32_t n_samples, const int8_t* stem_x, const QConv* stem) {
  memset(&a, 0, sizeof(a));
  if (stem) { a.stem_x = stem_x; a.stem = *stem; }
  a.x = x; a.x_ss = x_ss; a.y = y; a.y_ss = y_ss; a.B = B; a.n_samples = n_samples; a.z_in = z_x;
#ifdef QBNN_STAMP
  a.dbg = g_stamp_buf;
#endif
  float s_in = s_x; int z_in = z_x;
  for (int k = 0; k < NBLK; ++k) {
    const qbnn_block_desc& b = blk[k];
    qbnn_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.a_hi = a_hi;
    d.s_x = s_in; d.z_x = z_in; d.s_w = b.s_wa; d.z_w = b.z_wa; d.s_y = b.s_a; d.z_y = b.z_a; d.relu = 1; d.has_bias = b.bias_a != nullptr;
    int rc = fill_qconv(a.blk[k].a, b.w_a, b.w_a_sample_stride, b.bias_a, &d);
    if (rc) return rc;
    d.s_x = b.s_a; d.z_x = b.z_a; d.s_w = b.s_wb; d.z_w = b.z_wb; d.s_y = b.s_b; d.z_y = b.z_b; d.relu = 0; d.has_bias = b.bias_b != nullptr;
    if ((rc = fill_qconv(a.blk[k].b, b.w_b, b.w_b_sample_stride, b.bias_b, &d))) return rc;
    d.s_r = s_in; d.z_r = z_in; d.s_o = b.s_o; d.z_o = b.z_o;
    if ((rc = fill_qadd(a.blk[k].add, &d))) return rc;
    s_in = b.s_o; z_in = b.z_o;
  }
  return QBNN_OK;
}

static int build_stem_qconv(QConv& stem, const int8_t* w0_packed, int64_t w0_ss, const float* bias0, float s_x, float s_w0, int32_t z_w0,
                            float s_y0, int32_t z_y0, int32_t a_hi) {
  qbnn_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.a_hi = a_hi; d.s_x = s_x; d.z_x = 0; d.s_w = s_w0; d.z_w = z_w0; d.s_y = s_y0; d.z_y = z_y0; d.relu = 1; d.has_bias = bias0 != nullptr;
  memset(&stem, 0, sizeof(stem));
  return fill_qconv(stem, w0_packed, w0_ss, bias0, &d);
}

template <int NBLK>
static int block_chain_dispatch(const uint8_t* x, int64_t x_ss, float s_x, int32_t z_x, int32_t B, int32_t H, int32_t Cc,
                                int32_t a_hi, const qbnn_block_desc* blk, uint8_t* y, int64_t y_ss, int32_t n_samples,
                                hipStream_t st, const int8_t* stem_x = nullptr, const QConv* stem = nullptr) {
  ChainArgs<NBLK> a;
  if (int rc = build_chain_args<NBLK>(a, x, x_ss, s_x, z_x, B, a_hi, blk, y, y_ss, n_samples, stem_x, stem)) return rc;
  if (stem) {
    if (Cc != 24 || H != 32) return fail(QBNN_E_INVALID, "qbnn_stem_chain_i8_mc: the fused stem feeds the 32x32x24 chain only%s");
    return launch_block_chain_ws<Blk_24, NBLK, true, true>(a, st);
  }
  if (Cc == 24 && H == 32) return launch_block_chain_ws<Blk_24, NBLK>(a, st);
  if (Cc == 48 && H == 16) {
    if constexpr (chain_pp_lds<PP_48, NBLK>() <= 160 * 1024) {
      if (!no_pingpong() && ((B + PP_48::G - 1) / PP_48::G) % 2 == 0) return launch_block_chain_pp<PP_48, NBLK>(a, st);
    }
    if constexpr (chain_ws_lds<Blk_48, NBLK>() <= 160 * 1024) return launch_block_chain_ws<Blk_48, NBLK>(a, st);
    else return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_mc: one 48-channel block per launch for this batch size%s");
  }
  if (Cc == 96 && H == 8) {
    if constexpr (NBLK == 1) return launch_block_chain_ald<ALD_96, 8>(a, st);
    else return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_mc: one block per launch at 96 channels (its weights stream through the LDS ring)%s");
  }
  if (Cc == 192 && H == 4) {
    if constexpr (NBLK == 1) return launch_block_chain_ald<ALD_192, 8>(a, st);
    else return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_mc: one block per launch at 192 channels (its weights stream through the LDS ring)%s");
  }
  return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_mc: unsupported geometry%s C=%ld H=%ld", "", Cc, H);
}

QBNN_EXPORT int qbnn_block_chain_i8_mc(const uint8_t* x, int64_t x_ss, float s_x, int32_t z_x, int32_t B, int32_t H, int32_t Cc,
                                       int32_t a_hi, const qbnn_block_desc* host_blocks, int32_t n_blocks, uint8_t* y,
                                       int64_t y_ss, int32_t n_samples, void* stream) {
  if (!x || !y || !host_blocks || n_samples <= 0 || B <= 0) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_mc: bad argument%s");
  for (int k = 0; k < n_blocks; ++k)
    if (!host_blocks[k].w_a || !host_blocks[k].w_b) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_mc: NULL weights%s");
  hipStream_t st = (hipStream_t)stream;
  if (n_blocks == 1) return block_chain_dispatch<1>(x, x_ss, s_x, z_x, B, H, Cc, a_hi, host_blocks, y, y_ss, n_samples, st);
  if (n_blocks == 2) return block_chain_dispatch<2>(x, x_ss, s_x, z_x, B, H, Cc, a_hi, host_blocks, y, y_ss, n_samples, st);
  return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_mc: 1 or 2 blocks per launch%s");
}

QBNN_EXPORT int qbnn_stem_chain_i8_mc(const int8_t* im2col, int32_t B, const int8_t* w0_packed, int64_t w0_ss, const float* bias0,
                                      float s_x, float s_w0, int32_t z_w0, float s_y0, int32_t z_y0, int32_t a_hi,
                                      const qbnn_block_desc* host_blocks, int32_t n_blocks, uint8_t* y, int64_t y_ss,
                                      int32_t n_samples, void* stream) {
  if (!im2col || !w0_packed || !y || !host_blocks || n_samples <= 0 || B <= 0) return fail(QBNN_E_INVALID, "qbnn_stem_chain_i8_mc: bad argument%s");
  for (int k = 0; k < n_blocks; ++k)
    if (!host_blocks[k].w_a || !host_blocks[k].w_b) return fail(QBNN_E_INVALID, "qbnn_stem_chain_i8_mc: NULL weights%s");
  QConv stem;
  if (int rc = build_stem_qconv(stem, w0_packed, w0_ss, bias0, s_x, s_w0, z_w0, s_y0, z_y0, a_hi)) return rc;
  hipStream_t st = (hipStream_t)stream;
  // the chain's input is conv0's output: scale s_y0, zero point z_y0
  if (n_blocks == 1) return block_chain_dispatch<1>(nullptr, 0, s_y0, z_y0, B, 32, 24, a_hi, host_blocks, y, y_ss, n_samples, st, im2col, &stem);
  if (n_blocks == 2) return block_chain_dispatch<2>(nullptr, 0, s_y0, z_y0, B, 32, 24, a_hi, host_blocks, y, y_ss, n_samples, st, im2col, &stem);
  return fail(QBNN_E_INVALID, "qbnn_stem_chain_i8_mc: 1 or 2 blocks per launch%s");
}

// ---- fused multi-call launches (ensemble members): see ArgsArr ------------------------------------------------------------
QBNN_EXPORT int qbnn_block_chain_i8_multi(const qbnn_chain_call* calls, int32_t n_calls, int32_t with_stem, int32_t B, int32_t H,
                                          int32_t Cc, int32_t a_hi, int32_t n_blocks, void* stream) {
  if (!calls || n_calls <= 0 || B <= 0) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi: bad argument%s");
  hipStream_t st = (hipStream_t)stream;
  constexpr int NM2 = 4;                              // ChainArgs<2> with the stem: 4 argument blocks fit the 4 KiB of kernel arguments
  for (int c0 = 0; c0 < n_calls;) {
    const int lim = (with_stem || n_blocks == 2) ? NM2 : QBNN_FUSED_CALLS;
    const int n = n_calls - c0 < lim ? n_calls - c0 : lim;
    int rc = QBNN_OK;
    if (with_stem) {
      if (n_blocks != 2 || Cc != 24 || H != 32) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi: the fused stem feeds the two 32x32x24 blocks only%s");
      ChainArgs<2> arr[NM2];
      for (int i = 0; i < n; ++i) {
        const qbnn_chain_call& k = calls[c0 + i];
        if (!k.im2col || !k.w0_packed || !k.blocks || !k.y || k.n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi: bad call entry%s");
        QConv stem;
        if ((rc = build_stem_qconv(stem, k.w0_packed, k.w0_sample_stride, k.bias0, k.s_in, k.s_w0, k.z_w0, k.s_y0, k.z_y0, a_hi))) return rc;
        if ((rc = build_chain_args<2>(arr[i], nullptr, 0, k.s_y0, k.z_y0, B, a_hi, k.blocks, k.y, k.y_sample_stride, k.n_samples, k.im2col, &stem))) return rc;
      }
      rc = launch_block_chain_ws_multi<Blk_24, 2, true, NM2>(arr, n, st);
    } else {
      if (n_blocks != 1) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi: one block per call (two only behind the fused stem)%s");
      ChainArgs<1> arr[QBNN_FUSED_CALLS];
      for (int i = 0; i < n; ++i) {
        const qbnn_chain_call& k = calls[c0 + i];
        if (!k.x || !k.blocks || !k.y || k.n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi: bad call entry%s");
        if ((rc = build_chain_args<1>(arr[i], k.x, k.x_sample_stride, k.s_x, k.z_x, B, a_hi, k.blocks, k.y, k.y_sample_stride, k.n_samples, nullptr, nullptr))) return rc;
      }
      if (Cc == 48 && H == 16) rc = launch_block_chain_ws_multi<Blk_48, 1, false, QBNN_FUSED_CALLS>(arr, n, st);
      else if (Cc == 96 && H == 8) rc = launch_block_chain_ald_multi<ALD_96, 8>(arr, n, st);
      else if (Cc == 192 && H == 4) rc = launch_block_chain_ald_multi<ALD_192, 8>(arr, n, st);
      else return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi: unsupported geometry%s C=%ld H=%ld", "", Cc, H);
    }
    if (rc) return rc;
    c0 += n;
  }
  return QBNN_OK;
}

// Stand-alone quantized::add (+ clamp_activation, ReLU, clamp_activation) for graphs where something sits between the
// last conv of a block and its Add (MC-Dropout ResNet: mcdropout/models_mc.py:136-160).  Four elements per thread.
__global__ __launch_bounds__(256) void add_relu_q_kernel(const uint8_t* __restrict__ a, int64_t a_ss, const uint8_t* __restrict__ b,
                                                          int64_t b_ss, uint8_t* __restrict__ y, int64_t y_ss, int64_t n4, float s_a,
                                                          float nzs_a, float s_b, float nzs_b, float inv_s_o, int z_o, int a_hi, int relu) {
  const int s = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const uint32_t av = reinterpret_cast<const uint32_t*>(a + (int64_t)s * a_ss)[i];
    const uint32_t bv = reinterpret_cast<const uint32_t*>(b + (int64_t)s * b_ss)[i];
    uint32_t o = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o |= add_relu_one((av >> (8 * j)) & 0xffu, (bv >> (8 * j)) & 0xffu, s_a, nzs_a, s_b, nzs_b, inv_s_o, z_o, a_hi, relu) << (8 * j);
    }
    reinterpret_cast<uint32_t*>(y + (int64_t)s * y_ss)[i] = o;
  }
}

QBNN_EXPORT int qbnn_add_relu_q_mc(const uint8_t* a, int64_t a_ss, float s_a, int32_t z_a, const uint8_t* b, int64_t b_ss, float s_b,
                                   int32_t z_b, uint8_t* y, int64_t y_ss, int64_t n, float s_o, int32_t z_o, int32_t a_hi, int32_t relu,
                                   int32_t n_samples, void* stream) {
  if (!a || !b || !y || n <= 0 || (n & 3) || n_samples <= 0 || (a_ss & 3) || (b_ss & 3) || (y_ss & 3))
    return fail(QBNN_E_INVALID, "qbnn_add_relu_q_mc: bad argument (element counts and strides must be multiples of 4)%s");
  const int64_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(add_relu_q_kernel, dim3(blocks, n_samples), dim3(256), 0, (hipStream_t)stream, a, a_ss, b, b_ss, y, y_ss, n4, s_a,
                     (float)(-z_a) * s_a, s_b, (float)(-z_b) * s_b, 1.0f / s_o, z_o, a_hi, relu);
  return check_launch("qbnn_add_relu_q_mc");
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

// QuantStub + clamp_activation + the layer-0 patch gather for SEVERAL input quantisations at once (ensemble members each own
// a `quant.scale / zero_point`): fp32 NCHW [B][3][H][W] -> centred int8 patches out[m][B][H*W][32], member m = blockIdx.y.
struct QuantIm2colArgs { float inv[16]; int z[16]; };
__global__ __launch_bounds__(256) void quantize_im2col3x3_c3_kernel(const float* __restrict__ x, int B, int H, int W, const QuantIm2colArgs q,
                                                                     int a_hi, int8_t* __restrict__ out, int64_t out_stride) {
  const int m = blockIdx.y;
  const float inv = q.inv[m];
  const int z = q.z[m];
  const int64_t npix = (int64_t)B * H * W, plane = (int64_t)H * W;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
    const int ow = (int)(p % W);
    const int oh = (int)((p / W) % H);
    const int64_t b = p / plane;
    uint32_t wds[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int8_t* by = reinterpret_cast<int8_t*>(wds);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = oh + kh - 1, iw = ow + kw - 1;
        const bool in = ih >= 0 && ih < H && iw >= 0 && iw < W;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          int v = 0;
          if (in) {
            const float f = x[((b * 3 + c) * H + ih) * W + iw];
            v = min(min(max(z + rne_sat(f * inv), 0), 255), a_hi) - z;          // quantize_input_kernel, then im2col3x3_c3_kernel's centring
          }
          by[(kh * 3 + kw) * 3 + c] = (int8_t)v;
        }
      }
    v4i* o = reinterpret_cast<v4i*>(out + (int64_t)m * out_stride + p * 32);
    o[0] = v4i{(int)wds[0], (int)wds[1], (int)wds[2], (int)wds[3]};
    o[1] = v4i{(int)wds[4], (int)wds[5], (int)wds[6], (int)wds[7]};
  }
}

QBNN_EXPORT int qbnn_quantize_im2col3x3_c3_multi(const float* x, int32_t B, int32_t H, int32_t W, const float* scales, const int32_t* zero_points,
                                                 int32_t n, int32_t a_hi, int8_t* out, int64_t out_stride, void* stream) {
  if (!x || !scales || !zero_points || !out || B <= 0 || n <= 0) return fail(QBNN_E_INVALID, "qbnn_quantize_im2col3x3_c3_multi: bad argument%s");
  const int64_t npix = (int64_t)B * H * W;
  const int blocks = (int)((npix + 255) / 256 < 1024 ? (npix + 255) / 256 : 1024);
  for (int c0 = 0; c0 < n; c0 += 16) {
    const int k = n - c0 < 16 ? n - c0 : 16;
    QuantIm2colArgs q;
    memset(&q, 0, sizeof(q));
    for (int i = 0; i < k; ++i) {
      if (zero_points[c0 + i] < 0 || zero_points[c0 + i] > 127) return fail(QBNN_E_INVALID, "qbnn_quantize_im2col3x3_c3_multi: zero points must be in [0,127]%s");
      q.inv[i] = 1.0f / scales[c0 + i]; q.z[i] = zero_points[c0 + i];
    }
    hipLaunchKernelGGL(quantize_im2col3x3_c3_kernel, dim3(blocks, k), dim3(256), 0, (hipStream_t)stream, x, B, H, W, q, a_hi,
                       out + (int64_t)c0 * out_stride, out_stride);
    if (int rc = check_launch("qbnn_quantize_im2col3x3_c3_multi")) return rc;
  }
  return QBNN_OK;
}

// head: one wave per (sample, image).  C <= 256 channels, N <= 16 classes (4 lanes per class).
struct HeadArgs {
  const uint8_t* x; int64_t x_ss;
  const int8_t* w; int64_t w_ss;
  const float* bias;
  float* probs;
  int B, kk, C, N;
  int z_x, z_w, z_y, a_hi;
  float inv_kk, rcp, mult, s_y;
};

#define QBNN_HEAD_IMGS 1            // images per wave: the sample's Linear weights are loaded once for all of them
template <int NM = 1>
__global__ __launch_bounds__(256) void head_i8_kernel(const ArgsArr<HeadArgs, NM> all) {
  const HeadArgs& a = all.m[NM == 1 ? 0 : blockIdx.z];
  __shared__ int pooled[4][256];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = blockIdx.y;
  const int b0 = (blockIdx.x * 4 + wave) * QBNN_HEAD_IMGS;
  if (b0 >= a.B) return;
  const int8_t* ws = a.w + (int64_t)s * a.w_ss;
  const int n = lane >> 2, j = lane & 3;
  // packed form: C a multiple of 16 and dword-aligned weights -- pooled activations as int8 dwords, v_dot4 against the raw weight
  // dwords (kept in registers for the wave's images), the weights' zero point through the channel sum:  sum p (w - z_w) = p.w - z_w sum p
  const bool packed = (a.C & 15) == 0 && a.C <= 256 && ((reinterpret_cast<uintptr_t>(ws) | (uintptr_t)a.w_ss) & 3) == 0;
  for (int bi = 0; bi < QBNN_HEAD_IMGS; ++bi) {
    const int b = b0 + bi;
    if (b >= a.B) break;
    const uint8_t* xs = a.x + (int64_t)s * a.x_ss + (int64_t)b * a.kk * a.C;
    // AvgPool2d(k) on quint8, channels-last: q = clamp(rne((sum - kk z) / kk) + z, 0, 255); then clamp_activation.
    // Four channels per lane (one dword per pixel) when C is a multiple of 4.
    if ((a.C & 3) == 0) {
      for (int c4 = lane; c4 < a.C / 4; c4 += 64) {
        int sum[4] = {0, 0, 0, 0};
        for (int p = 0; p < a.kk; ++p) {
          const uint32_t v = *reinterpret_cast<const uint32_t*>(xs + p * a.C + 4 * c4);
          sum[0] += v & 0xffu; sum[1] += (v >> 8) & 0xffu; sum[2] += (v >> 16) & 0xffu; sum[3] += v >> 24;
        }
        uint32_t pk = 0;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          int q = min(max(rne_sat((float)(sum[jj] - a.kk * a.z_x) * a.inv_kk) + a.z_x, 0), 255);
          q = min(q, a.a_hi) - a.z_x;
          if (packed) pk |= ((uint32_t)q & 0xffu) << (8 * jj);
          else pooled[wave][4 * c4 + jj] = q;
        }
        if (packed) pooled[wave][c4] = (int)pk;
      }
    } else {
      for (int c = lane; c < a.C; c += 64) {
        int sum = 0;
        for (int p = 0; p < a.kk; ++p) sum += xs[p * a.C + c];
        int q = min(max(rne_sat((float)(sum - a.kk * a.z_x) * a.inv_kk) + a.z_x, 0), 255);
        pooled[wave][c] = min(q, a.a_hi) - a.z_x;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): same-wave LDS write -> read
    __builtin_amdgcn_wave_barrier();
    // Linear: lane = (output n, quarter j of the channels); integer partial sums, then a 4-lane butterfly (exact, any order)
    int acc = 0;
    if (n < a.N) {
      if (packed) {
        const int dpq = a.C / 16;                                  // dwords per quarter
        const uint32_t* wp = reinterpret_cast<const uint32_t*>(ws + n * a.C) + j * dpq;
        int ps = 0;
        for (int d = 0; d < dpq; ++d) {
          const int pv = pooled[wave][j * dpq + d];
          acc = __builtin_amdgcn_sdot4(pv, (int)wp[d], acc, false);
          ps = __builtin_amdgcn_sdot4(pv, 0x01010101, ps, false);
        }
        acc -= a.z_w * ps;
      } else {
        const int c_per = (a.C + 3) / 4, c0 = j * c_per, c1 = min(c0 + c_per, a.C);
        for (int c = c0; c < c1; ++c) acc += pooled[wave][c] * ((int)ws[n * a.C + c] - a.z_w);
      }
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    float logit = -INFINITY;
    if (n < a.N && j == 0) {
      float xf = (float)acc;
      if (a.bias) xf = __builtin_fmaf(a.bias[n], a.rcp, xf);
      int q = min(max(a.z_y + rne_sat(xf * a.mult), 0), 255);
      q = min(q, a.a_hi);
      logit = (float)(q - a.z_y) * a.s_y;     // DeQuantStub
    }
    float mx = logit;
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float e = (n < a.N && j == 0) ? expf(logit - mx) : 0.f;
    float sum = e;
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (n < a.N && j == 0) a.probs[((int64_t)s * a.B + b) * a.N + n] = e / sum;
    __builtin_amdgcn_wave_barrier();      // the next image overwrites pooled[wave]
  }
}

static int build_head_args(HeadArgs& a, const uint8_t* x, int64_t x_ss, const int8_t* w, int64_t w_ss, const float* bias, float* probs,
                           const qbnn_head_desc* d) {
  if (d->C > 256 || d->N > 16 || d->C <= 0 || d->N <= 0) return fail(QBNN_E_INVALID, "qbnn_head_i8: C <= 256 and N <= 16 required%s");
  a.x = x; a.x_ss = x_ss; a.w = w; a.w_ss = w_ss; a.bias = d->has_bias ? bias : nullptr; a.probs = probs;
  a.B = d->B; a.kk = d->k * d->k; a.C = d->C; a.N = d->N;
  a.z_x = d->z_x; a.z_w = d->z_w; a.z_y = d->z_y; a.a_hi = d->a_hi;
  a.inv_kk = 1.0f / (float)(d->k * d->k);
  const float atw = d->s_x * d->s_w;
  a.rcp = 1.0f / atw; a.mult = atw / d->s_y; a.s_y = d->s_y;
  return QBNN_OK;
}

QBNN_EXPORT int qbnn_head_i8_mc(const uint8_t* x, int64_t x_ss, const int8_t* w, int64_t w_ss, const float* bias,
                                float* probs, int32_t n_samples, const qbnn_head_desc* d, void* stream) {
  if (!x || !w || !probs || !d || n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_head_i8_mc: bad argument%s");
  ArgsArr<HeadArgs, 1> one;
  if (int rc = build_head_args(one.m[0], x, x_ss, w, w_ss, bias, probs, d)) return rc;
  hipLaunchKernelGGL(head_i8_kernel<1>, dim3(ceil_div(d->B, 4 * QBNN_HEAD_IMGS), n_samples), dim3(256), 0, (hipStream_t)stream, one);
  return check_launch("qbnn_head_i8_mc");
}

QBNN_EXPORT int qbnn_head_i8_multi(const qbnn_head_call* calls, int32_t n_calls, void* stream) {
  if (!calls || n_calls <= 0) return fail(QBNN_E_INVALID, "qbnn_head_i8_multi: bad argument%s");
  for (int c0 = 0; c0 < n_calls;) {
    const int n = n_calls - c0 < QBNN_FUSED_CALLS ? n_calls - c0 : QBNN_FUSED_CALLS;
    ArgsArr<HeadArgs, QBNN_FUSED_CALLS> all;
    memset(&all, 0, sizeof(all));
    int maxB = 0, maxS = 0;
    for (int i = 0; i < n; ++i) {
      const qbnn_head_call& k = calls[c0 + i];
      if (!k.x || !k.w || !k.probs || !k.desc || k.n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_head_i8_multi: bad call entry%s");
      if (int rc = build_head_args(all.m[i], k.x, k.x_sample_stride, k.w, k.w_sample_stride, k.bias, k.probs, k.desc)) return rc;
      if (k.n_samples != calls[c0].n_samples) return fail(QBNN_E_INVALID, "qbnn_head_i8_multi: the calls of one launch evaluate the same number of samples%s");
      maxB = k.desc->B > maxB ? k.desc->B : maxB; maxS = k.n_samples;
    }
    hipLaunchKernelGGL(head_i8_kernel<QBNN_FUSED_CALLS>, dim3(ceil_div(maxB, 4 * QBNN_HEAD_IMGS), maxS, n), dim3(256), 0, (hipStream_t)stream, all);
    if (int rc = check_launch("qbnn_head_i8_multi")) return rc;
    c0 += n;
  }
  return QBNN_OK;
}

// MC reduction.  The sums are kept in fp64: var = (sum p^2 - (sum p)^2 / S) / (S - 1) cancels catastrophically in fp32 when the
// spread of a class probability is small against its mean (the regression head's predictive variance feeds the reference's NLL).
// FINAL: this launch also finalises (single rank, last chunk): mean / unbiased variance as fp32, no further launches.
template <bool FINAL>
__global__ __launch_bounds__(256) void reduce_moments_kernel(const float* __restrict__ probs, int S, int64_t n, int accumulate,
                                                             double* __restrict__ mom, int total, float* __restrict__ mean_out,
                                                             float* __restrict__ var_out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s1 = accumulate ? mom[i] : 0.0, s2 = accumulate ? mom[n + i] : 0.0;
  int s = 0;
  for (; s + 8 <= S; s += 8) {            // 8 independent loads in flight, summed in sample order
    float p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = probs[(int64_t)(s + j) * n + i];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1 += (double)p[j]; s2 += (double)p[j] * (double)p[j]; }
  }
  for (; s < S; ++s) {
    const double p = (double)probs[(int64_t)s * n + i];
    s1 += p;
    s2 += p * p;
  }
  mom[i] = s1; mom[n + i] = s2;
  if (FINAL) {
    const double m = s1 / (double)total;
    mean_out[i] = (float)m;
    if (var_out) var_out[i] = total > 1 ? (float)(fmax(s2 - s1 * m, 0.0) / (double)(total - 1)) : 0.f;
  }
}

__global__ __launch_bounds__(256) void finalize_moments_kernel(const double* __restrict__ mom, int64_t n, int total,
                                                               float* __restrict__ mean_out, float* __restrict__ var_out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double s1 = mom[i], s2 = mom[n + i], m = s1 / (double)total;
  mean_out[i] = (float)m;
  if (var_out) var_out[i] = total > 1 ? (float)(fmax(s2 - s1 * m, 0.0) / (double)(total - 1)) : 0.f;
}

QBNN_EXPORT int qbnn_reduce_moments(const float* probs, int32_t S, int64_t n, int32_t accumulate, double* mom,
                                    int32_t finalize_total, float* mean_out, float* var_out, void* stream) {
  if (!probs || !mom || S <= 0 || n <= 0) return fail(QBNN_E_INVALID, "qbnn_reduce_moments: bad argument%s");
  if (finalize_total > 0 && !mean_out) return fail(QBNN_E_INVALID, "qbnn_reduce_moments: finalize_total > 0 needs mean_out%s");
  const dim3 grid((unsigned)((n + 255) / 256));
  if (finalize_total > 0)
    hipLaunchKernelGGL(reduce_moments_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, probs, S, n, accumulate, mom, finalize_total, mean_out, var_out);
  else
    hipLaunchKernelGGL(reduce_moments_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, probs, S, n, accumulate, mom, 0, nullptr, nullptr);
  return check_launch("qbnn_reduce_moments");
}

QBNN_EXPORT int qbnn_finalize_moments(const double* mom, int64_t n, int32_t total_samples, float* mean_out, float* var_out, void* stream) {
  if (!mom || !mean_out || n <= 0 || total_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_finalize_moments: bad argument%s");
  hipLaunchKernelGGL(finalize_moments_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mom, n, total_samples, mean_out, var_out);
  return check_launch("qbnn_finalize_moments");
}

// =====================================================================================
// MC-Dropout path (BASELINE config 2: LeNet, mcdropout/models_mc.py:75-111).  These nets are tiny and
// launch/latency-bound at their sizes: plain one-thread-per-output kernels, any geometry.
// =====================================================================================
struct GenConvArgs {
  const uint8_t* x; int64_t x_ss; const int8_t* w; int64_t w_ss; const float* bias; uint8_t* y; int64_t y_ss;
  int B, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
  int z_x, z_w, z_y, lo, hi;
  float rcp, mult;
};

__global__ __launch_bounds__(256) void conv_generic_i8_kernel(const GenConvArgs a) {
  const int64_t total = (int64_t)a.B * a.Ho * a.Wo * a.Cout;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int s = blockIdx.y;
  const int co = (int)(idx % a.Cout);
  int64_t t = idx / a.Cout;
  const int ow = (int)(t % a.Wo); t /= a.Wo;
  const int oh = (int)(t % a.Ho);
  const int b = (int)(t / a.Ho);
  const uint8_t* xs = a.x + (int64_t)s * a.x_ss + (int64_t)b * a.H * a.W * a.Cin;
  const int8_t* ws = a.w + (int64_t)s * a.w_ss + (int64_t)co * a.KH * a.KW * a.Cin;
  int acc = 0;
  for (int kh = 0; kh < a.KH; ++kh) {
    const int ih = oh * a.stride - a.pad + kh;
    if (ih < 0 || ih >= a.H) continue;
    for (int kw = 0; kw < a.KW; ++kw) {
      const int iw = ow * a.stride - a.pad + kw;
      if (iw < 0 || iw >= a.W) continue;
      const uint8_t* xp = xs + ((int64_t)ih * a.W + iw) * a.Cin;
      const int8_t* wp = ws + (kh * a.KW + kw) * a.Cin;
      for (int c = 0; c < a.Cin; ++c) acc += ((int)xp[c] - a.z_x) * ((int)wp[c] - a.z_w);
    }
  }
  float xf = (float)acc;
  if (a.bias) xf = __builtin_fmaf(a.bias[co], a.rcp, xf);
  int q = a.z_y + rne_sat(xf * a.mult);
  q = min(max(q, a.lo), a.hi);
  a.y[(int64_t)s * a.y_ss + idx] = (uint8_t)q;
}

// The same contract on the matrix pipe, any geometry.  Workgroup = 64 output pixels x 64 output channels (4 waves of
// 32 x 32), K = KH KW Cin walked in 32-byte chunks that are gathered byte by byte (im2col on the fly; Cin = 1, 20, 50, 2450 ...
// give no alignment to build on) into LDS rows of 48 bytes (conflict-free ds_read_b128 fragments).
// Neither x - z_x nor w - z_w fits a signed byte in general, so the MFMA runs on the raw bytes x' = x - 128 (= x ^ 0x80)
// and w, with out-of-map taps fed x = z_x (their exact contribution is then 0), and the zero points enter afterwards:
//   sum_k (x_k - z_x)(w_k - z_w) = acc - z_w R + a Wsum[co] - K a z_w,   a = 128 - z_x, R = sum_k x'_k, Wsum = sum_k w_k
// -- all int32-exact; R and Wsum are v_dot4 sums over the fragments the MFMA consumes.  Bit-identical to
// conv_generic_i8_kernel (tests compare the two), 50-200x faster on the LeNet / MLP layers.
// GB = 4 / 2: Cin % GB == 0 and GB-byte aligned operands -- a unit never straddles a tap, so the gather moves GB channels per
// load with one index step per unit instead of per byte (the gather's vector-ALU work is what bounds this kernel).
template <int GB>
__global__ __launch_bounds__(256) void conv_generic_mfma_i8_kernel(const GenConvArgs a) {
  constexpr int LD = 48;
  __shared__ __attribute__((aligned(16))) uint8_t As[64 * LD];      // weights [n][k]
  __shared__ __attribute__((aligned(16))) uint8_t Bs[64 * LD];      // pixels  [p][k], bytes x ^ 0x80
  __shared__ int wsum_lds[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int s = blockIdx.z;
  const int p0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int npix = a.B * a.Ho * a.Wo;
  const int K = a.KH * a.KW * a.Cin;
  const uint8_t* xs = a.x + (int64_t)s * a.x_ss;
  const int8_t* ws = a.w + (int64_t)s * a.w_ss;
  const int row = tid >> 2, kb = (tid & 3) * 8;
  const int p = p0 + row, n = n0 + row;
  int pb = -1, ih0 = 0, iw0 = 0;
  if (p < npix) { pb = p / (a.Ho * a.Wo); const int rem = p - pb * a.Ho * a.Wo; ih0 = (rem / a.Wo) * a.stride - a.pad; iw0 = (rem % a.Wo) * a.stride - a.pad; }
  const int64_t xbase = (int64_t)(pb < 0 ? 0 : pb) * a.H * a.W * a.Cin;
  const int8_t* wrow = ws + (int64_t)(n < a.Cout ? n : 0) * K;
  const uint32_t xpad = (uint32_t)(a.z_x ^ 0x80) & 0xffu;
  // the thread's 8 bytes of the chunk in units of GB = 1, 2 or 4 bytes (Cin % GB == 0: a unit never straddles a tap)
  auto gather = [&](int k0, uint32_t (&xv)[2], uint32_t (&wv)[2]) {
    int kk = k0 + kb;
    int tap = kk / a.Cin, c = kk - tap * a.Cin;
    int kh = tap / a.KW, kw = tap - kh * a.KW;
    xv[0] = xv[1] = wv[0] = wv[1] = 0u;
    constexpr uint32_t UMASK = GB == 4 ? 0xffffffffu : (GB == 2 ? 0xffffu : 0xffu);
#pragma unroll
    for (int j = 0; j < 8 / GB; ++j, kk += GB) {
      uint32_t xb = 0u, wb = 0u;
      if (kk < K) {
        if (pb >= 0) {
          const int ih = ih0 + kh, iw = iw0 + kw;
          const bool in = (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
          const uint8_t* src = xs + xbase + ((int64_t)ih * a.W + iw) * a.Cin + c;
          uint32_t v;
          if constexpr (GB == 4) v = *reinterpret_cast<const uint32_t*>(src);
          else if constexpr (GB == 2) v = *reinterpret_cast<const uint16_t*>(src);
          else v = *src;
          xb = in ? (v ^ (0x80808080u & UMASK)) : (xpad * 0x01010101u) & UMASK;
        }
        if (n < a.Cout) {
          if constexpr (GB == 4) wb = *reinterpret_cast<const uint32_t*>(wrow + kk);
          else if constexpr (GB == 2) wb = *reinterpret_cast<const uint16_t*>(wrow + kk);
          else wb = (uint32_t)(uint8_t)wrow[kk];
        }
      }
      constexpr int PER = 4 / GB;                       // units per dword
      xv[j / PER] |= xb << (8 * GB * (j % PER));
      wv[j / PER] |= wb << (8 * GB * (j % PER));
      c += GB;
      if (c == a.Cin) { c = 0; if (++kw == a.KW) { kw = 0; ++kh; } }
    }
  };
  v16i acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0;
  int rsum = 0, wsum = 0;
  uint32_t xv[2], wv[2];
  gather(0, xv, wv);
  for (int k0 = 0; k0 < K; k0 += 32) {
    *reinterpret_cast<v2i*>(&Bs[row * LD + kb]) = v2i{(int)xv[0], (int)xv[1]};
    *reinterpret_cast<v2i*>(&As[row * LD + kb]) = v2i{(int)wv[0], (int)wv[1]};
    __syncthreads();
    if (k0 + 32 < K) gather(k0 + 32, xv, wv);           // next chunk in flight under the MFMA
    const v4i av = *reinterpret_cast<const v4i*>(&As[(wn * 32 + (lane & 31)) * LD + 16 * (lane >> 5)]);
    const v4i bv = *reinterpret_cast<const v4i*>(&Bs[(wm * 32 + (lane & 31)) * LD + 16 * (lane >> 5)]);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rsum = __builtin_amdgcn_sdot4(bv[i], 0x01010101, rsum, false);
      wsum = __builtin_amdgcn_sdot4(av[i], 0x01010101, wsum, false);
    }
    __syncthreads();
  }
  const int R = rsum + __shfl_xor(rsum, 32);              // this lane's pixel (lane & 31), all k
  const int Wn = wsum + __shfl_xor(wsum, 32);             // weight row wn * 32 + (lane & 31), all k
  if (wm == 0 && lane < 32) wsum_lds[wn * 32 + lane] = Wn;
  __syncthreads();
  const int po = p0 + wm * 32 + (lane & 31);
  if (po >= npix) return;
  const int aoff = 128 - a.z_x;
  const int base = -a.z_w * R - K * aoff * a.z_w;
  uint8_t* yp = a.y + (int64_t)s * a.y_ss + (int64_t)po * a.Cout;
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nl = wn * 32 + 8 * g + 4 * (lane >> 5) + i, no = n0 + nl;
      if (no < a.Cout) {
        float xf = (float)(acc[4 * g + i] + aoff * wsum_lds[nl] + base);
        if (a.bias) xf = __builtin_fmaf(a.bias[no], a.rcp, xf);
        int q = a.z_y + rne_sat(xf * a.mult);
        q = min(max(q, a.lo), a.hi);
        yp[no] = (uint8_t)q;
      }
    }
}

static bool generic_naive() { static const bool v = [] { const char* e = getenv("QBNN_GENERIC_NAIVE"); return e && e[0] == '1'; }(); return v; }

static int conv2d_i8_generic(const uint8_t* x, int64_t x_ss, const int8_t* w_ohwi, int64_t w_ss, const float* bias, uint8_t* y,
                             int64_t y_ss, int32_t n_samples, const qbnn_conv_desc* d, void* stream, bool scalar_form);

QBNN_EXPORT int qbnn_conv2d_i8_generic_mc(const uint8_t* x, int64_t x_ss, const int8_t* w_ohwi, int64_t w_ss, const float* bias,
                                          uint8_t* y, int64_t y_ss, int32_t n_samples, const qbnn_conv_desc* d, void* stream) {
  return conv2d_i8_generic(x, x_ss, w_ohwi, w_ss, bias, y, y_ss, n_samples, d, stream, generic_naive());
}

QBNN_EXPORT int qbnn_conv2d_i8_generic_scalar_mc(const uint8_t* x, int64_t x_ss, const int8_t* w_ohwi, int64_t w_ss, const float* bias,
                                                 uint8_t* y, int64_t y_ss, int32_t n_samples, const qbnn_conv_desc* d, void* stream) {
  return conv2d_i8_generic(x, x_ss, w_ohwi, w_ss, bias, y, y_ss, n_samples, d, stream, true);
}

static int conv2d_i8_generic(const uint8_t* x, int64_t x_ss, const int8_t* w_ohwi, int64_t w_ss, const float* bias, uint8_t* y,
                             int64_t y_ss, int32_t n_samples, const qbnn_conv_desc* d, void* stream, bool scalar_form) {
  if (!x || !w_ohwi || !y || !d || n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_generic_mc: bad argument%s");
  if (d->a_hi > 255 || d->a_hi < 1) return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_generic_mc: bad a_hi%s");
  GenConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.x_ss = x_ss; a.w = w_ohwi; a.w_ss = w_ss; a.bias = d->has_bias ? bias : nullptr; a.y = y; a.y_ss = y_ss;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.KH = a.KW = d->ksize; a.stride = d->stride; a.pad = d->pad;
  a.Ho = (d->H + 2 * d->pad - d->ksize) / d->stride + 1; a.Wo = (d->W + 2 * d->pad - d->ksize) / d->stride + 1;
  a.z_x = d->z_x; a.z_w = d->z_w; a.z_y = d->z_y; a.lo = d->relu ? d->z_y : 0; a.hi = d->a_hi < 255 ? d->a_hi : 255;
  const float atw = d->s_x * d->s_w;
  a.rcp = 1.0f / atw; a.mult = atw / d->s_y;
  if (a.Ho <= 0 || a.Wo <= 0 || a.Cin <= 0 || a.Cout <= 0) return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_generic_mc: empty geometry%s");
  const int64_t npix = (int64_t)a.B * a.Ho * a.Wo;
  const int64_t total = npix * a.Cout;
  if (scalar_form || (int64_t)a.KH * a.KW * a.Cin > (1 << 16))      // (int32 head-room of the correction terms)
    hipLaunchKernelGGL(conv_generic_i8_kernel, dim3((unsigned)((total + 255) / 256), n_samples), dim3(256), 0, (hipStream_t)stream, a);
  else {
    const dim3 grid((unsigned)((npix + 63) / 64), (unsigned)((a.Cout + 63) / 64), n_samples);
    auto unit_ok = [&](int gb) {
      return (a.Cin % gb) == 0 && (x_ss % gb) == 0 && (w_ss % gb) == 0 && (reinterpret_cast<uintptr_t>(x) % gb) == 0 &&
             (reinterpret_cast<uintptr_t>(w_ohwi) % gb) == 0;
    };
    if (unit_ok(4)) hipLaunchKernelGGL(conv_generic_mfma_i8_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (unit_ok(2)) hipLaunchKernelGGL(conv_generic_mfma_i8_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(conv_generic_mfma_i8_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a);
  }
  return check_launch("qbnn_conv2d_i8_generic_mc");
}

// Quantised BernoulliDropout (mcdropout/dropout.py:15-40), x [S][B][HW][C]: one Bernoulli(keep) draw per (sample, b, c)
// from the Philox uniform stream {ctr = {i >> 2, layer, sample, 1}}[i & 3], i = b * C + c  (or mask_in in parity mode).
// One workgroup = one (sample, image): thread t owns channel slot t % CS (CS = C, or C / 4 dwords when C % 4 == 0) and
// draws that slot's mask ONCE, then walks the pixels t / CS, t / CS + 256 / CS, ... -- consecutive threads touch consecutive
// bytes.  (The first form drew a Philox block per element: HW-fold redundant, 2 ms per LeNet pass.)
template <bool VEC4>
__global__ __launch_bounds__(256) void dropout_q_kernel(const uint8_t* __restrict__ x, int64_t x_ss, int B, int HW, int C,
                                                         float keep, int z_x, float inv_sm, int z_m, float mult, int hi,
                                                         uint32_t seed_lo, uint32_t seed_hi, uint32_t layer_id, uint32_t sample_begin,
                                                         const float* __restrict__ mask_in, uint8_t* __restrict__ y, int64_t y_ss,
                                                         const uint32_t* __restrict__ nd) {
  if (nd) { seed_lo = nd[0]; seed_hi = nd[1]; sample_begin = nd[2]; }      // captured-graph mode: the seed lives in device memory
  const int b = blockIdx.x, s = blockIdx.y;
  const int CS = VEC4 ? C / 4 : C;
  const int per_pass = 256 / CS > 0 ? 256 / CS : 1;          // pixels covered by the workgroup per trip (CS <= 256), else slots loop
  const uint8_t* xs = x + (int64_t)s * x_ss + (int64_t)b * HW * C;
  uint8_t* ys = y + (int64_t)s * y_ss + (int64_t)b * HW * C;
  auto mask_q = [&](int c) {                                   // quantised mask value minus its zero point, channel c of image b
    return drop_mask_q(b * C + c, s, (int64_t)B * C, keep, inv_sm, z_m, seed_lo, seed_hi, layer_id, sample_begin, mask_in);
  };
  auto one = [&](int xb, int mq) { return drop_one(xb, mq, z_x, z_m, mult, hi); };
  for (int slot = threadIdx.x % (CS < 256 ? CS : 256); slot < CS; slot += 256) {       // one trip unless C > 256 (VEC4: C > 1024)
    const int first = CS < 256 ? threadIdx.x / CS : 0;
    if (CS < 256 && first >= per_pass) break;                                       // threads beyond a whole number of pixels idle
    if constexpr (VEC4) {
      int mq[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) mq[j] = mask_q(4 * slot + j);
      for (int hw = first; hw < HW; hw += per_pass) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>(xs + (int64_t)hw * C + 4 * slot);
        const uint32_t o = one((int)(v & 0xffu), mq[0]) | (one((int)((v >> 8) & 0xffu), mq[1]) << 8) |
                           (one((int)((v >> 16) & 0xffu), mq[2]) << 16) | (one((int)(v >> 24), mq[3]) << 24);
        *reinterpret_cast<uint32_t*>(ys + (int64_t)hw * C + 4 * slot) = o;
      }
    } else {
      const int mq = mask_q(slot);
      for (int hw = first; hw < HW; hw += per_pass) ys[(int64_t)hw * C + slot] = (uint8_t)one((int)xs[(int64_t)hw * C + slot], mq);
    }
  }
}

QBNN_EXPORT int qbnn_dropout_q_mc(const uint8_t* x, int64_t x_ss, int32_t B, int32_t HW, int32_t C, float keep_prob, float s_x,
                                  int32_t z_x, float s_m, int32_t z_m, int32_t a_hi, uint64_t seed, uint32_t layer_id,
                                  uint32_t sample_begin, const float* mask_in, uint8_t* y, int64_t y_ss, int32_t n_samples,
                                  void* stream) {
  if (!x || !y || B <= 0 || HW <= 0 || C <= 0 || n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_dropout_q_mc: bad argument%s");
  const float mult = (float)((double)s_x * (double)s_m / (double)s_m);     // ATen qmul: self_scale * other_scale / out_scale
  const bool vec4 = (C % 4) == 0 && (x_ss % 4) == 0 && (y_ss % 4) == 0 && (reinterpret_cast<uintptr_t>(x) % 4) == 0 &&
                    (reinterpret_cast<uintptr_t>(y) % 4) == 0;
  if (vec4)
    hipLaunchKernelGGL(dropout_q_kernel<true>, dim3((unsigned)B, n_samples), dim3(256), 0, (hipStream_t)stream,
                       x, x_ss, B, HW, C, keep_prob, z_x, 1.0f / s_m, z_m, mult, a_hi < 255 ? a_hi : 255, (uint32_t)seed,
                       (uint32_t)(seed >> 32), layer_id, sample_begin, mask_in, y, y_ss, g_noise_dev);
  else
    hipLaunchKernelGGL(dropout_q_kernel<false>, dim3((unsigned)B, n_samples), dim3(256), 0, (hipStream_t)stream,
                       x, x_ss, B, HW, C, keep_prob, z_x, 1.0f / s_m, z_m, mult, a_hi < 255 ? a_hi : 255, (uint32_t)seed,
                       (uint32_t)(seed >> 32), layer_id, sample_begin, mask_in, y, y_ss, g_noise_dev);
  return check_launch("qbnn_dropout_q_mc");
}

__global__ __launch_bounds__(256) void maxpool2_q_kernel(const uint8_t* __restrict__ x, int64_t x_ss, int B, int H, int W, int C,
                                                          int hi, uint8_t* __restrict__ y, int64_t y_ss) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)B * Ho * Wo * C;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int s = blockIdx.y;
  const int c = (int)(idx % C);
  int64_t t = idx / C;
  const int ow = (int)(t % Wo); t /= Wo;
  const int oh = (int)(t % Ho);
  const int64_t b = t / Ho;
  const uint8_t* xs = x + (int64_t)s * x_ss;
  int m = 0;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j) m = max(m, (int)xs[((b * H + oh * 2 + i) * W + ow * 2 + j) * C + c]);
  y[(int64_t)s * y_ss + idx] = (uint8_t)min(m, hi);
}

QBNN_EXPORT int qbnn_maxpool2_q_mc(const uint8_t* x, int64_t x_ss, int32_t B, int32_t H, int32_t W, int32_t C, int32_t a_hi,
                                   uint8_t* y, int64_t y_ss, int32_t n_samples, void* stream) {
  if (!x || !y || B <= 0 || H < 2 || W < 2 || C <= 0 || n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_maxpool2_q_mc: bad argument%s");
  const int64_t total = (int64_t)B * (H / 2) * (W / 2) * C;
  hipLaunchKernelGGL(maxpool2_q_kernel, dim3((unsigned)((total + 255) / 256), n_samples), dim3(256), 0, (hipStream_t)stream,
                     x, x_ss, B, H, W, C, a_hi < 255 ? a_hi : 255, y, y_ss);
  return check_launch("qbnn_maxpool2_q_mc");
}

// DeQuantStub + softmax over the last dim (models_mc.py:104-111): x [S][B][N] uint8 -> probs [S][B][N] fp32
__global__ __launch_bounds__(256) void dequant_softmax_kernel(const uint8_t* __restrict__ x, int64_t x_ss, int B, int N, float sc,
                                                               int z, float* __restrict__ probs) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const int s = blockIdx.y;
  const uint8_t* xs = x + (int64_t)s * x_ss + (int64_t)b * N;
  float mx = -INFINITY;
  for (int i = 0; i < N; ++i) mx = fmaxf(mx, (float)((int)xs[i] - z) * sc);
  float sum = 0.f;
  for (int i = 0; i < N; ++i) sum += expf((float)((int)xs[i] - z) * sc - mx);
  for (int i = 0; i < N; ++i) probs[((int64_t)s * B + b) * N + i] = expf((float)((int)xs[i] - z) * sc - mx) / sum;
}

QBNN_EXPORT int qbnn_dequant_softmax_mc(const uint8_t* x, int64_t x_ss, int32_t B, int32_t N, float scale, int32_t zero_point,
                                        float* probs, int32_t n_samples, void* stream) {
  if (!x || !probs || B <= 0 || N <= 0 || n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_dequant_softmax_mc: bad argument%s");
  hipLaunchKernelGGL(dequant_softmax_kernel, dim3((B + 255) / 256, n_samples), dim3(256), 0, (hipStream_t)stream, x, x_ss, B, N,
                     scale, zero_point, probs);
  return check_launch("qbnn_dequant_softmax_mc");
}

// =====================================================================================
// fp32 Bayes-by-backprop path (BASELINE config 0: 3x100 MLP; reference bbb/linear.py:42-50).  Tiny, latency-bound:
// plain VALU kernels.  W = mu + eps * sigma is two fp32 roundings (FloatFunctional mul then add), as in the reference.
// =====================================================================================
__global__ __launch_bounds__(256) void sample_weights_f32_kernel(const float* __restrict__ mu, const float* __restrict__ sigma,
                                                                  int64_t n, uint32_t seed_lo, uint32_t seed_hi, uint32_t layer_id,
                                                                  uint32_t sample_begin, const float* __restrict__ eps_in,
                                                                  float* __restrict__ w, const uint32_t* __restrict__ nd) {
  if (nd) { seed_lo = nd[0]; seed_hi = nd[1]; sample_begin = nd[2]; }      // captured-graph mode: the seed lives in device memory
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;      // group of 4 consecutive weights
  if (g * 4 >= n) return;
  const int s = blockIdx.y;
  float e[4];
  if (eps_in) {
    for (int j = 0; j < 4; ++j) e[j] = (g * 4 + j < n) ? eps_in[(int64_t)s * n + g * 4 + j] : 0.f;
  } else {
    qbnn::normal4(qbnn::philox4x32_10((uint32_t)g, layer_id, sample_begin + s, 0u, seed_lo, seed_hi), e);
  }
  if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(mu) | reinterpret_cast<uintptr_t>(sigma) | reinterpret_cast<uintptr_t>(w)) & 15) == 0) {
    // whole group in range and 16-byte aligned: one vector load per operand, one vector store (same arithmetic per element)
    const float4 m4 = reinterpret_cast<const float4*>(mu)[g], s4 = reinterpret_cast<const float4*>(sigma)[g];
    float4 o;
    { const float t = e[0] * s4.x; o.x = m4.x + t; }
    { const float t = e[1] * s4.y; o.y = m4.y + t; }
    { const float t = e[2] * s4.z; o.z = m4.z + t; }
    { const float t = e[3] * s4.w; o.w = m4.w + t; }
    reinterpret_cast<float4*>(w + (int64_t)s * n)[g] = o;
    return;
  }
  for (int j = 0; j < 4; ++j) {
    const int64_t i = g * 4 + j;
    if (i < n) { const float t = e[j] * sigma[i]; w[(int64_t)s * n + i] = mu[i] + t; }
  }
}

QBNN_EXPORT int qbnn_sample_weights_f32(const float* mu, const float* sigma, int64_t n, uint64_t seed, uint32_t layer_id,
                                        uint32_t sample_begin, int32_t n_samples, const float* eps_in, float* w_out, void* stream) {
  if (!mu || !sigma || !w_out || n <= 0 || n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_sample_weights_f32: bad argument%s");
  const int64_t groups = (n + 3) / 4;
  hipLaunchKernelGGL(sample_weights_f32_kernel, dim3((unsigned)((groups + 255) / 256), n_samples), dim3(256), 0, (hipStream_t)stream,
                     mu, sigma, n, (uint32_t)seed, (uint32_t)(seed >> 32), layer_id, sample_begin, eps_in, w_out, g_noise_dev);
  return check_launch("qbnn_sample_weights_f32");
}

__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ x, int64_t x_ss, const float* __restrict__ w,
                                                          int64_t w_ss, const float* __restrict__ bias, float* __restrict__ y,
                                                          int64_t y_ss, int B, int K, int N, int act) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * N) return;
  const int s = blockIdx.y;
  const int n = (int)(idx % N), b = (int)(idx / N);
  const float* xp = x + (int64_t)s * x_ss + (int64_t)b * K;
  const float* wp = w + (int64_t)s * w_ss + (int64_t)n * K;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = __builtin_fmaf(xp[k], wp[k], acc);
  if (bias) acc = acc + bias[n];
  if (act == 1) acc = fmaxf(acc, 0.f);
  else if (act == 2) acc = expf(acc);
  y[(int64_t)s * y_ss + idx] = acc;
}

QBNN_EXPORT int qbnn_linear_f32_mc(const float* x, int64_t x_ss, const float* w, int64_t w_ss, const float* bias, float* y,
                                   int64_t y_ss, int32_t B, int32_t K, int32_t N, int32_t act, int32_t n_samples, void* stream) {
  if (!x || !w || !y || B <= 0 || K <= 0 || N <= 0 || n_samples <= 0 || act < 0 || act > 2)
    return fail(QBNN_E_INVALID, "qbnn_linear_f32_mc: bad argument%s");
  // a Linear is a 1x1 conv over a 1x1 map with [out][in] = OHWI weights: the MFMA implicit-GEMM kernels of qbnn_f32.hip
  // (float4 path when in_features % 4 == 0); only the exp head (N = 1) stays on the one-thread-per-output kernel
  if (act != 2)
    return qbnn_conv2d_f32_fused_mc(x, x_ss, w, w_ss, nullptr, bias, nullptr, nullptr, nullptr, 0, y, y_ss, B, 1, 1, K, N, 1, 1, 0,
                                    (act == 1 ? 1 : 0) | 4, n_samples, nullptr, stream);
  const int64_t total = (int64_t)B * N;
  hipLaunchKernelGGL(linear_f32_kernel, dim3((unsigned)((total + 255) / 256), n_samples), dim3(256), 0, (hipStream_t)stream,
                     x, x_ss, w, w_ss, bias, y, y_ss, B, K, N, act);
  return check_launch("qbnn_linear_f32_mc");
}

// =====================================================================================
// Classification metrics on the reduced output (reference src/metrics.py:8-116, :355-430), on device so the [B,C]
// predictive mean need not return to the host per batch.  One thread per image; per-block partial sums
//   [0] errors  [1] sum -log(p_target + 1e-8)  [2] sum_c (p - onehot)^2  [3] sum_c -p log(p + 1e-8)
//   [4+b] count, [14+b] confidence sum, [24+b] accuracy sum of calibration bin b (10 uniform bins on max-prob;
//   bin = index of the first boundary >= confidence, minus 1: torch.bucketize(conf, linspace(0,1,11), right=True) - 1).
// The host sums the partial rows (deterministic).
// =====================================================================================
#define QBNN_METRIC_SLOTS 34
__global__ __launch_bounds__(256) void classification_metrics_kernel(const float* __restrict__ probs, const int64_t* __restrict__ target,
                                                                      int B, int C, float* __restrict__ partials) {
  __shared__ float red[QBNN_METRIC_SLOTS][4];
  const int b = blockIdx.x * 256 + threadIdx.x;
  float v[QBNN_METRIC_SLOTS];
#pragma unroll
  for (int i = 0; i < QBNN_METRIC_SLOTS; ++i) v[i] = 0.f;
  if (b < B) {
    const float* p = probs + (int64_t)b * C;
    const int t = (int)target[b];
    int am = 0; float conf = p[0], brier = 0.f, ent = 0.f;
    for (int c = 0; c < C; ++c) {
      const float pc = p[c];
      if (pc > conf) { conf = pc; am = c; }
      const float oh = c == t ? 1.f : 0.f;
      brier += (pc - oh) * (pc - oh);
      ent += -pc * logf(pc + 1e-8f);
    }
    const float acc = am == t ? 1.f : 0.f;
    v[0] = 1.f - acc;
    v[1] = -logf(p[t] + 1e-8f);
    v[2] = brier;
    v[3] = ent;
    int bin = 0;                                   // boundaries k/10: right=True -> first k with k/10 > conf ... minus 1
    for (int k = 1; k <= 10; ++k) bin = (conf >= (float)k * 0.1f) ? k : bin;
    bin = min(bin, 9);
    v[4 + bin] = 1.f; v[14 + bin] = conf; v[24 + bin] = acc;
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
  for (int i = 0; i < QBNN_METRIC_SLOTS; ++i) {
    float x = v[i];
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    if (lane == 0) red[i][wave] = x;
  }
  __syncthreads();
  if (threadIdx.x < QBNN_METRIC_SLOTS)
    partials[(int64_t)blockIdx.x * QBNN_METRIC_SLOTS + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

QBNN_EXPORT int qbnn_classification_metrics(const float* probs, const int64_t* target, int32_t B, int32_t C, float* partials, void* stream) {
  if (!probs || !target || !partials || B <= 0 || C <= 0) return fail(QBNN_E_INVALID, "qbnn_classification_metrics: bad argument%s");
  hipLaunchKernelGGL(classification_metrics_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, probs, target, B, C, partials);
  return check_launch("qbnn_classification_metrics");
}

// Regression metrics on the reduced MC output (reference src/metrics.py:119-230 fed by RegressionMetric.update :468-500):
//   partial sums per 256-row block of   nll = 0.5 log(2 pi var + 1e-8) + (t - mean)^2 / (2 var + 1e-8)   (:143),
//   squared error (:186), absolute error (:224).  Accumulated in fp64 by the caller.
__global__ __launch_bounds__(256) void regression_metrics_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                                                                  const float* __restrict__ target, int B, float* __restrict__ partials) {
  __shared__ float red[3][4];
  const int b = blockIdx.x * 256 + threadIdx.x;
  float v[3] = {0.f, 0.f, 0.f};
  if (b < B) {
    const float m = mean[b], vr = var ? var[b] : 1.0f, t = target[b];
    const float d = t - m;
    v[0] = 0.5f * logf(2.0f * 3.14159265358979323846f * vr + 1e-8f) + d * d / (2.0f * vr + 1e-8f);
    v[1] = d * d;
    v[2] = fabsf(d);
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float x = v[i];
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    if (lane == 0) red[i][wave] = x;
  }
  __syncthreads();
  if (threadIdx.x < 3) partials[(int64_t)blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

QBNN_EXPORT int qbnn_regression_metrics(const float* mean, const float* var, const float* target, int32_t B, float* partials, void* stream) {
  if (!mean || !target || !partials || B <= 0) return fail(QBNN_E_INVALID, "qbnn_regression_metrics: bad argument%s");
  hipLaunchKernelGGL(regression_metrics_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean, var, target, B, partials);
  return check_launch("qbnn_regression_metrics");
}

// Flatten (reference src/utils.py:40-47) of a channels-last activation into the reference's NCHW feature order:
// x [S][B][HW][C] -> y [S][B][C*HW], y[c * HW + p] = x[p * C + c].  Needed where a stochastic Linear follows a conv map:
// its noise stream is indexed by the reference's (c, h, w) column order.
__global__ __launch_bounds__(256) void flatten_nchw_kernel(const uint8_t* __restrict__ x, int64_t x_ss, int B, int HW, int C,
                                                            uint8_t* __restrict__ y, int64_t y_ss) {
  const int64_t total = (int64_t)B * HW * C;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int s = blockIdx.y;
  const int64_t b = idx / ((int64_t)HW * C);
  const int r = (int)(idx - b * HW * C);
  const int c = r / HW, p = r - c * HW;
  y[(int64_t)s * y_ss + idx] = x[(int64_t)s * x_ss + (b * HW + p) * C + c];
}

QBNN_EXPORT int qbnn_flatten_nchw_mc(const uint8_t* x, int64_t x_ss, int32_t B, int32_t HW, int32_t C, uint8_t* y, int64_t y_ss,
                                     int32_t n_samples, void* stream) {
  if (!x || !y || B <= 0 || HW <= 0 || C <= 0 || n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_flatten_nchw_mc: bad argument%s");
  const int64_t total = (int64_t)B * HW * C;
  hipLaunchKernelGGL(flatten_nchw_kernel, dim3((unsigned)((total + 255) / 256), n_samples), dim3(256), 0, (hipStream_t)stream,
                     x, x_ss, B, HW, C, y, y_ss);
  return check_launch("qbnn_flatten_nchw_mc");
}
