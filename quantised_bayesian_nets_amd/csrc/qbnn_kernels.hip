// libqbnn_hip.so -- hand-written gfx950 (CDNA4) kernels + C ABI (include/qbnn.h).
// Compiled with:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC
// (-ffp-contract=off: bit-exactness with the reference's ATen/FBGEMM arithmetic depends on
//  where a multiply-add is fused and where it is not; every fma below is explicit.)
// This translation unit: error reporting, the packed weight layout, the int8 weight sampler and the layer-level conv
// (qbnn_conv2d_i8_mc).  Fused BasicBlock kernels: qbnn_blocks.hip, qbnn_w16.hip; input / head / reduction / MC-Dropout and
// metrics kernels: qbnn_misc.hip; fp32 / QAT path: qbnn_f32.hip; small networks: qbnn_small.hip.
#include "qbnn_host.h"
#define QBNN_EPS_TABLE_QUALIFIER __device__ static const
#include "qbnn_eps_table.h"

static thread_local char g_err[512] = "";

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: `done` keeps one bit per device ordinal
// (set after a successful call), so a process that runs models on several GPUs, or from several threads, sets it on each.
int qbnn_ensure_dyn_lds(const void* fn, std::atomic<uint64_t>* done_, int bytes) {
  std::atomic<uint64_t>& done = *done_;
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
static thread_local const unsigned int* g_noise_dev_tls = nullptr;
const unsigned int* qbnn_noise_dev() { return g_noise_dev_tls; }
QBNN_EXPORT int qbnn_set_device_noise_source(const uint32_t* dev_seed3) { g_noise_dev_tls = dev_seed3; return QBNN_OK; }

int qbnn_fail_msg(int code, const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return code; }
int qbnn_check_launch_msg(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return QBNN_E_LAUNCH;
  }
  return QBNN_OK;
}

QBNN_EXPORT const char* qbnn_last_error(void) { return g_err; }
QBNN_EXPORT int qbnn_version(void) { return QBNN_ABI_VERSION; }

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
//   QBNN_LAYOUT_MFMA32_N24 (round 5; cout % 24 == 0): the same with 24 output channels per tile -- tile nt covers channels
//       [24 nt, 24 nt + 24), row 24 of EVERY tile is a ones row, rows 25..31 are 0.  A 48-channel conv is then two independent
//       (24 + 1)-row halves: each half's MFMAs deliver its own window sum, so a wave can requantise one half while the other half's
//       MFMAs are still in flight (qbnn_c48.hip); with the 32-row tiling the window sum of tile 0 lives in tile 1.
//   QBNN_LAYOUT_MFMA32_TAIL (round 5): MFMA32 with the kernel rows' ragged ends gathered.  A kernel row of krow = 32 F + T bytes (T > 0) costs
//       F + 1 k-steps above, the last one mostly padding: 3 x 3 taps of 24 channels = 72-byte rows -> 3 x 3 = 9 k-steps for 216 weights per
//       output channel.  Here the packed K axis is [row 0's first 32 F bytes | row 1's | ... | the rows' T-byte tails back to back | 0]:
//       rows * F + 1 k-steps (7 for the 24-channel convs), when rows * T <= 32 and T % 4 == 0.  Only the layer-1 kernel (qbnn_w16.hip) reads it.
//   QBNN_LAYOUT_MFMA32_N24_TAIL: both (24 output channels per tile AND gathered tails): stem.0 of the 24 -> 48 down block (qbnn_c48.hip).
// =====================================================================================

struct PackGeom { int cout, k, krow, rows, rbp, kp, KS, NT, tr, tail_f; };      // tr: output channels per tile (32, or 24 for _N24); tail_f: F of _TAIL, else 0
static inline PackGeom pack_geom(int cout, int k, int krow, int layout = QBNN_LAYOUT_MFMA32) {
  PackGeom g;
  g.cout = cout; g.k = k; g.krow = krow; g.rows = k / krow; g.rbp = ceil_div(krow, 32) * 32;
  g.tr = (layout == QBNN_LAYOUT_MFMA32_N24 || layout == QBNN_LAYOUT_MFMA32_N24_TAIL) ? 24 : 32;
  g.tail_f = (layout == QBNN_LAYOUT_MFMA32_TAIL || layout == QBNN_LAYOUT_MFMA32_N24_TAIL) ? krow / 32 : 0;
  g.kp = g.rows * g.rbp; g.KS = g.tail_f ? g.rows * g.tail_f + 1 : g.kp / 32; g.NT = ceil_div(cout, g.tr);
  return g;
}
static inline bool is_mfma_layout(int layout) {
  return layout == QBNN_LAYOUT_MFMA32 || layout == QBNN_LAYOUT_MFMA32_N24 || layout == QBNN_LAYOUT_MFMA32_TAIL || layout == QBNN_LAYOUT_MFMA32_N24_TAIL;
}
static inline bool layout_n24(int layout) { return layout == QBNN_LAYOUT_MFMA32_N24 || layout == QBNN_LAYOUT_MFMA32_N24_TAIL; }
static inline bool layout_tail(int layout) { return layout == QBNN_LAYOUT_MFMA32_TAIL || layout == QBNN_LAYOUT_MFMA32_N24_TAIL; }
// shapes the layout is defined for: ragged kernel rows whose tails fit ONE k-step, whole Philox blocks per tail
static inline bool tail_layout_ok(int k, int krow) {
  if (krow <= 0 || k % krow) return false;
  const int T = krow % 32, rows = k / krow;
  return krow > 32 && T > 0 && T % 4 == 0 && rows * T <= 32;
}
// position on the packed K axis of byte j of kernel row kh
static inline int packed_kp(const PackGeom& g, int kh, int j) {
  if (!g.tail_f) return kh * g.rbp + j;
  const int m = 32 * g.tail_f;
  return j < m ? kh * m + j : g.rows * m + kh * (g.krow - m) + (j - m);
}

QBNN_EXPORT size_t qbnn_packed_weight_bytes(int32_t cout, int32_t k, int32_t krow, int32_t layout) {
  if (layout == QBNN_LAYOUT_ROWMAJOR) return ((size_t)cout * k + 15) / 16 * 16;
  if (!is_mfma_layout(layout) || krow <= 0 || k % krow || (layout_n24(layout) && cout % 24)) return 0;
  if (layout_tail(layout) && !tail_layout_ok(k, krow)) return 0;
  const PackGeom g = pack_geom(cout, k, krow, layout);
  return (size_t)g.NT * g.KS * 1024;
}

QBNN_EXPORT int qbnn_pack_weights_host(const int8_t* src, int32_t cout, int32_t k, int32_t krow, int32_t layout, int8_t* dst) {
  if (!src || !dst || cout <= 0 || k <= 0) return fail(QBNN_E_INVALID, "qbnn_pack_weights_host: bad argument%s");
  if (layout == QBNN_LAYOUT_ROWMAJOR) {
    memset(dst, 0, qbnn_packed_weight_bytes(cout, k, krow, layout));
    memcpy(dst, src, (size_t)cout * k);
    return QBNN_OK;
  }
  if (!is_mfma_layout(layout)) return fail(QBNN_E_INVALID, "qbnn_pack_weights_host: unknown layout%s");
  if (krow <= 0 || k % krow) return fail(QBNN_E_INVALID, "qbnn_pack_weights_host: k must be a multiple of krow%s");
  if (layout_n24(layout) && cout % 24) return fail(QBNN_E_INVALID, "qbnn_pack_weights_host: the N24 layout takes cout % 24 == 0%s");
  const PackGeom g = pack_geom(cout, k, krow, layout);
  memset(dst, 0, (size_t)g.NT * g.KS * 1024);
  if (layout_tail(layout) && !tail_layout_ok(k, krow)) return fail(QBNN_E_INVALID, "qbnn_pack_weights_host: the TAIL layout takes ragged kernel rows whose tails fit one k-step%s");
  auto put = [&](int nt, int col, int kk, int8_t v) {
    const int kp = packed_kp(g, kk / krow, kk % krow);
    const int ks = kp >> 5, half = (kp >> 4) & 1, b = kp & 15;
    dst[(((size_t)nt * g.KS + ks) * 64 + half * 32 + col) * 16 + b] = v;
  };
  for (int n = 0; n < cout; ++n)
    for (int kk = 0; kk < k; ++kk) put(n / g.tr, n % g.tr, kk, src[(size_t)n * k + kk]);
  for (int kk = 0; kk < k; ++kk) {
    if (g.tr == 24) { for (int nt = 0; nt < g.NT; ++nt) put(nt, 24, kk, (int8_t)1); }      // a ones row in every tile
    else if (cout % 32) put(cout >> 5, cout & 31, kk, (int8_t)1);                          // the ones row behind a ragged cout
  }
  return QBNN_OK;
}

// =====================================================================================
// Weight sampler: one thread = one 16-byte chunk of the packed layout for one MC sample.
// HBM traffic: reads 2 B/weight (mu_q, sigma_q; L2-resident across samples), writes 1 B/weight/sample.
// =====================================================================================

// eps_q drawn directly: one 32-bit Philox word through the alias table of clamp(rne(N(0,1) / s_n), -128, 127)
// (qbnn_eps_table.h; `tab` = the workgroup's LDS copy).  Integer compare only: the same bits as oracle/qbnn_oracle.c.
__device__ __forceinline__ int eps_q_from_u32(uint32_t u, const uint32_t* tab) {
  const uint32_t e = tab[u >> 24];
  return (int)(((u & 0xffffffu) < (e >> 8)) ? (u >> 24) : (e & 0xffu)) - 128;
}
__device__ __forceinline__ void load_eps_table(uint32_t* tab, int tid) {       // 256 threads: one entry each
  tab[tid & 255] = QBNN_EPS_ALIAS[tid & 255];
}

// (kernel row, byte within it, valid) of byte `b` (0 .. 15) of the 16-byte chunk (k-step ks, k-half h) of a fragment tile.  tail_f = 0: the
// padded-row layouts (kernel rows of rbp bytes); tail_f = F: QBNN_LAYOUT_MFMA32_TAIL (rows of 32 F bytes, then the rows' tails in one k-step).
__device__ __forceinline__ void chunk_kpos(int ks, int h, int b, int rbp, int krow, int rows, int tail_f, int& kh, int& j, bool& valid) {
  if (tail_f == 0) {
    const int kp = ks * 32 + 16 * h + b;
    kh = kp / rbp; j = kp - kh * rbp; valid = j < krow;
  } else if (ks < rows * tail_f) {
    kh = ks / tail_f; j = (ks - kh * tail_f) * 32 + 16 * h + b; valid = true;
  } else {
    const int T = krow - 32 * tail_f, q = 16 * h + b;
    kh = q / T; j = 32 * tail_f + (q - kh * T); valid = q < rows * T;
  }
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
    const float* __restrict__ eps_in, int8_t* __restrict__ w_out, int64_t w_sample_stride, const uint32_t* __restrict__ nd, int tr, int tail_f) {
  if (nd) { seed_lo = nd[0]; seed_hi = nd[1]; sample_begin = nd[2]; }      // captured-graph mode: the seed lives in device memory
  __shared__ uint32_t eps_tab[256];
  load_eps_table(eps_tab, threadIdx.x);
  __syncthreads();
  const int chunk = blockIdx.x * 256 + threadIdx.x;
  if (chunk >= n_chunks) return;
  const int s = blockIdx.y;
  int n = 0, col = 0, cks = 0, ch = 0;       // MFMA32 (tr = 32) / _N24 (tr = 24) / _TAIL: output row, this chunk's k-step and k-half
  if (layout == QBNN_LAYOUT_MFMA32) {
    const int lane = chunk & 63, tile = chunk >> 6;
    const int nt = tile / KS;
    cks = tile - nt * KS;
    col = lane & 31;
    n = nt * tr + col;
    ch = lane >> 5;
  }
  const bool ones_row = (layout == QBNN_LAYOUT_MFMA32) && (tr == 24 ? col == 24 : ((cout & 31) && n == cout));
  const bool real_row = col < tr && n < cout;
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
      int kh, jj; bool kv;
      chunk_kpos(cks, ch, j, rbp, krow, K / krow, tail_f, kh, jj, kv);
      valid = real_row && kv;
      idx = (int64_t)n * K + kh * krow + jj;
      if (ones_row && kv) ow[j >> 2] |= 1u << (8 * (j & 3));
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
  if (!is_mfma_layout(layout) && layout != QBNN_LAYOUT_ROWMAJOR)
    return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8: unknown layout%s");
  if (is_mfma_layout(layout) && (krow <= 0 || k % krow))
    return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8: k must be a multiple of krow%s");
  if (layout_n24(layout) && cout % 24) return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8: the N24 layout takes cout % 24 == 0%s");
  if (layout_tail(layout) && !tail_layout_ok(k, krow)) return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8: the TAIL layout takes ragged kernel rows whose tails fit one k-step%s");
  const size_t bytes = qbnn_packed_weight_bytes(cout, k, krow, layout);
  if ((size_t)w_sample_stride < bytes || (w_sample_stride & 15))
    return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8: w_sample_stride too small or not 16-byte aligned%s");
  const int n_chunks = (int)(bytes / 16);
  PackGeom g = pack_geom(cout, k, is_mfma_layout(layout) ? krow : k, layout);
  dim3 grid(ceil_div(n_chunks, 256), n_samples);
  // (inside the kernels both fragment layouts are "MFMA32" with g.tr channels per tile)
  hipLaunchKernelGGL(sample_weights_i8_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                     (const v4i*)mu_packed, (const v4i*)sigma_packed, cout, k, g.krow, g.rbp, g.KS, is_mfma_layout(layout) ? QBNN_LAYOUT_MFMA32 : layout, n_chunks, *hp,
                     (uint32_t)seed, (uint32_t)(seed >> 32), layer_id, sample_begin, eps_in, w_out, w_sample_stride, g_noise_dev, g.tr, g.tail_f);
  return check_launch("qbnn_sample_weights_i8");
}

// ---- all stochastic layers of a model in ONE launch -------------------------------------------------------------
#define QBNN_MAX_SAMPLER_LAYERS 24
struct SamplerLayer {
  const v4i* mu; const v4i* sigma; int8_t* out; int64_t out_ss;
  int cout, K, krow, rbp, KS, layout, n_chunks, chunk_begin;     // chunk_begin: first 256-thread block of this layer; layout: MFMA32 (both fragment layouts) or ROWMAJOR
  int tr;                                                          // output channels per fragment tile: 32, or 24 (QBNN_LAYOUT_MFMA32_N24)
  int tail_f;                                                      // QBNN_LAYOUT_MFMA32_TAIL: full k-steps per kernel row, else 0
  uint32_t layer_id;
  qbnn_sample_params p;
};
struct SamplerTable { SamplerLayer l[QBNN_MAX_SAMPLER_LAYERS]; int n; };
static_assert(sizeof(SamplerTable) <= 3968, "the sampler table travels as a kernel argument (4 KiB incl. the other arguments)");

#ifndef QBNN_SAMPLER_NS
#define QBNN_SAMPLER_NS 4
#endif                              // MC samples per thread: the chunk's mu / sigma are loaded, unpacked and dequantised once for all of them
// UNALIGNED: the kernel of the layers whose chunks are not aligned to Philox blocks (K or the kernel-row length no multiple of 4) and are
// large enough to matter (LeNet's 2450-wide Linear); a kernel of its own, because compiled into the same kernel that path costs the
// aligned one 8 % (register allocation).  Small unaligned layers (layers.0: K = 27) stay on the element-wise path of the main kernel.
template <bool UNALIGNED>
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
  int n = 0, kh = 0, j0 = 0, col = 0, cks = 0, ch = 0;
  if (L.layout == QBNN_LAYOUT_MFMA32) {
    const int lane = chunk & 63, tile = chunk >> 6;
    const int nt = tile / L.KS, ks = tile - nt * L.KS;
    col = lane & 31;
    n = nt * L.tr + col;
    const int kp0 = ks * 32 + (lane >> 5) * 16;
    kh = kp0 / L.rbp; j0 = kp0 - kh * L.rbp;      // (the padded-row layouts; _TAIL goes through chunk_kpos)
    cks = ks; ch = lane >> 5;
  }
  const bool ones_row = (L.layout == QBNN_LAYOUT_MFMA32) && (L.tr == 24 ? col == 24 : ((L.cout & 31) && n == L.cout));
  const bool real_row = col < L.tr && n < L.cout;
  const v4i m4 = L.mu[chunk], s4 = L.sigma[chunk];
  int mw[4] = {m4.x, m4.y, m4.z, m4.w}, sw[4] = {s4.x, s4.y, s4.z, s4.w};
  const int64_t total = (int64_t)L.cout * L.K;
  if (!UNALIGNED && L.layout == QBNN_LAYOUT_MFMA32 && ((L.K | L.krow) & 3) == 0) {
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
    // the chunk's four Philox blocks: element index and validity of each (consecutive blocks in the padded-row layouts; in the tail
    // k-step of _TAIL they belong to different kernel rows)
    uint32_t blk_of[4]; bool blk_ok[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int bkh, bj; bool bv;
      chunk_kpos(cks, ch, 4 * g, L.rbp, L.krow, L.K / L.krow, L.tail_f, bkh, bj, bv);
      blk_of[g] = (uint32_t)(((int64_t)n * L.K + bkh * L.krow + bj) >> 2);
      blk_ok[g] = bv;
    }
#pragma unroll 1
    for (int ss = 0; ss < QBNN_SAMPLER_NS; ++ss) {
      const int s = s0 + ss;
      if (s >= n_samples) break;
      uint32_t ow[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (ones_row && blk_ok[g]) ow[g] = 0x01010101u;
        if (real_row && blk_ok[g]) {
          const qbnn::u32x4 r4 = qbnn::philox4x32_10(blk_of[g], L.layer_id, sample_begin + s, 0u, seed_lo, seed_hi);
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
  if constexpr (UNALIGNED) {
    // Any K / kernel-row length (layers.0: K = 27; LeNet's 2450-wide Linear): the chunk's 16 weights are still CONSECUTIVE elements
    // idx0 .. idx0 + 15 of the reference's order (a chunk never straddles a padded kernel row), only not aligned to a Philox block:
    // five blocks cover them, and a lane picks its 16 words at offset idx0 & 3.  Uniform control flow -- the element-wise form below
    // re-draws a block wherever ANY lane of the wave crosses one, 8 - 16 times per chunk when the rows' alignments differ.  Same
    // fp32 chain as the aligned path; padding bytes (row >= cout, byte >= krow) stay 0 (1 in the ones row).
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
    const int sh = (int)(idx0 & 3);
    const int nvalid = real_row ? min(max(L.krow - j0, 0), 16) : 0;       // valid bytes of the chunk: 0 .. nvalid - 1
    const int nones = ones_row ? min(max(L.krow - j0, 0), 16) : 0;
#pragma unroll 1
    for (int ss = 0; ss < QBNN_SAMPLER_NS; ++ss) {
      const int s = s0 + ss;
      if (s >= n_samples) break;
      uint32_t W[20];
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        const qbnn::u32x4 r4 = qbnn::philox4x32_10((uint32_t)((idx0 >> 2) + t), L.layer_id, sample_begin + s, 0u, seed_lo, seed_hi);
        W[4 * t] = r4.x; W[4 * t + 1] = r4.y; W[4 * t + 2] = r4.z; W[4 * t + 3] = r4.w;
      }
      uint32_t ow[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int j = 4 * g + i;
          const uint32_t ua = (sh & 1) ? W[j + 1] : W[j], ub = (sh & 1) ? W[j + 3] : W[j + 2];
          const float ef = (float)eps_q_from_u32((sh & 2) ? ub : ua, eps_tab);
          const float tq = __builtin_rintf(med3f((sg[j] * ef) * P.mul_multiplier, tlo, thi));
          const float dt = __builtin_fmaf(P.s_mul, tq, dl_mul);
          const float fv = (med3f((dw[j] + dt) * P.inv_s_add, wlo, whi) + QBNN_MAGIC) + zaf;
          f[i] = j < nvalid ? fv : (j < nones ? 1.0f + QBNN_MAGIC : QBNN_MAGIC);      // low byte: the weight, the ones row's 1, or 0
        }
        ow[g] = pack_low_bytes(f[0], f[1], f[2], f[3]);
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
        int ekh, ej; bool ev;
        chunk_kpos(cks, ch, j, L.rbp, L.krow, L.K / L.krow, L.tail_f, ekh, ej, ev);
        valid = real_row && ev;
        idx = (int64_t)n * L.K + ekh * L.krow + ej;
        if (ones_row && ev) ow[j >> 2] |= 1u << (8 * (j & 3));
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
  SamplerTable t, tu;               // tu: large MFMA32 layers whose chunks are not aligned to Philox blocks -> their own kernel
  memset(&t, 0, sizeof(t));
  memset(&tu, 0, sizeof(tu));
  int blocks = 0, blocks_u = 0;
  for (int i = 0; i < n_layers; ++i) {
    const qbnn_sampler_layer& q = layers[i];
    if (!q.mu_packed || !q.sigma_packed || !q.w_out || q.cout <= 0 || q.k <= 0)
      return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8_multi: bad layer entry%s");
    const bool frag = is_mfma_layout(q.layout);
    if (!frag && q.layout != QBNN_LAYOUT_ROWMAJOR) return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8_multi: unknown layout%s");
    if (frag && (q.krow <= 0 || q.k % q.krow))
      return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8_multi: k must be a multiple of krow%s");
    if (layout_n24(q.layout) && q.cout % 24) return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8_multi: the N24 layout takes cout % 24 == 0%s");
    if (layout_tail(q.layout) && !tail_layout_ok(q.k, q.krow)) return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8_multi: the TAIL layout takes ragged kernel rows whose tails fit one k-step%s");
    const size_t bytes = qbnn_packed_weight_bytes(q.cout, q.k, q.krow, q.layout);
    if ((size_t)q.w_sample_stride < bytes || (q.w_sample_stride & 15))
      return fail(QBNN_E_INVALID, "qbnn_sample_weights_i8_multi: w_sample_stride too small or unaligned%s");
    const PackGeom g = pack_geom(q.cout, q.k, frag ? q.krow : q.k, q.layout);
    const bool unaligned = frag && ((q.k | g.krow) & 3) != 0 && bytes / 16 >= 4096;
    SamplerTable& tt = unaligned ? tu : t;
    int& bb = unaligned ? blocks_u : blocks;
    SamplerLayer& L = tt.l[tt.n++];
    L.mu = (const v4i*)q.mu_packed; L.sigma = (const v4i*)q.sigma_packed; L.out = q.w_out; L.out_ss = q.w_sample_stride;
    L.cout = q.cout; L.K = q.k; L.krow = g.krow; L.rbp = g.rbp; L.KS = g.KS; L.layout = frag ? QBNN_LAYOUT_MFMA32 : q.layout; L.tr = g.tr; L.tail_f = g.tail_f;
    L.n_chunks = (int)(bytes / 16); L.chunk_begin = bb; L.layer_id = q.layer_id; L.p = q.params;
    bb += ceil_div(L.n_chunks, 256);
  }
  const dim3 gy(1, ceil_div(n_samples, QBNN_SAMPLER_NS));
  if (t.n)
    hipLaunchKernelGGL(sample_weights_multi_kernel<false>, dim3(blocks, gy.y), dim3(256), 0, (hipStream_t)stream, t, (uint32_t)seed, (uint32_t)(seed >> 32),
                       sample_begin, n_samples, g_noise_dev);
  if (tu.n)
    hipLaunchKernelGGL(sample_weights_multi_kernel<true>, dim3(blocks_u, gy.y), dim3(256), 0, (hipStream_t)stream, tu, (uint32_t)seed, (uint32_t)(seed >> 32),
                       sample_begin, n_samples, g_noise_dev);
  return check_launch("qbnn_sample_weights_i8_multi");
}

// ---- single-conv kernel (layer-level C ABI entry) ---------------------------------------------------------------

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
    EpiDenseDrop<C::COUT, C::HO * C::HO, HAS_RES> epi{outb, a.p, a.a, a.post, {mq_lds, 0.f}};
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
