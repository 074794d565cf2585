// libqbnn_hip.so -- the QAT evaluation's 3 x 3 convs on the int8 matrix pipe, LDS-tiled (round 6).
//
// Reference: quantized/conv_qat.py:139-167 in eval mode -- Z = conv(FQ(X), FQ(W)), Z / scale_factor (+ bias), BatchNorm, (ReLU), then the
// activation FakeQuantize whose observer needs Z's (min, max).  Both operands are integers on a per-sample grid (qbnn_q8.h), so the conv is the
// exact integer sum  s_x s_w * (sum m_x q_w - z_w sum m_x)  on v_mfma_i32_32x32x32_i8, scaled once in fp64 and followed by the fp32 conv's
// own tail -- the arithmetic of conv2d_q8v_kernel (qbnn_f32.hip), bit for bit; what changes is how the operands reach the matrix pipe.
//
// The gather forms move every input byte nine times (once per tap) from L2 in 4-byte units, 64 bytes of K per barrier pair: two MFMAs per
// wave between barriers, 2.3 % of the int8 peak (round 5: the convs were 2.7 of the 6.9 ms of a 10-sample pass).  Here a workgroup stages
//   * an input tile WITH its halo in LDS once per pixel block -- G images x RI rows x (W + 2) pixels x Cin bytes, NHWC, so the KW Cin bytes of a
//     kernel row are contiguous for every output pixel (the layout idea of the int8 product kernels: a k-step is 32 contiguous bytes of that
//     window; the row is padded to a multiple of 32 with ZERO WEIGHTS, whatever activation bytes the overrun reads),
//   * the sample's weights as [channel][k-step][32 B] rows (pitch + 16 B: conflict-free ds_read_b128), whole (NCHUNK = 1: K <= 432) or one
//     kernel row / half row per chunk (Cin = 96 / 192),
// and each wave owns one 32-pixel tile x NTW channel tiles: a pixel fragment is read once per k-step for NTW MFMAs.  The window sum sum m_x (the
// weights' zero point is not 0) comes from v_dot4 on the same fragments, the padded bytes masked.  HBM: the int8 input once, the fp32 output once.
#include <math.h>
#include <stdlib.h>
#include <type_traits>

#include "../../include/qbnn.h"
#include "qbnn_common.h"
#include "qbnn_q8.h"

namespace {

typedef float v4f_q8 __attribute__((ext_vector_type(4)));

// Workgroup barrier that orders LDS traffic only (qbnn_conv.h: lds_barrier): __syncthreads() also waits for every vector-memory operation of the wave --
// here the previous block's 16-byte output stores and the next block's input / the next chunk's weights on their way into registers, which nothing
// behind the barrier depends on (a register filled by a load is waited for where it is used).
__device__ __forceinline__ void q8t_lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// One accumulator tile (this lane: pixel `lane & 31`, channels nb0 + 8 g + 4 h + {0..3}) through conv2d_q8v_kernel's tail, step for step:
// fl32(fl64(N - z_w R) * fl64(s_x s_w)), / div, + bias, * alpha, + beta, ReLU; float4 stores; running (min, max) of what was stored.
__device__ __forceinline__ void q8_tail_tile(const ConvQ8Args& a, const v16i_q8& acc, int zwr, double sp, int nb0, int h, float* yp, float& vmin, float& vmax) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int nb = nb0 + 8 * g + 4 * h;
    const bool full = nb < a.Cout;          // Cout % 4 == 0 (the launcher checks): this lane's four channels exist together, as 16-byte aligned float4s
    if (full) {                             // (a padded channel group -- 24 .. 31 of a 24-channel layer -- costs nothing: a quarter of that layer's tail)
      v4f_q8 pd{1.f, 1.f, 1.f, 1.f}, pb{0.f, 0.f, 0.f, 0.f}, pa = pb, pe = pb;
      if (a.div) pd = *reinterpret_cast<const v4f_q8*>(a.div + nb);
      if (a.bias) pb = *reinterpret_cast<const v4f_q8*>(a.bias + nb);
      if (a.alpha) pa = *reinterpret_cast<const v4f_q8*>(a.alpha + nb);
      if (a.beta) pe = *reinterpret_cast<const v4f_q8*>(a.beta + nb);
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float tt = (float)((double)(acc[4 * g + i] - zwr) * sp);
        if (a.div) tt = tt / pd[i];
        if (a.bias) tt = tt + pb[i];
        if (a.alpha) tt = tt * pa[i];
        if (a.beta) tt = tt + pe[i];
        if (a.relu) tt = fmaxf(tt, 0.f);
        v[i] = tt;
        vmin = fminf(vmin, tt); vmax = fmaxf(vmax, tt);
      }
      *reinterpret_cast<v4f_q8*>(yp + nb) = v4f_q8{v[0], v[1], v[2], v[3]};
    }
    __builtin_amdgcn_sched_barrier(0);      // one channel group at a time (the unrolled tail otherwise loads every group's parameters up front)
  }
}

// the workgroup's (min, max) of everything it stored -> its slot of the observer's partials
__device__ __forceinline__ void q8_write_partials(const ConvQ8Args& a, float vmin, float vmax, float* red, int s) {
  if (!a.mm_partials) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

template <int CIN_, int STRIDE_, int WO_, int G_, int ROWS_, int NTW_, int CW_, int NCHUNK_, int IT_, int WPE_>
struct Q8TCfg {
  static constexpr int WPE = WPE_;            // waves per SIMD the kernel is compiled for (= workgroups per CU: 512 / WPE registers, 160 KiB / WPE of LDS)
  static constexpr int CIN = CIN_, STRIDE = STRIDE_, WO = WO_, HO = WO_, G = G_, ROWS = ROWS_, NTW = NTW_, CW = CW_, NCHUNK = NCHUNK_, IT = IT_;
  static constexpr int PIX = G * ROWS * WO, PW = PIX / 32;             // output pixels / 32-pixel tiles per block
  static_assert(PIX % 32 == 0 && PW * CW == 4, "four waves: PW pixel tiles x CW channel groups");
  static_assert(G == 1 || ROWS == HO, "several images per block only as whole images");
  static_assert(HO % ROWS == 0, "whole row blocks per image");
  static constexpr int W_IN = WO * STRIDE, H_IN = W_IN, CI = W_IN + 2, RI = (ROWS - 1) * STRIDE + 3;      // tile: RI rows x CI pixels (halo 1)
  static constexpr int ROWB = 3 * CIN, ROWPAD = (ROWB + 31) / 32 * 32, RSTEPS = ROWPAD / 32;               // a kernel row: bytes, padded, k-steps
  static constexpr int U = (CIN % 16 == 0) ? 16 : 8;                                                        // staging unit (bytes)
  static_assert(CIN % 8 == 0 && ROWB % U == 0, "kernel rows are whole staging units");
  static constexpr int TILE = G * RI * CI * CIN, TILE_LDS = (TILE + 64 + 15) / 16 * 16;                    // + slack: the last window's padded k-step
  static constexpr int COUT_WG = 32 * NTW * CW;
  static_assert(NCHUNK == 1 || NCHUNK == 3 || (NCHUNK == 6 && RSTEPS % 2 == 0 && ROWB == ROWPAD), "whole conv, kernel row, or half row per chunk");
  static constexpr int STEPS = NCHUNK == 1 ? 3 * RSTEPS : (NCHUNK == 3 ? RSTEPS : RSTEPS / 2);              // k-steps per chunk
  static constexpr int WPITCH = STEPS * 32 + 16;
  static constexpr int W_LDS = COUT_WG * WPITCH;
  static constexpr int LDS = TILE_LDS + W_LDS;
  static constexpr int BPI = HO / ROWS;                                                                     // blocks per image (G = 1)
  static_assert(LDS * WPE <= 160 * 1024, "LDS budget of WPE workgroups per CU");
};

// (HIP's second launch-bounds argument: minimum waves per execution unit -- without it the fully unrolled k loop takes all 512 registers)
template <class C>
__global__ __launch_bounds__(256, C::WPE) void conv2d_q8t_kernel(const ConvQ8Args a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ float red[8];
  uint8_t* tile = smem;
  uint8_t* wl = smem + C::TILE_LDS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pt = wave % C::PW, cw = wave / C::PW;
  const int s = blockIdx.z, cg = blockIdx.y;
  constexpr int K = 9 * C::CIN;
  const int8_t* xs = a.x + (int64_t)s * a.x_ss;
  const int8_t* ws = a.w + (int64_t)s * a.w_ss;
  const int nblocks = C::G == 1 ? a.B * C::BPI : (a.B + C::G - 1) / C::G;
  const int b0 = blockIdx.x * C::IT;

  for (int i = tid; i < C::TILE_LDS / 16; i += 256) reinterpret_cast<v4i_q8*>(tile)[i] = v4i_q8{0, 0, 0, 0};      // halo columns + slack stay 0

  // this lane's output pixel inside a block, its window's tap (0, 0) in the tile, its weight row
  const int q = pt * 32 + (lane & 31), h = lane >> 5;
  const int qg = q / (C::ROWS * C::WO), qr = (q / C::WO) % C::ROWS, qc = q % C::WO;
  const int boff = ((qg * C::RI + qr * C::STRIDE) * C::CI + qc * C::STRIDE) * C::CIN + 16 * h;
  const int aoff = (lane & 31) * C::WPITCH + 16 * h;
  const double sp = (double)a.s_x[s] * (double)a.s_w[s];
  const int zw = a.z_w[s];
  float vmin = INFINITY, vmax = -INFINITY;

  auto stage_w = [&](int ch) {
    constexpr int ROWS_CH = C::NCHUNK == 1 ? 3 : 1;
    constexpr int SEG = C::NCHUNK == 6 ? C::ROWPAD / 2 : C::ROWPAD;             // LDS bytes per (channel, kernel row) of a chunk
    constexpr int SEG_VALID = C::NCHUNK == 6 ? C::ROWB / 2 : C::ROWB;           // ... of which come from the weights (the rest: zeros)
    constexpr int UPS = SEG / C::U;
    for (int u = tid; u < C::COUT_WG * ROWS_CH * UPS; u += 256) {
      const int n = u / (ROWS_CH * UPS), rr = (u / UPS) % ROWS_CH, j = u % UPS;
      const int kh = C::NCHUNK == 1 ? rr : (C::NCHUNK == 3 ? ch : ch / 2);
      const int srcoff = (C::NCHUNK == 6 ? (ch % 2) * SEG_VALID : 0) + j * C::U;
      const int ng = cg * C::COUT_WG + n;
      const bool ok = j * C::U < SEG_VALID && ng < a.Cout;
      const int8_t* src = ws + (int64_t)(ok ? ng : 0) * K + kh * C::ROWB + (ok ? srcoff : 0);
      uint8_t* dst = wl + n * C::WPITCH + rr * C::ROWPAD + j * C::U;
      if constexpr (C::U == 16) {
        v4i_q8 v = *reinterpret_cast<const v4i_q8*>(src);
        if (!ok) v = v4i_q8{0, 0, 0, 0};
        *reinterpret_cast<v4i_q8*>(dst) = v;
      } else {
        v2i_q8 v = *reinterpret_cast<const v2i_q8*>(src);
        if (!ok) v = v2i_q8{0, 0};
        *reinterpret_cast<v2i_q8*>(dst) = v;
      }
    }
  };

  // Chunked weights (Cin = 96 / 192): chunk ch + 1 is fetched into registers while chunk ch multiplies, so a chunk's L2 latency hides under the
  // previous chunk's MFMAs instead of heading every chunk behind a barrier (the 4 x 4 x 192 layer staged six chunks back to back: 113 us per launch).
  constexpr int WCH_ROWS = C::NCHUNK == 1 ? 3 : 1;
  constexpr int WCH_SEG = C::NCHUNK == 6 ? C::ROWPAD / 2 : C::ROWPAD;
  constexpr int WCH_UPS = WCH_SEG / C::U;
  constexpr int WCH_UNITS = C::COUT_WG * WCH_ROWS * WCH_UPS, WCH_PER = (WCH_UNITS + 255) / 256;
  v4i_q8 wreg[C::NCHUNK > 1 ? WCH_PER : 1];
  auto load_w = [&](int ch) {
    static_assert(C::NCHUNK == 1 || C::U == 16, "chunked weights move 16-byte units");
#pragma unroll
    for (int k = 0; k < WCH_PER; ++k) {
      const int u = tid + 256 * k;
      const int n = u / WCH_UPS, j = u % WCH_UPS;
      const int kh = C::NCHUNK == 3 ? ch : ch / 2;
      const int seg_valid = C::NCHUNK == 6 ? C::ROWB / 2 : C::ROWB;
      const int srcoff = (C::NCHUNK == 6 ? (ch % 2) * seg_valid : 0) + j * C::U;
      const int ng = cg * C::COUT_WG + n;
      const bool ok = u < WCH_UNITS && j * C::U < seg_valid && ng < a.Cout;
      v4i_q8 v = *reinterpret_cast<const v4i_q8*>(ws + (int64_t)(ok ? ng : 0) * K + kh * C::ROWB + (ok ? srcoff : 0));
      if (!ok) v = v4i_q8{0, 0, 0, 0};
      wreg[k] = v;
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int k = 0; k < WCH_PER; ++k) {
      const int u = tid + 256 * k;
      if (u < WCH_UNITS) *reinterpret_cast<v4i_q8*>(wl + (u / WCH_UPS) * C::WPITCH + (u % WCH_UPS) * C::U) = wreg[k];
    }
  };

  // A block's input tile goes global -> registers -> LDS: the loads of block it + 1 are issued right after block it's tile is in LDS, so their latency
  // hides under a whole block of MFMAs and tail arithmetic instead of standing between two barriers (IT > 1: the 24- and 48-channel layers).
  constexpr int T_UPR = C::W_IN * C::CIN / C::U;               // staging units per input row
  constexpr int T_UNITS = C::G * C::RI * T_UPR, T_PER = (T_UNITS + 255) / 256;
  typedef typename std::conditional<C::U == 16, v4i_q8, v2i_q8>::type tunit;
  tunit treg[T_PER];
  auto load_tile = [&](int blk) {
    const int img0 = C::G == 1 ? blk / C::BPI : blk * C::G;
    const int oh0 = C::G == 1 ? (blk % C::BPI) * C::ROWS : 0;
#pragma unroll
    for (int k = 0; k < T_PER; ++k) {
      const int u = tid + 256 * k;
      const int g = u / (C::RI * T_UPR), r = (u / T_UPR) % C::RI, j = u % T_UPR;
      const int ih = oh0 * C::STRIDE - 1 + r, img = img0 + g;
      const bool ok = u < T_UNITS && (unsigned)ih < (unsigned)C::H_IN && img < a.B;      // rows above / below the map, images beyond a ragged batch: m_x = 0
      tunit v = *reinterpret_cast<const tunit*>(xs + (((int64_t)(ok ? img : 0) * C::H_IN + (ok ? ih : 0)) * C::W_IN) * C::CIN + (ok ? j : 0) * C::U);
      if (!ok) v = tunit{};
      treg[k] = v;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int k = 0; k < T_PER; ++k) {
      const int u = tid + 256 * k;
      const int g = u / (C::RI * T_UPR), r = (u / T_UPR) % C::RI, j = u % T_UPR;
      if (u < T_UNITS) *reinterpret_cast<tunit*>(tile + ((g * C::RI + r) * C::CI + 1) * C::CIN + j * C::U) = treg[k];
    }
  };
  if (b0 < nblocks) load_tile(b0);

  for (int it = 0; it < C::IT; ++it) {
    const int blk = b0 + it;
    if (blk >= nblocks) break;                                  // workgroup-uniform
    const int img0 = C::G == 1 ? blk / C::BPI : blk * C::G;
    const int oh0 = C::G == 1 ? (blk % C::BPI) * C::ROWS : 0;
    q8t_lds_barrier();                                            // the previous block's fragment reads (first block: the zero fill) are done
    store_tile();
    if (C::IT > 1 && it + 1 < C::IT && blk + 1 < nblocks) load_tile(blk + 1);      // the next block's input: in flight under this block's MFMAs and tail
    if (C::NCHUNK == 1 && it == 0) stage_w(0);                 // the whole conv's weights: once per workgroup
    if constexpr (C::NCHUNK > 1) load_w(0);

    v16i_q8 acc[C::NTW];
#pragma unroll
    for (int j = 0; j < C::NTW; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[j][i] = 0;
    int rsum = 0;
#pragma unroll
    for (int ch = 0; ch < C::NCHUNK; ++ch) {
      if constexpr (C::NCHUNK > 1) {
        if (ch > 0) q8t_lds_barrier();                            // the previous chunk's weight reads are done
        store_w();
        if (ch + 1 < C::NCHUNK) load_w(ch + 1);                 // in flight under this chunk's MFMAs
      }
      q8t_lds_barrier();
      // One k-step ahead: the fragments of step i + 1 are requested before the MFMAs of step i (a scheduling fence per step keeps the compiler
      // from hoisting every LDS read of the unrolled chunk to its top -- 500+ registers, or spills under the occupancy bound).
      v4i_q8 bq[2], aq[2][C::NTW];
      auto fetch = [&](int i, int slot) {
        const int kh = C::NCHUNK == 1 ? i / C::RSTEPS : (C::NCHUNK == 3 ? ch : ch / 2);
        const int t = C::NCHUNK == 1 ? i % C::RSTEPS : (C::NCHUNK == 3 ? i : (ch % 2) * C::STEPS + i);
        const uint8_t* bp = tile + boff + kh * C::CI * C::CIN + 32 * t;
        if constexpr (C::U == 16) bq[slot] = *reinterpret_cast<const v4i_q8*>(bp);
        else {
          const v2i_q8 lo = *reinterpret_cast<const v2i_q8*>(bp), hi = *reinterpret_cast<const v2i_q8*>(bp + 8);
          bq[slot] = v4i_q8{lo.x, lo.y, hi.x, hi.y};
        }
#pragma unroll
        for (int j = 0; j < C::NTW; ++j) aq[slot][j] = *reinterpret_cast<const v4i_q8*>(wl + aoff + (cw + C::CW * j) * 32 * C::WPITCH + 32 * i);
      };
      fetch(0, 0);
#pragma unroll
      for (int i = 0; i < C::STEPS; ++i) {
        const int t = C::NCHUNK == 1 ? i % C::RSTEPS : (C::NCHUNK == 3 ? i : (ch % 2) * C::STEPS + i);
        if (i + 1 < C::STEPS) fetch(i + 1, (i + 1) & 1);
        const v4i_q8 bv = bq[i & 1];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          int v = bv[d];
          if (32 * t + 32 > C::ROWB) v = (32 * t + 16 * h + 4 * d + 4 <= C::ROWB) ? v : 0;      // the row's padded tail reads the next pixels: not in the window
          rsum = __builtin_amdgcn_sdot4(v, 0x01010101, rsum, false);
        }
#pragma unroll
        for (int j = 0; j < C::NTW; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq[i & 1][j], bv, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // epilogue: conv2d_q8v_kernel's, step for step
    const int R = rsum + __shfl_xor(rsum, 32);
    const int img = img0 + qg;
    if (img < a.B) {
      const int64_t po = ((int64_t)img * C::HO + oh0 + qr) * C::WO + qc;
      const int zwr = zw * R;
      float* yp = a.y + (int64_t)s * a.y_ss + po * a.Cout;
#pragma unroll
      for (int j = 0; j < C::NTW; ++j) q8_tail_tile(a, acc[j], zwr, sp, cg * C::COUT_WG + (cw + C::CW * j) * 32, h, yp, vmin, vmax);
    }
  }
  q8_write_partials(a, vmin, vmax, red, s);
}

// layers.0: 3 -> Cout <= 32 channels on a 32 x 32 map (K = 27: one k-step).  A 3-channel pixel is 3 bytes, so nothing about a window is aligned;
// the workgroup stages 6 input rows (96 B each, at a 16-byte aligned interior offset of a 128-byte pitch, halo bytes zero), every thread then
// assembles half of one output pixel's 27-byte patch (+ 5 zeros) from LDS bytes into a [128 pixels][32 B] operand matrix -- the im2col the
// int8 product path does once per batch (qbnn_im2col3x3_c3), here per block in LDS because every MC sample has its own input grid.
struct T_C3 { static constexpr int IT = 8, ROWS = 4, HW = 32, PITCH = 128, IN0 = 16, RI = 6, PP = 48, WPITCH = 48; };
__global__ __launch_bounds__(256, 4) void conv2d_q8_c3_kernel(const ConvQ8Args a) {
  using C = T_C3;
  __shared__ __attribute__((aligned(16))) uint8_t tile[C::RI * C::PITCH];
  __shared__ __attribute__((aligned(16))) uint8_t pm[128 * C::PP];          // patches [pixel][32 B] (pitch 48)
  __shared__ __attribute__((aligned(16))) uint8_t wl[32 * C::WPITCH];        // weights [channel][32 B]
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int s = blockIdx.z;
  const int8_t* xs = a.x + (int64_t)s * a.x_ss;
  const int8_t* ws = a.w + (int64_t)s * a.w_ss;
  const int nblocks = a.B * (C::HW / C::ROWS), b0 = blockIdx.x * C::IT;
  for (int i = tid; i < C::RI * C::PITCH / 4; i += 256) reinterpret_cast<int*>(tile)[i] = 0;
  for (int i = tid; i < 32 * C::WPITCH / 4; i += 256) reinterpret_cast<int*>(wl)[i] = 0;
  q8t_lds_barrier();
  for (int i = tid; i < 32 * 27; i += 256) {
    const int n = i / 27, k = i - n * 27;
    if (n < a.Cout) wl[n * C::WPITCH + k] = (uint8_t)ws[n * 27 + k];
  }
  const double sp = (double)a.s_x[s] * (double)a.s_w[s];
  const int zw = a.z_w[s];
  float vmin = INFINITY, vmax = -INFINITY;
  for (int it = 0; it < C::IT; ++it) {
    const int blk = b0 + it;
    if (blk >= nblocks) break;
    const int img = blk / (C::HW / C::ROWS), oh0 = (blk % (C::HW / C::ROWS)) * C::ROWS;
    q8t_lds_barrier();                              // the previous block's patch assembly is done with the tile (first block: zero fill, weights)
    if (tid < C::RI * 6) {                        // 6 rows x 6 units of 16 B
      const int r = tid / 6, j = tid - r * 6, ih = oh0 - 1 + r;
      const bool ok = (unsigned)ih < (unsigned)C::HW;
      v4i_q8 v = *reinterpret_cast<const v4i_q8*>(xs + (((int64_t)img * C::HW + (ok ? ih : 0)) * C::HW) * 3 + j * 16);
      if (!ok) v = v4i_q8{0, 0, 0, 0};
      *reinterpret_cast<v4i_q8*>(tile + r * C::PITCH + C::IN0 + j * 16) = v;
    }
    q8t_lds_barrier();
    {                                             // patch bytes [16 part, 16 part + 16) of pixel q: k = kh 9 + kw 3 + c  <-  tile[(r + kh)][IN0 - 3 + 3 c0 + (k - 9 kh)]
      const int q = tid & 127, part = tid >> 7, r = q >> 5, c0 = q & 31;
      uint32_t wds[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int b = 0; b < 16; ++b) {
        const int k = 16 * part + b;
        if (k < 27) {
          const int kh = k / 9, j = k - 9 * kh;
          wds[b >> 2] |= (uint32_t)tile[(r + kh) * C::PITCH + C::IN0 - 3 + 3 * c0 + j] << (8 * (b & 3));
        }
      }
      *reinterpret_cast<v4i_q8*>(pm + q * C::PP + 16 * part) = v4i_q8{(int)wds[0], (int)wds[1], (int)wds[2], (int)wds[3]};
    }
    q8t_lds_barrier();
    const v4i_q8 bv = *reinterpret_cast<const v4i_q8*>(pm + (wave * 32 + (lane & 31)) * C::PP + 16 * h);
    const v4i_q8 av = *reinterpret_cast<const v4i_q8*>(wl + (lane & 31) * C::WPITCH + 16 * h);
    v16i_q8 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0;
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc, 0, 0, 0);
    int rsum = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) rsum = __builtin_amdgcn_sdot4(bv[d], 0x01010101, rsum, false);      // (the pad bytes are zeros)
    const int R = rsum + __shfl_xor(rsum, 32);
    const int64_t po = ((int64_t)img * C::HW + oh0 + wave) * C::HW + (lane & 31);
    q8_tail_tile(a, acc, zw * R, sp, 0, h, a.y + (int64_t)s * a.y_ss + po * a.Cout, vmin, vmax);
  }
  q8_write_partials(a, vmin, vmax, red, s);
}

//                   Cin  s  Wo  G rows NTW CW chunks IT WPE
using T_L1 = Q8TCfg<24, 1, 32, 1, 4, 1, 1, 1, 8, 5>;        // 32 x 32 x 24 -> 24 (.. 32 channels): one image per workgroup
using T_D24 = Q8TCfg<24, 2, 16, 1, 8, 2, 1, 1, 2, 3>;       // 32 x 32 x 24 -> 16 x 16 x 48 (.. 64)
using T_C48 = Q8TCfg<48, 1, 16, 1, 8, 2, 1, 1, 2, 3>;       // 16 x 16 x 48 -> 48 (.. 64)
using T_D48 = Q8TCfg<48, 2, 8, 2, 8, 3, 1, 1, 1, 2>;        // 16 x 16 x 48 -> 8 x 8 x 96
using T_C96 = Q8TCfg<96, 1, 8, 2, 8, 3, 1, 3, 1, 2>;        // 8 x 8 x 96 -> 96
using T_D96 = Q8TCfg<96, 2, 4, 8, 4, 3, 1, 3, 1, 1>;        // 8 x 8 x 96 -> 4 x 4 x 192: two channel groups of 96
using T_C192 = Q8TCfg<192, 1, 4, 8, 4, 3, 1, 6, 1, 1>;      // 4 x 4 x 192 -> 192: two channel groups of 96

bool tiled_on() {
  static const bool on = [] { const char* e = getenv("QBNN_Q8_TILED"); return !(e && e[0] == '0'); }();
  return on;
}

template <class C> bool match(int H, int W, int Cin, int stride) { return H == C::H_IN && W == C::W_IN && Cin == C::CIN && stride == C::STRIDE; }
template <class C> void grid_of(int B, int Cout, int& gx, int& gy) {
  const int nblocks = C::G == 1 ? B * C::BPI : (B + C::G - 1) / C::G;
  gx = (nblocks + C::IT - 1) / C::IT;
  gy = (Cout + C::COUT_WG - 1) / C::COUT_WG;
}
template <class C> int launch(const ConvQ8Args& a, int n_samples, hipStream_t st) {
  static std::atomic<uint64_t> attr{0};
  if (int rc = qbnn_ensure_dyn_lds((const void*)conv2d_q8t_kernel<C>, &attr, C::LDS)) return rc;
  int gx, gy;
  grid_of<C>(a.B, a.Cout, gx, gy);
  hipLaunchKernelGGL(conv2d_q8t_kernel<C>, dim3(gx, gy, n_samples), dim3(256), C::LDS, st, a);
  return qbnn_check_launch_msg("qbnn_conv2d_q8_f32_mc");
}

#define Q8T_FOR_EACH(X) X(T_L1) X(T_D24) X(T_C48) X(T_D48) X(T_C96) X(T_D96) X(T_C192)

}  // namespace

static bool match_c3(int H, int W, int Cin, int Cout, int stride) { return H == 32 && W == 32 && Cin == 3 && stride == 1 && Cout <= 32; }

int qbnn_conv_q8t_blocks(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int pad) {
  if (!tiled_on() || ksize != 3 || pad != 1 || B <= 0 || Cout <= 0 || (Cout & 3) != 0) return 0;
  if (match_c3(H, W, Cin, Cout, stride)) return (B * (T_C3::HW / T_C3::ROWS) + T_C3::IT - 1) / T_C3::IT;
  int gx = 0, gy = 0;
#define X(C) if (match<C>(H, W, Cin, stride)) { grid_of<C>(B, Cout, gx, gy); return gx * gy; }
  Q8T_FOR_EACH(X)
#undef X
  return 0;
}

// operand alignment the matching form needs (16-byte staging units; 8-byte ones at Cin = 24; the 3-channel form reads its weights bytewise);
// float4 output / parameter accesses: Cout % 4 == 0 is part of the match, y and y_ss are checked here
bool qbnn_conv_q8t_aligned(const ConvQ8Args& a) {
  auto al = [](const void* p, int n) { return (reinterpret_cast<uintptr_t>(p) & (uintptr_t)(n - 1)) == 0; };
  if (!al(a.y, 16) || (a.y_ss & 3) != 0 || (a.div && !al(a.div, 16)) || (a.bias && !al(a.bias, 16)) || (a.alpha && !al(a.alpha, 16)) || (a.beta && !al(a.beta, 16)))
    return false;
  if (match_c3(a.H, a.W, a.Cin, a.Cout, a.stride)) return al(a.x, 16) && (a.x_ss & 15) == 0;
  const int u = (a.Cin % 16 == 0) ? 16 : 8;
  return al(a.x, u) && al(a.w, u) && (a.x_ss % u) == 0 && (a.w_ss % u) == 0;
}

int qbnn_launch_conv_q8t(const ConvQ8Args& a, int n_samples, hipStream_t st) {
  if (match_c3(a.H, a.W, a.Cin, a.Cout, a.stride)) {
    const int gx = (a.B * (T_C3::HW / T_C3::ROWS) + T_C3::IT - 1) / T_C3::IT;
    hipLaunchKernelGGL(conv2d_q8_c3_kernel, dim3(gx, 1, n_samples), dim3(256), 0, st, a);
    return qbnn_check_launch_msg("qbnn_conv2d_q8_f32_mc");
  }
#define X(C) if (match<C>(a.H, a.W, a.Cin, a.stride)) return launch<C>(a, n_samples, st);
  Q8T_FOR_EACH(X)
#undef X
  return qbnn_fail_msg(QBNN_E_INVALID, "qbnn_conv2d_q8_f32_mc: no tiled form for this geometry");
}
