// Device code shared by the int8 conv kernels of libqbnn_hip.so: the conv geometry (ConvCfg), MFMA pass loops, requantising
// epilogue functors, LDS-DMA helpers and the argument blocks of the fused BasicBlock kernels.  Header-only (templates and
// __forceinline__ device functions); included by qbnn_kernels.hip, qbnn_blocks.hip, qbnn_w16.hip and qbnn_misc.hip.
#ifndef QBNN_CONV_H_
#define QBNN_CONV_H_
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/qbnn.h"
#include "qbnn_rng.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v2iperm __attribute__((ext_vector_type(2)));

// Cross-half exchanges of a wave in ONE vector instruction (v_permlane32_swap, gfx950): no LDS round trip (ds_bpermute, which
// __shfl_xor(v, 32) compiles to, is ~100+ cycles of latency at the head of every epilogue's dependent chain).
// v_permlane32_swap vdst, src: swaps vdst[lanes 32..63] with src[lanes 0..31]; with both operands = v the first result holds
// v's lower-half value in both halves, the second its upper-half value.
__device__ __forceinline__ int half_lo_bcast(int v) {            // lane l gets v of lane l & 31
  const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  return (int)r[0];
}
__device__ __forceinline__ int half_sum(int v) {                 // lane l gets v(l & 31) + v((l & 31) + 32)
  const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  return (int)r[0] + (int)r[1];
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// rne of an fp32 that may be far outside the int range (clamp first: monotone, so the later
// integer clamp gives the same result as the reference's saturating conversion)
__device__ __forceinline__ int rne_sat(float v) {
  v = fminf(fmaxf(v, -1.0e9f), 1.0e9f);
  return __float2int_rn(v);
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
// buffer that nothing else reads.  The shipped library contains none of this.  (Per translation unit; the fused block
// kernels of qbnn_blocks.hip are the ones that are stamped, and that file exports the buffer setters.)
#ifdef QBNN_STAMP
static __device__ unsigned long long* g_stamp_dev_ptr = nullptr;
#define g_stamp_dev g_stamp_dev_ptr
static __device__ unsigned long long g_inner[4];

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
        const int zwr = p.z_w * half_lo_bcast(rv);
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
        R = half_lo_bcast(rv);
      } else {
        R = half_sum(rsum[mb]);
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

// ---- the explicit form of the LDS-DMA ring (qbnn_down_ring.hip, qbnn_chain_ring.hip): NBUF slabs, NBUF - 1 in flight, per-wave vmcnt accounting
// 16 bytes per lane, global -> LDS at lds_addr + 16 * lane, without passing through registers.  Issued from inline assembly: see the
// header comment (the compiler must not know that LDS is written behind its back; this file does its own vmcnt accounting).
__device__ __forceinline__ void dma16(const void* gptr, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gptr), "s"(lds_addr) : "memory");
}
// The same with the global address as a wave-uniform base (SGPR pair) + a 32-bit per-lane offset: the ring kernels' fragment tiles are
// 1 KiB blocks at uniform addresses, lane l's 16 bytes at + 16 l -- the base moves on the scalar unit and the per-lane offset is ONE
// register for the whole kernel, where the 64-bit per-lane form costs two v_lshl_add_u64 / v_add per request (round 5: 200 vector
// instructions per wave and work item in the 96 -> 192 block, whose vector issue slots are what the kernel is short of).
__device__ __forceinline__ void dma16_s(const void* gbase_uniform, uint32_t lane_off, uint32_t lds_addr) {
  uint32_t keep;
  // (readfirstlane: a no-op where the compiler already knows the base to be uniform, and what makes it an SGPR pair where it does not)
  const uint64_t g = (uint64_t)gbase_uniform;
  const uint64_t gs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(g >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)g);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_off), "s"(gs), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr_of(const uint8_t* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// s_waitcnt vmcnt(K) only (gfx9 encoding: vmcnt = {[15:14], [3:0]}, expcnt [6:4] and lgkmcnt [11:8] left at "no wait")
template <int K> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(K >= 0 && K < 64, "vmcnt is a 6-bit counter");
  __builtin_amdgcn_s_waitcnt((K & 0xf) | ((K >> 4) << 14) | 0x0f70);
}

// The workgroup's weight ring: consumer position (buffer of the slab being multiplied) and producer position (flat index of the next
// slab to request; one item = NSI slabs, conv after conv).
struct WeightRing {
  uint8_t* base; int cbuf, pbuf, pnext;
};


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
  float mq1;                       // quantised value of a kept mask element minus z_m (the bit tables of the fused block kernels)
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

// ---- dropout behind the convs of a fused block (conv_resnet_mc: mcdropout/models_mc.py:116-160) -----------------------------
// argument block of the dropouts of one launch, passed beside the kernel's own arguments (empty for the graphs without dropout)
template <int N> struct DropSet { PostArgs d[N]; };
template <> struct DropSet<0> {};

// Mask table of one dropout for the images of a work item, in LDS: fp32 [G][COUT] (the quantised mask value minus its zero
// point), or -- BITS, where LDS is short -- one bit per (image, channel): a Bernoulli mask has two values, 0 and mq1.
template <int COUT, bool BITS> struct MaskTab {
  static constexpr int WPI = (COUT + 31) / 32;                 // BITS: words per image
  static constexpr int bytes(int G) { return BITS ? G * WPI * 4 : G * COUT * 4; }
  const void* tab; float mq1;
  // mask values of image g, channels c0 .. c0 + 3 (c0 a multiple of 4)
  __device__ __forceinline__ float4 get(int g, int c0) const {
    if constexpr (!BITS) {
      return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(tab) + g * COUT + c0);
    } else {
      const int w = reinterpret_cast<const int*>(tab)[g * WPI + (c0 >> 5)] >> (c0 & 31);
      const int mb = __float_as_int(mq1);           // bit set -> all ones (sign-extended 1-bit field) & mq1, else +0.0
      return float4{__int_as_float(__builtin_amdgcn_sbfe(w, 0, 1) & mb), __int_as_float(__builtin_amdgcn_sbfe(w, 1, 1) & mb),
                    __int_as_float(__builtin_amdgcn_sbfe(w, 2, 1) & mb), __int_as_float(__builtin_amdgcn_sbfe(w, 3, 1) & mb)};
    }
  }
};
// the table of MC sample s, images img0 .. img0 + G - 1 (every thread of the workgroup calls it; a barrier publishes it)
// (a pointer out of a by-value descriptor is generic to the compiler: read through it as it stands it becomes a FLAT load, which returns out of
//  order with the ring kernels' counted global loads -- tests/test_host_logic.py disassembles them for exactly this)
template <class T> __device__ __forceinline__ const T* mask_global(const T* p) {
  return (const T*)reinterpret_cast<const __attribute__((address_space(1))) T*>(reinterpret_cast<uintptr_t>(p));
}
template <int G, int COUT, bool BITS, int NTHR>
__device__ __forceinline__ void fill_mask_tab(void* tab, const PostArgs& q_, int s, int img0, int B, int tid) {
  PostArgs q = q_;
  q.nd = mask_global(q.nd);
  q.mask_in = mask_global(q.mask_in);
  uint32_t seed_lo = q.seed_lo, seed_hi = q.seed_hi, sample_begin = q.sample_begin;
  if (q.nd) { seed_lo = q.nd[0]; seed_hi = q.nd[1]; sample_begin = q.nd[2]; }
  if constexpr (!BITS) {
    for (int i = tid; i < G * COUT; i += NTHR) {
      const int b = img0 + i / COUT;
      reinterpret_cast<float*>(tab)[i] = b < B ? (float)drop_mask_q(b * COUT + i % COUT, s, (int64_t)B * COUT, q.keep, q.inv_sm, q.z_m, seed_lo,
                                                                    seed_hi, q.layer_id, sample_begin, q.mask_in) : 0.f;
    }
  } else {
    constexpr int WPI = MaskTab<COUT, true>::WPI;
    for (int i = tid; i < G * WPI * 32; i += NTHR) {          // one lane per (image, channel), 32 consecutive lanes per word
      const int b = img0 + i / (WPI * 32), c = i % (WPI * 32);
      const bool on = b < B && c < COUT && drop_mask_q(b * COUT + c, s, (int64_t)B * COUT, q.keep, q.inv_sm, q.z_m, seed_lo, seed_hi, q.layer_id,
                                                       sample_begin, q.mask_in) != 0;
      const uint64_t bal = __ballot(on);
      if ((tid & 31) == 0) reinterpret_cast<uint32_t*>(tab)[i >> 5] = (uint32_t)(bal >> (tid & 32));
    }
  }
}
// r' = quantized::mul(conv output, mask) centred on z_m, as an integer-valued float before its rounding (EpiDenseDrop's first lines)
__device__ __forceinline__ float drop_val(float v, const QConv& p, const PostArgs& q, float mm) {
  const float qc = __builtin_rintf(med3f(v, p.vlo, p.vhi));
  return med3f((qc * mm) * q.dmult, q.dlo, q.dhi);
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
template <int COUT, int IMG_PIX, bool HAS_RES, int PITCH = COUT, bool BITS = false>
struct EpiDenseDrop {
  static constexpr int VALU_PER_MFMA = HAS_RES ? 26 : 16;
  uint8_t* outb; QConv p; QAdd a; PostArgs q; MaskTab<COUT, BITS> mt;
  __device__ __forceinline__ int pixel(int m) const { return m * PITCH; }
  __device__ __forceinline__ uint32_t load(int po, int c0) const {
    return HAS_RES ? *reinterpret_cast<const uint32_t*>(outb + po + c0) : 0u;
  }
  __device__ __forceinline__ void store(int po, int c0, float v0, float v1, float v2, float v3, uint32_t rq) const {
    uint32_t* o = reinterpret_cast<uint32_t*>(outb + po + c0);
    const int g = po / (IMG_PIX * PITCH);
    const float4 m4 = mt.get(g, c0);
    const float r[4] = {drop_val(v0, p, q, m4.x), drop_val(v1, p, q, m4.y), drop_val(v2, p, q, m4.z), drop_val(v3, p, q, m4.w)};
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
// RES_U8: the residual bytes are known to be non-negative (a tile written by ReLU-fused epilogues of the same kernel): one
//     v_cvt_f32_ubyteN per value instead of v_bfe_i32 + v_cvt_f32_i32.
template <int HO, int PIXB, int TILE_BYTES, bool RES_U8 = false>
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
    const float rf[4] = {RES_U8 ? (float)(rqu & 0xffu) : (float)((rq << 24) >> 24), RES_U8 ? (float)((rqu >> 8) & 0xffu) : (float)((rq << 16) >> 24),
                         RES_U8 ? (float)((rqu >> 16) & 0xffu) : (float)((rq << 8) >> 24), RES_U8 ? (float)(rqu >> 24) : (float)(rq >> 24)};   // centred r'
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

// (b'') / (c'') the same two with a quantised channel dropout behind the conv (conv_resnet_mc): the tile then holds the dropped
// value centred on the mask's zero point (= the next conv's input zero point); the Add takes it as its first operand.
template <int HO, int PIXB, int TILE_BYTES, int COUT, bool BITS = false>
struct EpiTileDrop {
  static constexpr int VALU_PER_MFMA = 16;
  uint8_t* dst; QConv p; PostArgs q; MaskTab<COUT, BITS> mt;
  __device__ __forceinline__ int pixel(int m) const { return tile_px_off<HO, PIXB, TILE_BYTES>(m, 0); }
  __device__ __forceinline__ uint32_t load(int, int) const { return 0u; }
  __device__ __forceinline__ void store(int po, int c0, float v0, float v1, float v2, float v3, uint32_t) const {
    const float4 m4 = mt.get(po / TILE_BYTES, c0);
    *reinterpret_cast<uint32_t*>(dst + po + c0) =
        pack_low_bytes(drop_val(v0, p, q, m4.x) + QBNN_MAGIC, drop_val(v1, p, q, m4.y) + QBNN_MAGIC, drop_val(v2, p, q, m4.z) + QBNN_MAGIC,
                       drop_val(v3, p, q, m4.w) + QBNN_MAGIC);
  }
};
template <int HO, int PIXB, int TILE_BYTES, int COUT, bool BITS = false>
struct EpiTileResInPlaceDrop {
  static constexpr int VALU_PER_MFMA = 28;
  uint8_t* xt; QConv p; QAdd a; PostArgs q; MaskTab<COUT, BITS> mt;
  __device__ __forceinline__ int pixel(int m) const { return tile_px_off<HO, PIXB, TILE_BYTES>(m, 0); }
  __device__ __forceinline__ uint32_t load(int po, int c0) const { return *reinterpret_cast<const uint32_t*>(xt + po + c0); }
  __device__ __forceinline__ void store(int po, int c0, float v0, float v1, float v2, float v3, uint32_t rqu) const {
    const float4 m4 = mt.get(po / TILE_BYTES, c0);
    const float r[4] = {drop_val(v0, p, q, m4.x), drop_val(v1, p, q, m4.y), drop_val(v2, p, q, m4.z), drop_val(v3, p, q, m4.w)};
    const int rq = (int)rqu;
    const float rf[4] = {(float)((rq << 24) >> 24), (float)((rq << 16) >> 24), (float)((rq << 8) >> 24), (float)(rq >> 24)};   // centred residual
    float t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float da = __builtin_fmaf(q.s_a, __builtin_rintf(r[i]), q.dl_a);
      const float db = __builtin_fmaf(a.s_r, rf[i], a.dl_r);
      t[i] = (da + db) * a.inv_s_o;
    }
    *reinterpret_cast<uint32_t*>(xt + po + c0) = pack_rne_u8(t[0], t[1], t[2], t[3], a.vhi);
  }
};

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
// argument block of the single-conv kernels (layer-level C ABI entry qbnn_conv2d_i8_mc)
struct ConvArgs {
  const uint8_t* x; int64_t x_ss;
  const uint8_t* res; int64_t res_ss;
  uint8_t* y; int64_t y_ss;
  int B;
  QConv p; QAdd a;
  PostArgs post;                   // POST kernels only
};

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
  int pool;                           // QBNN_BLOCK_POOL_OUT of the chain's last block: y is the AvgPool2d(H) of the output, [S][B][C]
  unsigned long long* dbg;            // diagnostic builds only
  BlockParams blk[NBLK];
  const int8_t* stem_x;               // fused layer-0 conv (STEM kernels): centred im2col patches [B][32*32][32], shared by the samples
  QConv stem;
};

// argument block of the fused down-sampling BasicBlock kernels (qbnn_blocks.hip: block_down_ws_kernel; qbnn_down_ring.hip)
struct DownArgs {
  const uint8_t* x; int64_t x_ss;
  uint8_t* y; int64_t y_ss;
  int B, n_samples, z_in;
  QConv s, a, b; QAdd add;      // shortcut 1x1/s2, stem.0 3x3/s2 (ReLU), stem.3 3x3, Add
};

// Several independent launches of one fused kernel in ONE grid: gridDim.y (head: gridDim.z) picks the argument block.  Used for
// ensemble members (reference sgld/models_sgld.py:277-288): every member has its own tensors, weights AND quantisation
// parameters, so it cannot ride the MC-sample dimension of a launch -- but member m's workgroups can sit next to member
// m+1's.  NM = 1 is the ordinary launch (same code, argument block 0).
#define QBNN_FUSED_CALLS 8            // argument blocks per launch (kernel arguments are limited to 4 KiB)
template <class A, int NM> struct ArgsArr { A m[NM]; };
// NM = 0: any number of argument blocks, in DEVICE memory (the prepared multi-call launches: qbnn_*_multi_prepare / _launch) -- the
// blocks are read with scalar loads instead of arriving as kernel arguments, so one grid holds all 16 members of an ensemble.
template <class A> struct ArgsArr<A, 0> { const A* m; };
// The argument block of this workgroup's call.  NM = 0: read through the CONSTANT address space, i.e. with scalar loads into SGPRs once, like
// kernel arguments -- as plain global memory the compiler fetched every field with (uniform) VECTOR loads wherever it was used: 20 - 35 extra
// global / flat loads and as many `s_waitcnt vmcnt` per kernel, inside the item loops, where they also wait for whatever else is in flight
// (the weight ring's slabs).  The blocks are written by qbnn_*_multi_prepare before the launch and never during it.
// (pointers that arrive as kernel arguments are known to point to global memory; pointers loaded from an argument block in memory are
//  "generic" to the compiler, and every access through them a flat_load / flat_store, which counts on the LDS counter too: tell it)
template <class T> __device__ __forceinline__ T* as_global(T* p) {
  return (T*)reinterpret_cast<__attribute__((address_space(1))) T*>(reinterpret_cast<uintptr_t>(p));      // integer -> global pointer -> generic
}
__device__ __forceinline__ void globalize(QConv& q) { q.w = as_global(q.w); q.bias = as_global(q.bias); }
__device__ __forceinline__ void globalize(DownArgs& a) { a.x = as_global(a.x); a.y = as_global(a.y); globalize(a.s); globalize(a.a); globalize(a.b); }
template <int NBLK> __device__ __forceinline__ void globalize(ChainArgs<NBLK>& a) {
  a.x = as_global(a.x); a.y = as_global(a.y); a.stem_x = as_global(a.stem_x); globalize(a.stem);
#pragma unroll
  for (int k = 0; k < NBLK; ++k) { globalize(a.blk[k].a); globalize(a.blk[k].b); }
}
template <class A, int NM>
__device__ __forceinline__ A args_of(const ArgsArr<A, NM>& all, int idx) {
  if constexpr (NM == 0) {
    static_assert(sizeof(A) % 4 == 0, "argument blocks are whole dwords");
    union U { A a; uint32_t w[sizeof(A) / 4]; __device__ U() {} } u;
    const __attribute__((address_space(4))) uint32_t* q =
        reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(reinterpret_cast<uintptr_t>(all.m + idx));
#pragma unroll
    for (int i = 0; i < (int)(sizeof(A) / 4); ++i) u.w[i] = q[i];
    globalize(u.a);
    return u.a;
  } else {
    return all.m[NM == 1 ? 0 : idx];
  }
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
//   PRESUB: the accumulators were initialised with -z_w R (the window sum was known before the K loop: channel-sum tables), so they already
//   hold sum x' (W - z_w) -- one vector instruction fewer per output
template <class C, class Epi, class ResFn, class AheadFn, bool PRESUB = false>
__device__ __forceinline__ void conv_epi_phase_with(const float* bias_lds, const QConv& p, Epi& epi, ConvAcc<C>& A, int pass, int lane,
                                                    ResFn resfn, AheadFn ahead) {
  const int r = lane & 31, h = lane >> 5;
  const int mblk = pass / C::NBLKS, nblk = pass - mblk * C::NBLKS;
#pragma unroll
  for (int mb = 0; mb < C::MB; ++mb) {
    int R = 0;
    if constexpr (PRESUB) {
    } else if (C::USE_ONES) {
      const int rv = A.acc[mb][C::ONES_TILE % C::NB][C::ONES_REG];
      R = half_lo_bcast(rv);
    } else {
      R = half_sum(A.rsum[mb]);
    }
    const int zwr = PRESUB ? 0 : p.z_w * R;
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
          const int zwr = p.z_w * half_lo_bcast(rv);
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

// the same two epilogues with a quantised channel dropout behind the conv (conv_resnet_mc)
template <int PIXB, int COUT, int IMG_PX, bool BITS = false>
struct EpiDenseTileDrop {
  uint8_t* dst; QConv p; PostArgs q; MaskTab<COUT, BITS> mt;
  mutable int csum;
  __device__ __forceinline__ int pixel(int m) const { return m * PIXB; }
  __device__ __forceinline__ uint32_t load(int, int) const { return 0u; }
  __device__ __forceinline__ void store(int po, int c0, float v0, float v1, float v2, float v3, uint32_t) const {
    const float4 m4 = mt.get(po / (IMG_PX * PIXB), c0);
    const uint32_t pk = pack_low_bytes(drop_val(v0, p, q, m4.x) + QBNN_MAGIC, drop_val(v1, p, q, m4.y) + QBNN_MAGIC,
                                       drop_val(v2, p, q, m4.z) + QBNN_MAGIC, drop_val(v3, p, q, m4.w) + QBNN_MAGIC);
    *reinterpret_cast<uint32_t*>(dst + po + c0) = pk;
    csum = __builtin_amdgcn_sdot4((int)pk, 0x01010101, csum, false);
  }
};
template <int PIXB, int CCH, int IMG_PX, bool BITS = false>
struct EpiDenseTileResGlobalDrop {
  uint8_t* xt; const uint8_t* res; int n_valid_px; QConv p; QAdd a; PostArgs q; MaskTab<CCH, BITS> mt;
  __device__ __forceinline__ int pixel(int m) const { return m * PIXB; }
  __device__ __forceinline__ uint32_t load_px(int m, int c0) const {
    return m < n_valid_px ? *reinterpret_cast<const uint32_t*>(res + (int64_t)m * CCH + c0) : 0u;
  }
  __device__ __forceinline__ uint32_t load(int, int) const { return 0u; }
  __device__ __forceinline__ void store(int po, int c0, float v0, float v1, float v2, float v3, uint32_t rq) const {
    const float4 m4 = mt.get(po / (IMG_PX * PIXB), c0);
    const float r[4] = {drop_val(v0, p, q, m4.x), drop_val(v1, p, q, m4.y), drop_val(v2, p, q, m4.z), drop_val(v3, p, q, m4.w)};
    const float rf[4] = {(float)(rq & 0xffu), (float)((rq >> 8) & 0xffu), (float)((rq >> 16) & 0xffu), (float)(rq >> 24)};
    float t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float da = __builtin_fmaf(q.s_a, __builtin_rintf(r[i]), q.dl_a);
      const float db = __builtin_fmaf(a.s_r, rf[i], a.nzs_r);
      t[i] = (da + db) * a.inv_s_o;
    }
    *reinterpret_cast<uint32_t*>(xt + po + c0) = pack_rne_u8(t[0], t[1], t[2], t[3], a.vhi);
  }
};

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

#endif
