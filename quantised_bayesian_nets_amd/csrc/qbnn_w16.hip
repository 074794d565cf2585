// libqbnn_hip.so -- 16-wave forms of the weights-stationary fused kernels (layers whose epilogue, not the matrix pipe, is the
// bound: 24 / 48 channels).
//
// Why 16 waves.  The exact FBGEMM requantisation costs 6 vector instructions per output (13 with the residual Add), and one
// wave issues a vector instruction only every ~6 cycles: measured in-kernel (tools/issue_bench.hip, profiles/r03_issue_bench.txt)
// a SIMD issues the epilogue's instruction mix at 4.1 cycles per wave-instruction with 2 resident waves, 2.9 with 4 and 2.2
// with 8, and a 9-MFMA + 72-VALU tile takes 450 / 351 / 270 cycles at 2 / 4 / 8 waves per SIMD (the MFMAs alone: 288).  The
// 8-wave kernels of qbnn_blocks.hip hold 200+ VGPRs per wave (two waves per SIMD); the kernels here stay under 128 so that a
// 1024-thread workgroup fits a CU: four waves per SIMD, each wave a few output rows of the image.
//
// Layer 1 (qbnn_stem_chain_i8_mc with two blocks): layers.0 on the 27-tap patches + 2 x [stem.0, stem.3 + Add + ReLU] on a
// 32 x 32 x 24 map.  Work item = (MC sample, G images).  Wave w owns the 2 G output rows w * 2 G ... of the item in every conv.
//   * the patch fragments of layers.0 come straight from global memory (one 16-byte load per lane and output row, L2 / MALL
//     resident: the patch tensor is shared by all MC samples) -- no patch tile in LDS, no copy-in pass;
//   * per conv a wave keeps the 9 weight fragments in registers and walks its rows with a rolling 3-row window of pixel
//     fragments (each input row is read from LDS once per wave): 9 MFMAs, then the row's epilogue;
//   * X tile (block input / output, updated in place) and T tile (stem.0 output) per image; weights of the 5 convs in LDS,
//     re-read per MC sample only (contiguous item ranges per workgroup).
// Same arithmetic, epilogue functors and packed weight layout as block_chain_ws_kernel: bit-identical results.
#include "qbnn_host.h"

// Diagnostic build only (-DQBNN_W16_STAMP, scratch library; tools/stamp_w16.py): s_memtime of every wave of workgroup 0 at the
// phase boundaries of one work item -- after each row's MFMAs and after its epilogue -- written to a debug buffer that nothing
// else reads.  The shipped library contains none of this.
#ifdef QBNN_W16_STAMP
static __device__ unsigned long long* g_w16_stamp = nullptr;
QBNN_EXPORT void qbnn_debug_w16_stamp_buffer(void* p) { hipMemcpyToSymbol(HIP_SYMBOL(g_w16_stamp), &p, sizeof(p)); }
#define W16_STAMP() do { if (st_on) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    if ((threadIdx.x & 63) == 0) g_w16_stamp[(threadIdx.x >> 6) * 64 + (st_k & 63)] = t_; ++st_k; __builtin_amdgcn_sched_barrier(0); } } while (0)
#define W16_STAMP_ARGS , bool st_on, int& st_k
#define W16_STAMP_PASS , st_on, st_k
#else
#define W16_STAMP() do {} while (0)
#define W16_STAMP_ARGS
#define W16_STAMP_PASS
#endif

// Progress-based issue priority (round 5; -DQBNN_W16_PRIO=0 for the A/B): a wave lowers its priority with every output row it finishes
// inside a conv phase (3, 2, 1, 0), so the waves of a SIMD that are behind win arbitration over those ahead -- the hardware's oldest-first
// rule lets the four waves of a SIMD finish a phase 2 - 7 k cycles apart (profiles/r03_stamp_w16_g2.txt), and the tail of every phase runs
// at one or two waves per SIMD, where a vector instruction costs 4 - 8 cycles instead of 2.9.  s_setprio is also a scheduling fence: the
// rows of a wave are issued one after the other.  Measured: 1.044 -> 1.014 ms (plain), 0.989 -> 0.940 ms (MAGIC).
#ifndef QBNN_W16_PRIO
#define QBNN_W16_PRIO 1
#endif
#if QBNN_W16_PRIO
#define W16_PRIO(i) do { if ((i) == 0) __builtin_amdgcn_s_setprio(3); else if ((i) == 1) __builtin_amdgcn_s_setprio(2); \
                         else if ((i) == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); } while (0)
#else
#define W16_PRIO(i) do {} while (0)
#endif

namespace {

constexpr int W16_THREADS = 1024, W16_WAVES = 16;
using L1 = ConvCfg<24, 24, 3, 1, 32, 1, 1, 4, 1>;      // one 32 x 32 x 24 image tile (halo 1): pitch 34 * 24 bytes
constexpr int W16_KS = 7, W16_WB = W16_KS * 1024;     // k-steps / packed bytes of a 24 -> 24 3x3 conv in the QBNN_LAYOUT_MFMA32_TAIL layout
struct W24T { static constexpr int NT = 1, KS = W16_KS; };      // (dma_conv)
using L0 = ConvCfg<32, 24, 1, 1, 32, 0, 1, 4, 1>;      // layers.0 on the patch tensor: K = 27 -> 32, one k-step

template <int G, int NBLK, bool DROP = false> constexpr int w16_lds() {
  return 2 * (G * L1::TILE_BYTES + L1::TILE_SLACK) + 2 * NBLK * W16_WB + WConv<L0>::BYTES + (2 * NBLK + 1) * L1::COUT * 4 +
         (DROP ? 2 * (2 * NBLK + 1) * MaskTab<L1::COUT, false>::bytes(G) : 0);      // two sets of mask tables: consecutive items alternate
}

// requantise one 32-pixel x 24-channel accumulator tile (ones row -> window sum, see conv_core) through `epi`.
// MAGIC: the accumulators were started at the bit pattern of 1.5 * 2^23 (MAGIC16) instead of 0, so a register read as fp32 IS
// 1.5 * 2^23 + sum exactly while |sum| < 2^22 -- K * 128 * max|x'| = 216 * 128 * 127 = 3.5 M here (the launcher checks the activation
// bound) -- and (float)(sum - z_w R) = as_float(acc) - (1.5 * 2^23 + z_w R): ONE v_sub_f32 of two exactly represented integers with an
// exactly representable difference, where the plain form needs v_sub_u32 + v_cvt_f32_i32.  The window sum leaves the ones row the
// same way: R = as_float(acc[ones]) - 1.5 * 2^23, and 1.5 * 2^23 + z_w R is one fma on integers below 2^24.  Same bits.
#define QBNN_MAGIC_BITS 0x4B400000
template <bool MAGIC, class Epi>
__device__ __forceinline__ void epilogue24(const v16i& acc, const float4 (&b4)[3], const QConv& p, const Epi& epi, int po, int h) {
  const int rv = acc[L1::ONES_REG];
  int zwr = 0;
  float cm = 0.f;
  if constexpr (MAGIC) cm = __builtin_fmaf((float)p.z_w, __int_as_float(half_lo_bcast(rv)) - QBNN_MAGIC, QBNN_MAGIC);
  else zwr = p.z_w * half_lo_bcast(rv);
  uint32_t pre[3];
#pragma unroll
  for (int g4 = 0; g4 < 3; ++g4) pre[g4] = epi.load(po, 8 * g4 + 4 * h);
#pragma unroll
  for (int g4 = 0; g4 < 3; ++g4) {
    const float4 bb = b4[g4];
    float a0, a1, a2, a3;
    if constexpr (MAGIC) {
      a0 = __int_as_float(acc[4 * g4 + 0]) - cm; a1 = __int_as_float(acc[4 * g4 + 1]) - cm;
      a2 = __int_as_float(acc[4 * g4 + 2]) - cm; a3 = __int_as_float(acc[4 * g4 + 3]) - cm;
    } else {
      a0 = (float)(acc[4 * g4 + 0] - zwr); a1 = (float)(acc[4 * g4 + 1] - zwr);
      a2 = (float)(acc[4 * g4 + 2] - zwr); a3 = (float)(acc[4 * g4 + 3] - zwr);
    }
    const float v0 = __builtin_fmaf(bb.x, p.rcp, a0) * p.mult;
    const float v1 = __builtin_fmaf(bb.y, p.rcp, a1) * p.mult;
    const float v2 = __builtin_fmaf(bb.z, p.rcp, a2) * p.mult;
    const float v3 = __builtin_fmaf(bb.w, p.rcp, a3) * p.mult;
    epi.store(po, 8 * g4 + 4 * h, v0, v1, v2, v3, pre[g4]);
  }
}
template <bool MAGIC> __device__ __forceinline__ v16i acc_start() {
  constexpr int m = MAGIC ? QBNN_MAGIC_BITS : 0;
  return v16i{m, m, m, m, m, m, m, m, m, m, m, m, m, m, m, m};
}
// The start block of a conv phase, built INSIDE the phase from an opaque register (16 moves per phase and wave).  As a plain constant
// the compiler builds it once per kernel and keeps it in scratch across the item loop: a scratch reload per phase is a vector-memory
// round trip behind a vmcnt(0) that also drains the output stores and the patch prefetch (measured: 1.29 against 1.10 ms).
__device__ __forceinline__ v16i magic_block() {
  int m = QBNN_MAGIC_BITS;
  asm volatile("" : "+v"(m));
  v16i b = v16i{m, m, m, m, m, m, m, m, m, m, m, m, m, m, m, m};
  asm volatile("" : "+v"(b));
  return b;
}

// byte offset of interior pixel (oh, ow) of image g inside a tile array
__device__ __forceinline__ int px_off(int g, int oh, int ow) { return g * L1::TILE_BYTES + ((oh + 1) * L1::TW + ow + 1) * L1::PIXB; }

// The chain's LAST conv: Add(residual from the X tile) + ReLU as EpiTileResInPlace, but the block output leaves as quint8 straight to
// global memory (three dwords per lane and row: the 24-byte pixel records of a row are contiguous) -- nothing reads the X tile after
// it, so the item needs no read-out pass (tile -> registers -> + z_o -> HBM by all threads) and no barrier in front of one.
struct OutRow { uint8_t* y; int n_ok; };      // this lane's pixel column of the ITEM's first image in the output tensor (row 0); n_ok: images of the item that exist (ragged batch)
template <bool DROP>
struct EpiResToGlobal {
  const uint8_t* xt; OutRow o; QConv p; QAdd a; PostArgs q; MaskTab<L1::COUT, false> mt;
  mutable uint8_t* yrow; mutable bool ok;
  // called before a row's stores (image slot g and output row oh: wave-uniform)
  __device__ __forceinline__ void set_row(int g, int oh) const { yrow = o.y + (g * L1::HIN + oh) * (L1::HIN * L1::COUT); ok = g < o.n_ok; }
  __device__ __forceinline__ uint32_t load(int po, int c0) const { return *reinterpret_cast<const uint32_t*>(xt + po + c0); }
  __device__ __forceinline__ void store(int po, int c0, float v0, float v1, float v2, float v3, uint32_t rqu) const {
    float r[4] = {v0, v1, v2, v3};
    float sa = p.s_y, dl = p.dl_y;
    if constexpr (DROP) {
      const float4 m4 = mt.get(po / L1::TILE_BYTES, c0);
      r[0] = drop_val(v0, p, q, m4.x); r[1] = drop_val(v1, p, q, m4.y); r[2] = drop_val(v2, p, q, m4.z); r[3] = drop_val(v3, p, q, m4.w);
      sa = q.s_a; dl = q.dl_a;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = med3f(r[i], p.vlo, p.vhi);
    }
    const int rq = (int)rqu;
    // the residual is the X tile: without dropout its bytes come from ReLU-fused epilogues of this kernel (layers.0, the previous
    // block's Add + ReLU), centred and non-negative -- one v_cvt_f32_ubyteN each; a dropped value may lie below the mask's zero point
    const float rf[4] = {DROP ? (float)((rq << 24) >> 24) : (float)(rqu & 0xffu), DROP ? (float)((rq << 16) >> 24) : (float)((rqu >> 8) & 0xffu),
                         DROP ? (float)((rq << 8) >> 24) : (float)((rqu >> 16) & 0xffu), DROP ? (float)(rq >> 24) : (float)(rqu >> 24)};
    float t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float da = __builtin_fmaf(sa, __builtin_rintf(r[i]), dl);
      const float db = __builtin_fmaf(a.s_r, rf[i], a.dl_r);
      t[i] = (da + db) * a.inv_s_o;
    }
    if (ok) *reinterpret_cast<uint32_t*>(yrow + c0) = pack_rne_u8(t[0], t[1], t[2], t[3], a.vhi) + (uint32_t)a.z_o * 0x01010101u;
  }
};
template <class E> __device__ __forceinline__ auto epi_set_row(const E& e, int g, int oh, int) -> decltype(e.set_row(g, oh), void()) { e.set_row(g, oh); }
template <class E> __device__ __forceinline__ void epi_set_row(const E&, int, int, long) {}

// 3x3 / stride 1 conv of RW consecutive output rows [oh0, oh0 + RW) of image slot g: tile -> epi.  Weights in the QBNN_LAYOUT_MFMA32_TAIL
// layout (round 5): a 72-byte kernel row is two full k-steps (bytes 0 .. 63: pixels c - 1, c and two thirds of c + 1) and an 8-byte tail (the
// last 8 channels of pixel c + 1); the three rows' tails share ONE k-step -- 7 MFMAs per output row instead of 9 for the same 216 weights per
// channel.  The tail step's pixel fragment is gathered per output row: k-half 0 = [tail of row i | tail of row i + 1], k-half 1 = [tail of
// row i + 2 | anything: those weights are 0] -- one ds_read2_b64 from a per-half base.  `w` holds this conv's 7 weight fragments; after the
// last row's MFMAs it is refilled from `wnext` (the NEXT conv's weights, which do not depend on the barrier in between: the refill rides
// under the last epilogue instead of heading the next phase, where all 16 waves would burst it).
template <int RW, bool MAGIC, class Epi>
__device__ __forceinline__ void conv3x3_rows_w16(const uint8_t* tile, v4i (&w)[W16_KS], const uint8_t* wnext, const float* bias_lds,
                                                 const QConv& p, const Epi& epi, int g, int oh0, int lane, const v16i& mg_item W16_STAMP_ARGS) {
  int l_ = lane;
  asm volatile("" : "+v"(l_));     // per-lane offsets are recomputed per phase (hoisted out of the item loop they spill)
  const int r = l_ & 31, h = l_ >> 5;
  // tile row oh0 + j is input row oh0 - 1 + j; tile column r is input column r - 1: the 72-byte window of (row, r) starts there
  const uint8_t* base = tile + g * L1::TILE_BYTES + (oh0 * L1::TW + r) * L1::PIXB + 16 * h;
  const uint8_t* tbase = base - 16 * h + 64 + (h ? 2 * L1::PITCH : 0);      // tails: k-half 0 reads rows i, i + 1; k-half 1 rows i + 2 (, i + 3: unused)
  v4i x[RW + 2][2];
  auto load_row = [&](int j) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const v2i lo = *reinterpret_cast<const v2i*>(base + j * L1::PITCH + t * 32);
      const v2i hi = *reinterpret_cast<const v2i*>(base + j * L1::PITCH + t * 32 + 8);
      x[j][t] = v4i{lo.x, lo.y, hi.x, hi.y};
    }
  };
  auto load_tail = [&](int i) {
    const v2i lo = *reinterpret_cast<const v2i*>(tbase + i * L1::PITCH);
    const v2i hi = *reinterpret_cast<const v2i*>(tbase + (i + 1) * L1::PITCH);
    return v4i{lo.x, lo.y, hi.x, hi.y};
  };
  auto mfma_row = [&](int i, const v4i& xt, const v16i& start) {
    v16i acc = start;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[kh * 2 + t], x[i + kh][t], acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(w[6], xt, acc, 0, 0, 0);
  };
  float4 b4[3];
#pragma unroll
  for (int g4 = 0; g4 < 3; ++g4) b4[g4] = *reinterpret_cast<const float4*>(bias_lds + 8 * g4 + 4 * h);
  if constexpr (MAGIC) {
    // ONE live copy of the start block, one accumulator, the rows one after the other (a scheduling fence per row: with all four rows'
    // MFMAs run ahead, as the compiler schedules the plain form, 4 x 16 accumulators + the start block do not fit 128 VGPRs); an input
    // row is read when its first output row needs it.
    const v16i& mg = mg_item;
    load_row(0); load_row(1);
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      W16_PRIO(i);
      load_row(i + 2);
      const v4i xt = load_tail(i);
      const v16i acc = mfma_row(i, xt, mg);
      if (i == RW - 1 && wnext) {
#pragma unroll
        for (int ks = 0; ks < W16_KS; ++ks) w[ks] = *reinterpret_cast<const v4i*>(wnext + l_ * 16 + ks * 1024);
      }
      W16_STAMP();
      epi_set_row(epi, g, oh0 + i, 0);
      epilogue24<true>(acc, b4, p, epi, px_off(g, oh0 + i, r), h);
      W16_STAMP();
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  load_row(0); load_row(1); load_row(2);
  const v16i zero16 = acc_start<false>();
  // One accumulator, no software pipeline inside the wave: while this wave waits on its MFMAs the SIMD's other three waves
  // issue their epilogues (a wave of its own issues a vector instruction every ~6 cycles, the SIMD one every ~2-3).
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    W16_PRIO(i);
    const v4i xt = load_tail(i);
    const v16i acc = mfma_row(i, xt, zero16);
    if (i + 3 < RW + 2) load_row(i + 3);
    if (i == RW - 1 && wnext) {
#pragma unroll
      for (int ks = 0; ks < W16_KS; ++ks) w[ks] = *reinterpret_cast<const v4i*>(wnext + l_ * 16 + ks * 1024);
    }
    W16_STAMP();
    epi_set_row(epi, g, oh0 + i, 0);
    epilogue24<false>(acc, b4, p, epi, px_off(g, oh0 + i, r), h);
    W16_STAMP();
  }
}

// All 16 waves work on one item of G images; hardware barriers (LDS-only fences) between the convs.
// (Measured alternatives, kept out of the build -- tools/experiments/r03_w16_teams_and_ablations.hip.txt: G = 1: 1.13-1.18 ms against
//  1.09; two 8-wave teams on one image each, half an item apart, synchronised through LDS counters: 1.10 ms -- the older team
//  takes the issue slots (53 k against 97 k cycles per image) and the sum does not change; the upper half of the waves started
//  3 / 6 / 10 x 64 cycles late after every barrier: no change.  Timing ablations of the same source: without the MFMAs 0.93 ms,
//  without the epilogue arithmetic 0.61, without the pixel-fragment reads 1.08, without the barriers 1.04.)
// DROP (conv_resnet_mc): a quantised channel dropout behind every conv -- dr.d = layers.3, then per block stem.3, stem.6 -- applied in the
// epilogues from per-item mask tables in LDS (block_chain_ws_kernel's DROP form on 16 waves).
template <int G, int NBLK, int NM, bool DROP = false, bool MAGIC = false>
__global__ __launch_bounds__(W16_THREADS) void stem_chain_w16_kernel(const ArgsArr<ChainArgs<NBLK>, NM> all, const DropSet<DROP ? 2 * NBLK + 1 : 0> dr) {
  const ChainArgs<NBLK> a = args_of(all, blockIdx.y);
  static_assert(!DROP || NM == 1, "dropout variants are single-call");
  constexpr int MTB = MaskTab<L1::COUT, false>::bytes(G);
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr int RW = 2 * G;                                  // output rows per wave and conv (32 rows per image)
  constexpr int TILES = G * L1::TILE_BYTES + L1::TILE_SLACK;
  constexpr int WB = W16_WB;
  uint8_t* xt = smem;
  uint8_t* tt = smem + TILES;
  uint8_t* wl = smem + 2 * TILES;                            // [NBLK][2] whole convs
  uint8_t* wl0 = wl + 2 * NBLK * WB;                         // layers.0: one fragment tile
  float* bias_lds = reinterpret_cast<float*>(wl0 + WConv<L0>::BYTES);      // [2 NBLK][24], then layers.0's
  float* bias0 = bias_lds + 2 * NBLK * L1::COUT;
  uint8_t* mtab0 = reinterpret_cast<uint8_t*>(bias0 + L0::COUT);             // DROP: mask tables [2][2 NBLK + 1][G][24]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = (wave * RW) / 32, woh0 = (wave * RW) % 32;  // this wave's image within the item and its first output row

  const int groups = (a.B + G - 1) / G;
  int begin, count;
  item_range(a.n_samples * groups, blockIdx.x, gridDim.x, begin, count);
  if (count <= 0) return;

  zero_halo<L1::TW, L1::PIXB, L1::TILE_BYTES, G, W16_THREADS>(xt, tid);
  zero_halo<L1::TW, L1::PIXB, L1::TILE_BYTES, G, W16_THREADS>(tt, tid);
#pragma unroll
  for (int k = 0; k < NBLK; ++k) {
    load_bias<L1::COUT, W16_THREADS>(bias_lds + (2 * k) * L1::COUT, a.blk[k].a.bias, tid);
    load_bias<L1::COUT, W16_THREADS>(bias_lds + (2 * k + 1) * L1::COUT, a.blk[k].b.bias, tid);
  }
  load_bias<L0::COUT, W16_THREADS>(bias0, a.stem.bias, tid);
  auto sync = [&]() { lds_barrier(); };

  // layers.0's pixel fragments of this wave's rows, one 16-byte load per lane and row straight from the patch tensor
  v4i pf[RW];
  // (per-lane addresses are recomputed from an opaque copy of the lane / thread index wherever they are used once per item:
  //  hoisted out of the item loop they are spilled, and a spill reload is a vmcnt wait behind the prefetch in flight)
  auto fetch_patches = [&](int item) {
    const int s = item / groups;
    int img = (item - s * groups) * G + wg;
    img = img < a.B ? img : a.B - 1;
    int l_ = lane;
    asm volatile("" : "+v"(l_));
    const int8_t* ps = a.stem_x + ((int64_t)img * 1024 + woh0 * 32 + (l_ & 31)) * 32 + 16 * (l_ >> 5);
#pragma unroll
    for (int i = 0; i < RW; ++i) pf[i] = *reinterpret_cast<const v4i*>(ps + i * 32 * 32);
  };
  fetch_patches(begin);

  constexpr int CPR = L1::ROWB / 16, CPI = L1::HIN * CPR;                       // 16-byte chunks per image row / per image
  constexpr int NCH = G * CPI, OTHR = W16_THREADS;                            // read-out: 16-byte chunks of the item, threads
  constexpr int PER_TO = (NCH + OTHR - 1) / OTHR;
  v4i w[W16_KS];
  int cur_s = -1;
  for (int it = 0; it < count; ++it) {
    const int item = begin + it;
    const int s = item / groups, img0 = (item - s * groups) * G;
#ifdef QBNN_W16_STAMP
    const bool st_on = g_w16_stamp != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && it == 2;
    int st_k = 0;
#endif
    if (s != cur_s) {            // workgroup-uniform; a few times per launch
      __syncthreads();           // every wave is done with the previous sample's weights (and the prologue's LDS writes)
      int l_ = lane;
      asm volatile("" : "+v"(l_));
#pragma unroll
      for (int k = 0; k < NBLK; ++k) {
        dma_conv<W24T, W16_WAVES>(wl + (2 * k) * WB, a.blk[k].a.w + (int64_t)s * a.blk[k].a.w_ss, wave, l_);
        dma_conv<W24T, W16_WAVES>(wl + (2 * k + 1) * WB, a.blk[k].b.w + (int64_t)s * a.blk[k].b.w_ss, wave, l_);
      }
      dma_conv<L0, W16_WAVES>(wl0, a.stem.w + (int64_t)s * a.stem.w_ss, wave, l_);
      dma_barrier();             // vmcnt(0) + barrier: the weights have landed
      cur_s = s;
    }
    {                            // stem.0's weights of the first block: live across layers.0 (which only needs one fragment)
      int l_ = lane;
      asm volatile("" : "+v"(l_));
#pragma unroll
      for (int ks = 0; ks < W16_KS; ++ks) w[ks] = *reinterpret_cast<const v4i*>(wl + l_ * 16 + ks * 1024);
    }
    v16i mg_item = acc_start<false>();
    if constexpr (MAGIC) mg_item = magic_block();          // ONE copy of the start block per item (round 6 trial)
    W16_STAMP();
    // (the item's last epilogue -- straight to HBM -- is not followed by a barrier: a wave may fill the next item's tables while
    //  another still reads this item's, so consecutive items use different table sets)
    uint8_t* mtab = mtab0 + (it & 1) * (2 * NBLK + 1) * MTB;
    if constexpr (DROP) {        // this item's masks
      int t_ = tid;
      asm volatile("" : "+v"(t_));
#pragma unroll
      for (int d = 0; d < 2 * NBLK + 1; ++d) fill_mask_tab<G, L1::COUT, false, W16_THREADS>(mtab + d * MTB, dr.d[d], s, img0, a.B, t_);
    }
    sync();                      // the previous item's X tile has been read out by every thread
    W16_STAMP();
    auto stem_conv = [&](const auto& epi) {        // layers.0 (ConvReLU2d): patches -> X tile, centred on its own zero point
      int l_ = lane;
      asm volatile("" : "+v"(l_));
      const int r = l_ & 31, h = l_ >> 5;
      const v4i w0 = *reinterpret_cast<const v4i*>(wl0 + l_ * 16);
      float4 b4[3];
#pragma unroll
      for (int g4 = 0; g4 < 3; ++g4) b4[g4] = *reinterpret_cast<const float4*>(bias0 + 8 * g4 + 4 * h);
      v16i zero16 = acc_start<false>();
      if constexpr (MAGIC) zero16 = mg_item;
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        W16_PRIO(i);
        const v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0, pf[i], zero16, 0, 0, 0);
        W16_STAMP();
        epilogue24<MAGIC>(acc, b4, a.stem, epi, px_off(wg, woh0 + i, r), h);
        W16_STAMP();
      }
    };
    if constexpr (DROP) stem_conv(EpiTileDrop<L1::HO, L1::PIXB, L1::TILE_BYTES, L1::COUT>{xt, a.stem, dr.d[0], {mtab, 0.f}});
    else stem_conv(EpiTile<L1::HO, L1::PIXB, L1::TILE_BYTES>{xt, a.stem});
    sync();
    W16_STAMP();
#pragma unroll
    for (int k = 0; k < NBLK; ++k) {
      const BlockParams& bp = a.blk[k];
      auto conv_a = [&](const auto& epi) {
        conv3x3_rows_w16<RW, MAGIC>(xt, w, wl + (2 * k + 1) * WB, bias_lds + (2 * k) * L1::COUT, bp.a, epi, wg, woh0, lane, mg_item W16_STAMP_PASS);
      };
      auto conv_b = [&](const auto& epi) {
        conv3x3_rows_w16<RW, MAGIC>(tt, w, k + 1 < NBLK ? wl + (2 * k + 2) * WB : nullptr, bias_lds + (2 * k + 1) * L1::COUT, bp.b, epi, wg, woh0,
                             lane, mg_item W16_STAMP_PASS);
      };
      if constexpr (DROP) conv_a(EpiTileDrop<L1::HO, L1::PIXB, L1::TILE_BYTES, L1::COUT>{tt, bp.a, dr.d[1 + 2 * k], {mtab + (1 + 2 * k) * MTB, 0.f}});
      else conv_a(EpiTile<L1::HO, L1::PIXB, L1::TILE_BYTES>{tt, bp.a});
      sync();
      W16_STAMP();
      if (k + 1 == NBLK) {             // the chain's output: quint8 straight to HBM
        int l_ = lane;
        asm volatile("" : "+v"(l_));
        const OutRow orow{a.y + (int64_t)s * a.y_ss + (int64_t)img0 * (L1::HIN * L1::HIN * L1::COUT) + (l_ & 31) * L1::COUT, a.B - img0};
        if constexpr (DROP) conv_b(EpiResToGlobal<true>{xt, orow, bp.b, bp.add, dr.d[2 + 2 * k], {mtab + (2 + 2 * k) * MTB, 0.f}, nullptr, false});
        else conv_b(EpiResToGlobal<false>{xt, orow, bp.b, bp.add, PostArgs{}, {nullptr, 0.f}, nullptr, false});
      } else {
        if constexpr (DROP) conv_b(EpiTileResInPlaceDrop<L1::HO, L1::PIXB, L1::TILE_BYTES, L1::COUT>{xt, bp.b, bp.add, dr.d[2 + 2 * k], {mtab + (2 + 2 * k) * MTB, 0.f}});
        else conv_b(EpiTileResInPlace<L1::HO, L1::PIXB, L1::TILE_BYTES, true>{xt, bp.b, bp.add});      // RES_U8: see EpiResToGlobal
        sync();
      }
      W16_STAMP();
    }
    // the next item's patch fragments: in flight during the read-out, the stores and the barrier at the loop top
    fetch_patches(it + 1 < count ? item + 1 : item);
    W16_STAMP();
  }
}

// May the accumulators start at 1.5 * 2^23 (epilogue24, MAGIC)?  Every 3x3 conv of the launch reads a tile whose bytes lie in
// [0, min(255, a_hi) - z] (ReLU-fused requantisation / Add + ReLU centred on their zero points), so |sum| <= 216 * 128 * min(255, a_hi):
// below 2^22 for a_hi <= 151 -- every activation precision the reference allows (quant_utils.py:120 caps it at 7 bits).  layers.0
// reads int8 patches: 27 * 128 * 128.  QBNN_W16_MAGIC=0 keeps the plain form (A/B checks).
bool w16_magic_ok(int a_hi) {
  static const bool on = [] { const char* e = getenv("QBNN_W16_MAGIC"); return !(e && e[0] == '0'); }();
  return on && a_hi >= 0 && (int64_t)L1::KSZ * L1::KSZ * L1::CIN * 128 * (a_hi < 255 ? a_hi : 255) < (1 << 22);
}

template <int G, int NBLK, int NM, bool MAGIC>
int launch_w16(const ChainArgs<NBLK>* arr, int n, hipStream_t st) {
  constexpr int LDS = w16_lds<G, NBLK>();
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static_assert(sizeof(ArgsArr<ChainArgs<NBLK>, NM>) <= 3840, "kernel arguments are limited to 4 KiB (incl. the hidden ones)");
  static std::atomic<uint64_t> attr{0};
  if (int rc_attr = ensure_dyn_lds((const void*)stem_chain_w16_kernel<G, NBLK, NM, false, MAGIC>, attr, LDS)) return rc_attr;
  ArgsArr<ChainArgs<NBLK>, NM> all;
  memset(&all, 0, sizeof(all));
  int items = 0;
  for (int i = 0; i < n; ++i) {
    all.m[i] = arr[i];
    const int it = arr[i].n_samples * ((arr[i].B + G - 1) / G);
    items = it > items ? it : items;
  }
  const int per = 256 / n > 0 ? 256 / n : 1;
  const int gx = items < per ? (items > 0 ? items : 1) : per;
  hipLaunchKernelGGL((stem_chain_w16_kernel<G, NBLK, NM, false, MAGIC>), dim3(gx, n), dim3(W16_THREADS), LDS, st, all, DropSet<0>{});
  return check_launch("qbnn_stem_chain_i8_mc");
}

template <int G, int NBLK, bool MAGIC>
int launch_w16_drop(const ChainArgs<NBLK>& a, const DropSet<2 * NBLK + 1>& dr, hipStream_t st) {
  constexpr int LDS = w16_lds<G, NBLK, true>();
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static std::atomic<uint64_t> attr{0};
  if (int rc_attr = ensure_dyn_lds((const void*)stem_chain_w16_kernel<G, NBLK, 1, true, MAGIC>, attr, LDS)) return rc_attr;
  ArgsArr<ChainArgs<NBLK>, 1> one;
  one.m[0] = a;
  const int items = a.n_samples * ((a.B + G - 1) / G);
  hipLaunchKernelGGL((stem_chain_w16_kernel<G, NBLK, 1, true, MAGIC>), dim3(items < 256 ? items : 256), dim3(W16_THREADS), LDS, st, one, dr);
  return check_launch("qbnn_stem_chain_drop_i8_mc");
}

template <bool MAGIC>
int launch_w16_dev(const ChainArgs<2>* dev, int n, int items, hipStream_t st) {
  constexpr int LDS = w16_lds<2, 2>();
  static std::atomic<uint64_t> attr{0};
  if (int rc_attr = ensure_dyn_lds((const void*)stem_chain_w16_kernel<2, 2, 0, false, MAGIC>, attr, LDS)) return rc_attr;
  const int per = 256 / n > 0 ? 256 / n : 1;
  const int gx = items < per ? (items > 0 ? items : 1) : per;
  hipLaunchKernelGGL((stem_chain_w16_kernel<2, 2, 0, false, MAGIC>), dim3(gx, n), dim3(W16_THREADS), LDS, st, ArgsArr<ChainArgs<2>, 0>{dev}, DropSet<0>{});
  return check_launch("qbnn_block_chain_i8_multi_launch");
}

}  // namespace

int qbnn_launch_stem_chain_w16_dev(const ChainArgs<2>* dev, int n, int items, int a_hi, hipStream_t st) {
  return w16_magic_ok(a_hi) ? launch_w16_dev<true>(dev, n, items, st) : launch_w16_dev<false>(dev, n, items, st);
}

// (with dropout a tile holds dropped values centred on the mask's zero point: bytes in [-z_m, min(255, a_hi) - z_m], |x'| <= a_hi for the
//  z_m in [0, a_hi] the library accepts -- the same accumulator bound as without)
int qbnn_launch_stem_chain_w16_drop(const ChainArgs<2>& a, const DropSet<5>& dr, int a_hi, hipStream_t st) {
  bool zm_ok = true;
  for (int d = 0; d < 5; ++d) zm_ok = zm_ok && dr.d[d].z_m >= 0 && dr.d[d].z_m <= a_hi;
  return (w16_magic_ok(a_hi) && zm_ok) ? launch_w16_drop<2, 2, true>(a, dr, st) : launch_w16_drop<2, 2, false>(a, dr, st);
}

// entry point for qbnn_blocks.hip (declared in qbnn_host.h): 1 to 4 argument blocks in one grid, two images per work item
int qbnn_launch_stem_chain_w16(const ChainArgs<2>* arr, int n, int a_hi, hipStream_t st) {
  if (n < 1 || n > 4) return fail(QBNN_E_INVALID, "qbnn_launch_stem_chain_w16: 1 to 4 argument blocks per launch%s");
  const bool mg = w16_magic_ok(a_hi);
  if (n == 1) return mg ? launch_w16<2, 2, 1, true>(arr, 1, st) : launch_w16<2, 2, 1, false>(arr, 1, st);
  return mg ? launch_w16<2, 2, 4, true>(arr, n, st) : launch_w16<2, 2, 4, false>(arr, n, st);
}
