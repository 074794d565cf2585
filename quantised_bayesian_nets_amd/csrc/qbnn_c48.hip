// libqbnn_hip.so -- 16-wave fused kernels of the 48-channel layers (round 5): the 16 x 16 x 48 identity BasicBlock
// (models_bbb.py:170-183 with stride 1: stem.0 ConvReLU2d, stem.3 Conv2d, Add, ReLU) on weights in the QBNN_LAYOUT_MFMA32_N24 layout.
//
// Why a kernel of its own.  The 8-wave kernels of qbnn_blocks.hip run a 48-channel conv as passes of (32 pixels) x (2 channel tiles of
// 32): both tiles in one pass, because the window sum that the sampled weights' zero point needs (sum x'(W - z_w) = acc - z_w R) lives
// in the ones row of the LAST tile.  A pass holds 64 accumulator registers, the kernel 200+ VGPRs: two waves per SIMD, where a vector
// instruction of the exact requantisation costs 4.1 cycles of issue instead of 2.9 (profiles/r03_issue_bench.txt), and every wave is in
// the same phase -- MFMAs, barrier, epilogue -- so the matrix pipe idles through the epilogues (13 - 52 % co-execution).
// With 24 + 1 rows per tile (N24 layout) a conv splits into two independent channel halves.  Here:
//   * work item = (MC sample, 2 images); a 32-pixel fragment = output row oh of BOTH images (16 + 16 pixels), so vertically adjacent
//     fragments share two of their three kernel rows' pixel fragments, as in the 32-wide layer-1 kernel (qbnn_w16.hip);
//   * wave w owns channel half w & 1 and output rows 2 (w >> 1), 2 (w >> 1) + 1 of both convs: its half's 15 weight fragments stay in
//     registers (60 VGPRs) for the whole conv, the four input rows it needs are read from LDS once each (5 fragments per row: 144-byte
//     kernel rows padded to 160) and feed two accumulators: 30 MFMAs per 20 KiB of LDS reads;
//   * 16 waves = four per SIMD (<= 128 VGPRs): the hardware runs the older waves' MFMA streams first, so the younger waves'
//     MFMAs overlap the older waves' epilogues without any explicit pipeline;
//   * stem.3's epilogue (Add with the X tile as residual, ReLU) writes quint8 straight to HBM: no read-out pass.
// Same arithmetic and epilogue formulas as block_chain_ws_kernel / block_chain_pp_kernel: bit-identical results.
#include "qbnn_host.h"

#ifndef QBNN_C48_PRIO
#define QBNN_C48_PRIO 0
#endif
#ifndef QBNN_C48_SEQ
#define QBNN_C48_SEQ 0
#endif

// Diagnostic build only (-DQBNN_C48_STAMP, scratch library; tools/stamp_c48.py): s_memtime of every wave of workgroup 0 at the phase
// boundaries of one work item, written to a debug buffer that nothing else reads.  The shipped library contains none of this.
#ifdef QBNN_C48_STAMP
static __device__ unsigned long long* g_c48_stamp = nullptr;
QBNN_EXPORT void qbnn_debug_c48_stamp_buffer(void* p) { hipMemcpyToSymbol(HIP_SYMBOL(g_c48_stamp), &p, sizeof(p)); }
#define C48_STAMP() do { if (st_on) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    if ((threadIdx.x & 63) == 0) g_c48_stamp[(threadIdx.x >> 6) * 32 + (st_k & 31)] = t_; ++st_k; __builtin_amdgcn_sched_barrier(0); } } while (0)
#define C48_STAMP_ARGS , bool st_on, int& st_k
#define C48_STAMP_PASS , st_on, st_k
#else
#define C48_STAMP() do {} while (0)
#define C48_STAMP_ARGS
#define C48_STAMP_PASS
#endif

namespace {

constexpr int C48_THREADS = 1024, C48_WAVES = 16, C48_G = 2;
struct T48 {                                   // 16 x 16 x 48 map with a one-pixel zero halo
  // TILE_BYTES: + 64.  A fragment read (ds_read_b128) is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (x 2 halves):
  // every group mixes pixels of both images, and with the images 15552 bytes apart (48 banks mod 64) half of image 1's lanes fall on the
  // banks of image 0's -- 8 LDS cycles per read instead of 4 (tools/lds_conflicts.py; measured: 41 % of the LDS cycles were conflicts).
  static constexpr int H = 16, CH = 48, TW = 18, PIXB = 48, PITCH = TW * PIXB, TILE_BYTES = TW * TW * PIXB + 64;
  static constexpr int SPR = 5, KS = 15;       // k-steps per kernel row / per conv
  static constexpr int ONES_REG = 12;          // accumulator register of tile row 24 (lanes 0..31): the window sum
};
struct W48 { static constexpr int NT = 2, KS = T48::KS; };      // one conv's packed weights: two halves x 15 fragment tiles (dma_conv)
constexpr int C48_WHALF = T48::KS * 1024, C48_WCONV = 2 * C48_WHALF;
constexpr int C48_TILES = C48_G * T48::TILE_BYTES + 64;          // + slack: the last k-step of a tile row over-reads 16 bytes
constexpr int C48_IMG = T48::H * T48::H * T48::CH;               // bytes of one image in HBM
constexpr int c48_lds() { return 3 * C48_TILES + 2 * C48_WCONV + 2 * T48::CH * 4 + 256; }      // X (two buffers), T, both convs' weights, biases, ready flags

// tile offset of pixel (img, oh, col) -- interior coordinates
__device__ __forceinline__ int px48(int img, int oh, int col) { return img * T48::TILE_BYTES + (oh + 1) * T48::PITCH + (col + 1) * T48::PIXB; }

// stem.0's epilogue: ReLU-fused requantisation, centred bytes into the T tile
struct EpiT48 {
  static constexpr int VPM = 6;          // vector instructions of one row's epilogue per MFMA of the next row (interleave hint)
  uint8_t* tt; float vhi;
  __device__ __forceinline__ uint32_t load(int, int) const { return 0u; }
  __device__ __forceinline__ void store(int po, int, int, int, int c0, float v0, float v1, float v2, float v3, uint32_t) const {
    *reinterpret_cast<uint32_t*>(tt + po + c0) = pack_rne_u8(v0, v1, v2, v3, vhi);
  }
};
// stem.3's epilogue: requantise, quantized::add with the centred residual in the X tile, ReLU, quint8 straight to HBM
// (EpiTileResInPlace's arithmetic, EpiResToGlobal's store)
struct EpiOut48 {
  static constexpr int VPM = 12;
  const uint8_t* xt; uint8_t* y; int n_ok; QConv p; QAdd a;      // y: this item's first image in the output tensor; n_ok: images of the item that exist
  __device__ __forceinline__ uint32_t load(int po, int c0) const { return *reinterpret_cast<const uint32_t*>(xt + po + c0); }
  __device__ __forceinline__ void store(int, int img, int oh, int col, int c0, float v0, float v1, float v2, float v3, uint32_t rqu) const {
    const int rq = (int)rqu;
    const float vv[4] = {v0, v1, v2, v3};
    const float rf[4] = {(float)((rq << 24) >> 24), (float)((rq << 16) >> 24), (float)((rq << 8) >> 24), (float)(rq >> 24)};      // centred residual (any sign: the block input)
    float t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float da = __builtin_fmaf(p.s_y, __builtin_rintf(med3f(vv[i], p.vlo, p.vhi)), p.dl_y);
      const float db = __builtin_fmaf(a.s_r, rf[i], a.dl_r);
      t[i] = (da + db) * a.inv_s_o;
    }
    if (img < n_ok)
      *reinterpret_cast<uint32_t*>(y + (int64_t)img * C48_IMG + (oh * T48::H + col) * T48::CH + c0) = pack_rne_u8(t[0], t[1], t[2], t[3], a.vhi) + (uint32_t)a.z_o * 0x01010101u;
  }
};

// requantise one accumulator tile: 32 pixels (row oh of both images) x the 24 channels of `half`
template <class Epi>
__device__ __forceinline__ void epilogue48(const v16i& acc, const float4 (&b4)[3], const QConv& p, const Epi& epi, int half, int img, int oh, int col, int h) {
  const int zwr = __mul24(p.z_w, half_lo_bcast(acc[T48::ONES_REG]));      // |R| <= 432 * 127, |z_w| <= 128: a 24-bit product
  const int po = px48(img, oh, col);
  uint32_t pre[3];
#pragma unroll
  for (int g4 = 0; g4 < 3; ++g4) pre[g4] = epi.load(po, 24 * half + 8 * g4 + 4 * h);
#pragma unroll
  for (int g4 = 0; g4 < 3; ++g4) {
    const float4 bb = b4[g4];
    const float v0 = __builtin_fmaf(bb.x, p.rcp, (float)(acc[4 * g4 + 0] - zwr)) * p.mult;
    const float v1 = __builtin_fmaf(bb.y, p.rcp, (float)(acc[4 * g4 + 1] - zwr)) * p.mult;
    const float v2 = __builtin_fmaf(bb.z, p.rcp, (float)(acc[4 * g4 + 2] - zwr)) * p.mult;
    const float v3 = __builtin_fmaf(bb.w, p.rcp, (float)(acc[4 * g4 + 3] - zwr)) * p.mult;
    epi.store(po, img, oh, col, 24 * half + 8 * g4 + 4 * h, v0, v1, v2, v3, pre[g4]);
  }
}

// One 3x3 / stride-1 conv for this wave: channel half `half`, output rows 2 rp and 2 rp + 1 of both images.  `w` holds the half's 15
// weight fragments; after the MFMAs are issued it is refilled from `wnext` (the next conv's half: no barrier protects or needs it).
// `mid()` runs between the MFMAs and the epilogues (the next item's tile write: LDS stores beside the matrix pipe's tail).
template <class Epi, class Mid>
__device__ __forceinline__ void conv48_pair(const uint8_t* tile, v4i (&w)[T48::KS], const uint8_t* wnext, const float* bias_lds, const QConv& p,
                                            const Epi& epi, int half, int rp, int lane, int prank, Mid mid C48_STAMP_ARGS) {
  int l_ = lane;
  asm volatile("" : "+v"(l_));       // per-lane offsets are recomputed per phase (hoisted out of the item loop they are spilled)
  const int r = l_ & 31, h = l_ >> 5, img = r >> 4, col = r & 15, oh0 = 2 * rp;
  // tile row oh0 + j is input row oh0 - 1 + j, tile column `col` is input column col - 1: the 144-byte window of (row, col) starts there
  const uint8_t* base = tile + img * T48::TILE_BYTES + oh0 * T48::PITCH + col * T48::PIXB + 16 * h;
  const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  v16i acc0 = zero16, acc1 = zero16;
  // Staggered issue priority (-DQBNN_C48_PRIO=0 for the A/B).  All 16 waves leave a barrier together, and left alone the four waves of a SIMD
  // share the matrix pipe evenly: they finish their MFMAs together and requantise together, the pipe idle (measured: a vector instruction
  // beside only 34 % of the MFMA cycles).  With the wave's rank on its SIMD (wave >> 2: waves w, w + 4, w + 8, w + 12 share one) as its
  // priority while it multiplies, rank 0 takes the pipe alone -- two accumulator chains keep it full -- and requantises (priority 0, below
  // every multiplying wave) while rank 1 multiplies, and so on down the ranks.
#if QBNN_C48_PRIO
  if (prank == 0) __builtin_amdgcn_s_setprio(3);
  else if (prank == 1) __builtin_amdgcn_s_setprio(2);
  else if (prank == 2) __builtin_amdgcn_s_setprio(1);
#endif
#if QBNN_C48_SEQ
  // Row after row: the second row's 15 MFMAs are issued interleaved with the FIRST row's epilogue (one MFMA : VPM vector instructions,
  // sched_group_barrier), so every wave carries matrix and vector work at the same time instead of 30 MFMAs, then two epilogues -- with
  // four waves per SIMD in the same phase of a conv that is what lets the pipes overlap.  Costs the sharing of the two middle input
  // rows' fragments (30 fragment reads per row pair instead of 20).
  {
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int t = 0; t < T48::SPR; ++t)
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[kh * T48::SPR + t], *reinterpret_cast<const v4i*>(base + kh * T48::PITCH + 32 * t), acc0, 0, 0, 0);
    float4 b4s[3];
#pragma unroll
    for (int g4 = 0; g4 < 3; ++g4) b4s[g4] = *reinterpret_cast<const float4*>(bias_lds + 24 * half + 8 * g4 + 4 * h);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int t = 0; t < T48::SPR; ++t)
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[kh * T48::SPR + t], *reinterpret_cast<const v4i*>(base + (kh + 1) * T48::PITCH + 32 * t), acc1, 0, 0, 0);
    epilogue48(acc0, b4s, p, epi, half, img, oh0, col, h);
#pragma unroll
    for (int i = 0; i < T48::KS; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, Epi::VPM, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (wnext) {
#pragma unroll
      for (int ks = 0; ks < T48::KS; ++ks) w[ks] = *reinterpret_cast<const v4i*>(wnext + l_ * 16 + ks * 1024);
    }
    C48_STAMP();
    mid();
    C48_STAMP();
    C48_STAMP();
    epilogue48(acc1, b4s, p, epi, half, img, oh0 + 1, col, h);
    C48_STAMP();
    return;
  }
#endif
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v4i x[T48::SPR];
#pragma unroll
    for (int t = 0; t < T48::SPR; ++t) x[t] = *reinterpret_cast<const v4i*>(base + j * T48::PITCH + 32 * t);
#pragma unroll
    for (int t = 0; t < T48::SPR; ++t) {
      if (j <= 2) acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[j * T48::SPR + t], x[t], acc0, 0, 0, 0);
      if (j >= 1) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[(j - 1) * T48::SPR + t], x[t], acc1, 0, 0, 0);
    }
  }
  if (wnext) {
#pragma unroll
    for (int ks = 0; ks < T48::KS; ++ks) w[ks] = *reinterpret_cast<const v4i*>(wnext + l_ * 16 + ks * 1024);
  }
  C48_STAMP();          // MFMAs issued (not yet complete)
#if QBNN_C48_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  mid();
  C48_STAMP();
  float4 b4[3];
#pragma unroll
  for (int g4 = 0; g4 < 3; ++g4) b4[g4] = *reinterpret_cast<const float4*>(bias_lds + 24 * half + 8 * g4 + 4 * h);
  epilogue48(acc0, b4, p, epi, half, img, oh0, col, h);
  C48_STAMP();
  epilogue48(acc1, b4, p, epi, half, img, oh0 + 1, col, h);
  C48_STAMP();
}

template <int NM>
__global__ __launch_bounds__(C48_THREADS) void chain48_w16_kernel(const ArgsArr<ChainArgs<1>, NM> all) {
  const ChainArgs<1> a = args_of(all, blockIdx.y);
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* xt0 = smem;                                      // centred block input (the residual), two buffers: item i + 1's tile is written while
                                                            // item i's second conv still reads item i's as its residual
  uint8_t* tt = smem + 2 * C48_TILES;                       // centred stem.0 output
  uint8_t* wl = smem + 3 * C48_TILES;                       // [conv a | conv b] x [half 0 | half 1] x 15 fragment tiles
  float* bias_lds = reinterpret_cast<float*>(wl + 2 * C48_WCONV);      // [2][48]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave & 1, rp = wave >> 1;

  const int groups = (a.B + C48_G - 1) / C48_G;
  int begin, count;
  item_range(a.n_samples * groups, blockIdx.x, gridDim.x, begin, count);
  if (count <= 0) return;

  zero_halo<T48::TW, T48::PIXB, T48::TILE_BYTES, C48_G, C48_THREADS>(xt0, tid);
  zero_halo<T48::TW, T48::PIXB, T48::TILE_BYTES, C48_G, C48_THREADS>(xt0 + C48_TILES, tid);
  zero_halo<T48::TW, T48::PIXB, T48::TILE_BYTES, C48_G, C48_THREADS>(tt, tid);
  load_bias<T48::CH, C48_THREADS>(bias_lds, a.blk[0].a.bias, tid);
  load_bias<T48::CH, C48_THREADS>(bias_lds + T48::CH, a.blk[0].b.bias, tid);

  // the item's input: 2 x 12288 bytes = 1536 16-byte chunks over 1024 threads, fetched one item ahead
  constexpr int CPI = C48_IMG / 16, NCH = C48_G * CPI, PER_T = (NCH + C48_THREADS - 1) / C48_THREADS;
  v4i pre[PER_T];
  auto fetch = [&](int item) {
    const int s = item / groups, img0 = (item - s * groups) * C48_G;
    const uint8_t* xs = a.x + (int64_t)s * a.x_ss;
    int t_ = tid;
    asm volatile("" : "+v"(t_));
#pragma unroll
    for (int j = 0; j < PER_T; ++j) {
      const int i = t_ + j * C48_THREADS;
      const int g = i / CPI, rem = i - g * CPI;
      const int gi = img0 + g < a.B ? img0 + g : a.B - 1;             // (a missing second image: any valid address, its results are not stored)
      const bool ok = i < NCH;
      pre[j] = *reinterpret_cast<const v4i*>(xs + (ok ? (int64_t)gi * C48_IMG + (int64_t)rem * 16 : 0));
    }
  };
  auto write_x = [&](uint8_t* xt) {
    const uint32_t z4 = (uint32_t)a.z_in * 0x01010101u;
    int t_ = tid;
    asm volatile("" : "+v"(t_));
#pragma unroll
    for (int j = 0; j < PER_T; ++j) {
      const int i = t_ + j * C48_THREADS;
      if (i < NCH) {
        const int g = i / CPI, rem = i - g * CPI, row = rem / (T48::H * T48::CH / 16), within = rem - row * (T48::H * T48::CH / 16);
        const v4i v = pre[j];
        *reinterpret_cast<v4i*>(xt + g * T48::TILE_BYTES + (row + 1) * T48::PITCH + T48::PIXB + within * 16) =
            v4i{(int)sub_bytes(v.x, z4), (int)sub_bytes(v.y, z4), (int)sub_bytes(v.z, z4), (int)sub_bytes(v.w, z4)};
      }
    }
  };
  fetch(begin);
  write_x(xt0);
  fetch(count > 1 ? begin + 1 : begin);

  v4i w[T48::KS];
  int cur_s = -1;
  for (int it = 0; it < count; ++it) {
    const int item = begin + it;
    const int s = item / groups, img0 = (item - s * groups) * C48_G;
    uint8_t* xt = xt0 + (it & 1) * C48_TILES;
    uint8_t* xn = xt0 + ((it + 1) & 1) * C48_TILES;
    if (s != cur_s) {            // workgroup-uniform; a few times per launch
      __syncthreads();           // every wave is done with the previous sample's weights (they are read into registers before each conv's barrier)
      int l_ = lane;
      asm volatile("" : "+v"(l_));
      dma_conv<W48, C48_WAVES>(wl, a.blk[0].a.w + (int64_t)s * a.blk[0].a.w_ss, wave, l_);
      dma_conv<W48, C48_WAVES>(wl + C48_WCONV, a.blk[0].b.w + (int64_t)s * a.blk[0].b.w_ss, wave, l_);
      dma_barrier();             // vmcnt(0) + barrier: the weights have landed
      cur_s = s;
    }
    {
      int l_ = lane;
      asm volatile("" : "+v"(l_));
#pragma unroll
      for (int ks = 0; ks < T48::KS; ++ks) w[ks] = *reinterpret_cast<const v4i*>(wl + half * C48_WHALF + l_ * 16 + ks * 1024);
    }
#ifdef QBNN_C48_STAMP
    const bool st_on = g_c48_stamp != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && it == 3;
    int st_k = 0;
#endif
    C48_STAMP();
    lds_barrier();               // this item's X tile is complete (written during the previous item's second conv), T is free
    C48_STAMP();
    conv48_pair(xt, w, wl + C48_WCONV + half * C48_WHALF, bias_lds, a.blk[0].a, EpiT48{tt, a.blk[0].a.vhi}, half, rp, lane, wave >> 2, [] {} C48_STAMP_PASS);
    lds_barrier();               // T complete
    C48_STAMP();
    conv48_pair(tt, w, nullptr, bias_lds + T48::CH, a.blk[0].b,
                EpiOut48{xt, a.y + (int64_t)s * a.y_ss + (int64_t)img0 * C48_IMG, a.B - img0, a.blk[0].b, a.blk[0].add}, half, rp, lane, wave >> 2,
                [&] {            // the other X buffer (last read as item it - 1's residual, two barriers ago) <- item it + 1; then item it + 2 into registers
                  if (it + 1 < count) write_x(xn);
                  fetch(it + 2 < count ? item + 2 : item);
                } C48_STAMP_PASS);
    C48_STAMP();
  }
}

// ---- the same block WITHOUT workgroup barriers inside an item: ready flags between the stages (round 5) ---------------------------------
// Why: with a barrier behind every conv the last wave's epilogue -- alone on its SIMD, 10 - 18 cycles per vector instruction -- is the tail of
// every phase (profiles/r05_stamp_c48_w16.txt: 36 % / 58 % of the two phases).  Here a wave waits only for the waves whose data it reads:
//   xr[rp][img] = i + 1 : rows 2 rp, 2 rp + 1 of image `img` of item i are in X[i & 1]        (written by wave (rp, half = img))
//   ta[rp][half] = i + 1 : stem.0's output rows 2 rp, 2 rp + 1, channels of `half`, item i, are in T
//   bm[rp][half] = i + 1 : that wave's stem.3 MFMAs of item i have read T (rows 2 rp - 1 .. 2 rp + 2)
//   be[rp][half] = i + 1 : its stem.3 epilogue of item i has read the residual from X[i & 1]
// and per item: wait xr[rp-1..rp+1] >= i + 1 | MFMAs a | wait bm[rp-1..rp+1] >= i | epilogue a -> T, ta = i + 1 | wait ta[rp-1..rp+1] >= i + 1 |
// MFMAs b, bm = i + 1 | wait be[rp][other half] >= i, item i + 1's rows -> X[(i + 1) & 1], xr = i + 2 | epilogue b -> HBM, be = i + 1.
// Every wait names work that lies earlier in program order of the wave it waits for, so the waves cannot wait in a cycle; the flags only grow.
// The waves of a SIMD then stay one MFMA phase apart instead of meeting at a barrier twice per item.  A wait polls LDS with s_sleep between
// polls and gives up after 2^20 polls (wrong results that the parity tests catch, never a hung GPU).  The sample change keeps its barrier.
struct DFlags { int xr[16], ta[16], bm[16], be[16]; };
__device__ __forceinline__ void df_wait(const int* f, int lo, int hi, int need) {      // f[2 lo .. 2 hi + 1] >= need
#pragma unroll 1
  for (int spin = 0; spin < (1 << 20); ++spin) {
    int mn = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int idx = 2 * lo + q;
      const int v = idx <= 2 * hi + 1 ? __hip_atomic_load(f + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0x7fffffff;
      mn = v < mn ? v : mn;
    }
    if (__builtin_amdgcn_readfirstlane(mn) >= need) break;
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void df_wait1(const int* f, int need) {
#pragma unroll 1
  for (int spin = 0; spin < (1 << 20); ++spin) {
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >= need) break;
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void df_set(int* f, int v) {      // (every lane stores the same value)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int NM>
__global__ __launch_bounds__(C48_THREADS) void chain48_df_kernel(const ArgsArr<ChainArgs<1>, NM> all) {
  const ChainArgs<1> a = args_of(all, blockIdx.y);
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* xt0 = smem;
  uint8_t* tt = smem + 2 * C48_TILES;
  uint8_t* wl = smem + 3 * C48_TILES;
  float* bias_lds = reinterpret_cast<float*>(wl + 2 * C48_WCONV);
  DFlags* fl = reinterpret_cast<DFlags*>(bias_lds + 2 * T48::CH);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave & 1, rp = wave >> 1, lo = rp > 0 ? rp - 1 : 0, hi = rp < 7 ? rp + 1 : 7, me = 2 * rp + half;

  const int groups = (a.B + C48_G - 1) / C48_G;
  int begin, count;
  item_range(a.n_samples * groups, blockIdx.x, gridDim.x, begin, count);
  if (count <= 0) return;

  zero_halo<T48::TW, T48::PIXB, T48::TILE_BYTES, C48_G, C48_THREADS>(xt0, tid);
  zero_halo<T48::TW, T48::PIXB, T48::TILE_BYTES, C48_G, C48_THREADS>(xt0 + C48_TILES, tid);
  zero_halo<T48::TW, T48::PIXB, T48::TILE_BYTES, C48_G, C48_THREADS>(tt, tid);
  load_bias<T48::CH, C48_THREADS>(bias_lds, a.blk[0].a.bias, tid);
  load_bias<T48::CH, C48_THREADS>(bias_lds + T48::CH, a.blk[0].b.bias, tid);
  if (tid < 64) reinterpret_cast<int*>(fl)[tid] = tid < 16 ? 1 : 0;      // xr = 1: item 0's rows are written below, in front of the first barrier

  // this wave's two rows of image `half`: 2 x 768 bytes = 96 16-byte chunks over 64 lanes, fetched one item ahead
  constexpr int RCH = T48::H * T48::CH / 16;        // chunks per image row
  v4i pre[2];
  auto fetch_rows = [&](int item) {
    const int s = item / groups, img0 = (item - s * groups) * C48_G;
    const int gi = img0 + half < a.B ? img0 + half : a.B - 1;             // (a missing second image: any valid address, its results are not stored)
    int l_ = lane;
    asm volatile("" : "+v"(l_));
    const uint8_t* src = a.x + (int64_t)s * a.x_ss + (int64_t)gi * C48_IMG + (2 * rp) * (RCH * 16) + l_ * 16;
    pre[0] = *reinterpret_cast<const v4i*>(src);
    pre[1] = *reinterpret_cast<const v4i*>(src + (l_ < 32 ? 64 * 16 : 0));
  };
  auto write_rows = [&](uint8_t* xt) {
    const uint32_t z4 = (uint32_t)a.z_in * 0x01010101u;
    int l_ = lane;
    asm volatile("" : "+v"(l_));
    uint8_t* dst = xt + half * T48::TILE_BYTES + (2 * rp + 1) * T48::PITCH + T48::PIXB;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = l_ + 64 * j;
      if (c < 2 * RCH) {
        const int r = c >= RCH ? 1 : 0, within = c - r * RCH;
        const v4i v = pre[j];
        *reinterpret_cast<v4i*>(dst + r * T48::PITCH + within * 16) =
            v4i{(int)sub_bytes(v.x, z4), (int)sub_bytes(v.y, z4), (int)sub_bytes(v.z, z4), (int)sub_bytes(v.w, z4)};
      }
    }
  };
  fetch_rows(begin);
  write_rows(xt0);
  fetch_rows(count > 1 ? begin + 1 : begin);

  v4i w[T48::KS];
  int cur_s = -1;
#ifdef QBNN_C48_STAMP
  const bool st_on = false;
  int st_k = 0;
#endif
  for (int it = 0; it < count; ++it) {
    const int item = begin + it;
    const int s = item / groups, img0 = (item - s * groups) * C48_G;
    uint8_t* xt = xt0 + (it & 1) * C48_TILES;
    uint8_t* xn = xt0 + ((it + 1) & 1) * C48_TILES;
    if (s != cur_s) {            // workgroup-uniform; a few times per launch (and at the first item: publishes the prologue's LDS writes)
      __syncthreads();
      int l_ = lane;
      asm volatile("" : "+v"(l_));
      dma_conv<W48, C48_WAVES>(wl, a.blk[0].a.w + (int64_t)s * a.blk[0].a.w_ss, wave, l_);
      dma_conv<W48, C48_WAVES>(wl + C48_WCONV, a.blk[0].b.w + (int64_t)s * a.blk[0].b.w_ss, wave, l_);
      dma_barrier();
      cur_s = s;
    }
    {
      int l_ = lane;
      asm volatile("" : "+v"(l_));
#pragma unroll
      for (int ks = 0; ks < T48::KS; ++ks) w[ks] = *reinterpret_cast<const v4i*>(wl + half * C48_WHALF + l_ * 16 + ks * 1024);
    }
    df_wait(fl->xr, lo, hi, it + 1);
    conv48_pair(xt, w, wl + C48_WCONV + half * C48_WHALF, bias_lds, a.blk[0].a, EpiT48{tt, a.blk[0].a.vhi}, half, rp, lane, wave >> 2,
                [&] { df_wait(fl->bm, lo, hi, it); } C48_STAMP_PASS);
    df_set(fl->ta + me, it + 1);
    df_wait(fl->ta, lo, hi, it + 1);
    conv48_pair(tt, w, nullptr, bias_lds + T48::CH, a.blk[0].b,
                EpiOut48{xt, a.y + (int64_t)s * a.y_ss + (int64_t)img0 * C48_IMG, a.B - img0, a.blk[0].b, a.blk[0].add}, half, rp, lane, wave >> 2,
                [&] {
                  df_set(fl->bm + me, it + 1);
                  if (it + 1 < count) {
                    df_wait1(fl->be + (me ^ 1), it);
                    write_rows(xn);
                    df_set(fl->xr + me, it + 2);
                  }
                  fetch_rows(it + 2 < count ? item + 2 : item);
                } C48_STAMP_PASS);
    df_set(fl->be + me, it + 1);
  }
}

static bool c48_dataflow() {
  static const bool v = [] { const char* e = getenv("QBNN_C48_DF"); return e && e[0] == '1'; }();      // off: measured 6 % slower than the barrier form (profiles/r05_stamp_c48_w16.txt)
  return v;
}

template <int NM>
int launch_c48(const ChainArgs<1>* arr, int n, hipStream_t st) {
  constexpr int LDS = c48_lds();
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static_assert(sizeof(ArgsArr<ChainArgs<1>, NM>) <= 3840, "kernel arguments are limited to 4 KiB (incl. the hidden ones)");
  static std::atomic<uint64_t> attr{0}, attr_df{0};
  if (c48_dataflow()) {
    if (int rc_attr = ensure_dyn_lds((const void*)chain48_df_kernel<NM>, attr_df, LDS)) return rc_attr;
    ArgsArr<ChainArgs<1>, NM> all;
    memset(&all, 0, sizeof(all));
    int items = 0;
    for (int i = 0; i < n; ++i) {
      all.m[i] = arr[i];
      const int it = arr[i].n_samples * ((arr[i].B + C48_G - 1) / C48_G);
      items = it > items ? it : items;
    }
    const int per = 256 / n > 0 ? 256 / n : 1;
    const int gx = items < per ? (items > 0 ? items : 1) : per;
    hipLaunchKernelGGL((chain48_df_kernel<NM>), dim3(gx, n), dim3(C48_THREADS), LDS, st, all);
    return check_launch("qbnn_block_chain_i8_mc (48 channels, N24 layout, ready flags)");
  }
  if (int rc_attr = ensure_dyn_lds((const void*)chain48_w16_kernel<NM>, attr, LDS)) return rc_attr;
  ArgsArr<ChainArgs<1>, NM> all;
  memset(&all, 0, sizeof(all));
  int items = 0;
  for (int i = 0; i < n; ++i) {
    all.m[i] = arr[i];
    const int it = arr[i].n_samples * ((arr[i].B + C48_G - 1) / C48_G);
    items = it > items ? it : items;
  }
  const int per = 256 / n > 0 ? 256 / n : 1;
  const int gx = items < per ? (items > 0 ? items : 1) : per;
  hipLaunchKernelGGL((chain48_w16_kernel<NM>), dim3(gx, n), dim3(C48_THREADS), LDS, st, all);
  return check_launch("qbnn_block_chain_i8_mc (48 channels, N24 layout)");
}

}  // namespace

// entry points for qbnn_blocks.hip (declared in qbnn_host.h)
int qbnn_launch_chain48_w16(const ChainArgs<1>* arr, int n, hipStream_t st) {
  if (n < 1 || n > QBNN_FUSED_CALLS) return fail(QBNN_E_INVALID, "qbnn_launch_chain48_w16: 1 to 8 argument blocks per launch%s");
  if (n == 1) return launch_c48<1>(arr, 1, st);
  return launch_c48<QBNN_FUSED_CALLS>(arr, n, st);
}
int qbnn_launch_chain48_w16_dev(const ChainArgs<1>* dev, int n, int items, hipStream_t st) {
  constexpr int LDS = c48_lds();
  static std::atomic<uint64_t> attr{0}, attr_df{0};
  const int per = 256 / n > 0 ? 256 / n : 1;
  const int gx = items < per ? (items > 0 ? items : 1) : per;
  if (c48_dataflow()) {
    if (int rc_attr = ensure_dyn_lds((const void*)chain48_df_kernel<0>, attr_df, LDS)) return rc_attr;
    hipLaunchKernelGGL((chain48_df_kernel<0>), dim3(gx, n), dim3(C48_THREADS), LDS, st, ArgsArr<ChainArgs<1>, 0>{dev});
    return check_launch("qbnn_block_chain_i8_multi_launch (48 channels, N24 layout, ready flags)");
  }
  if (int rc_attr = ensure_dyn_lds((const void*)chain48_w16_kernel<0>, attr, LDS)) return rc_attr;
  hipLaunchKernelGGL((chain48_w16_kernel<0>), dim3(gx, n), dim3(C48_THREADS), LDS, st, ArgsArr<ChainArgs<1>, 0>{dev});
  return check_launch("qbnn_block_chain_i8_multi_launch (48 channels, N24 layout)");
}
