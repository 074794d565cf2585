// libqbnn_hip.so -- 16-wave fused kernels of the layers with 48 OUTPUT channels (round 5), on weights whose fragment tiles hold 24 output
// channels + a ones row (QBNN_LAYOUT_MFMA32_N24):
//   chain48_w16_kernel : the 16 x 16 x 48 identity BasicBlock (models_bbb.py:170-183 with stride 1: stem.0 ConvReLU2d, stem.3 Conv2d, Add, ReLU)
//   down24_w16_kernel  : the 24 -> 48 down-sampling BasicBlock (:146-183 with stride 2: shortcut 1x1/s2, stem.0 3x3/s2 ConvReLU, stem.3 3x3, Add, ReLU)
//
// Why kernels of their own.  The 8-wave kernels of qbnn_blocks.hip run a 48-channel conv as passes of (32 pixels) x (2 channel tiles of
// 32): both tiles in one pass, because the window sum that the sampled weights' zero point needs (sum x'(W - z_w) = acc - z_w R) lives
// in the ones row of the LAST tile.  A pass holds 64 accumulator registers, the kernel 200+ VGPRs: two waves per SIMD, where a vector
// instruction of the exact requantisation costs 4.1 cycles of issue instead of 2.9 (profiles/r03_issue_bench.txt).  With 24 + 1 rows per tile
// a conv splits into two independent channel halves, a wave's unit of work is (one output row of TWO images = 32 pixels) x (24 channels),
// 16 accumulator registers, and 16 waves fit a CU:
//   * work item = (MC sample, 2 images); a 32-pixel fragment = output row oh of BOTH images (16 + 16 pixels);
//   * wave w owns channel half w & 1 and output rows 2 (w >> 1), 2 (w >> 1) + 1 of every conv of the block; a conv's weight fragments of that
//     half stay in registers for the whole conv (15 fragments = 60 VGPRs for a 48 -> 48 conv; 7 for the 24 -> 48 one in the TAIL form);
//   * in a 48 -> 48 conv the wave's two rows share two of the four input rows they read: 5 fragments per row, 30 MFMAs per 20 KiB of LDS reads;
//   * the block's last epilogue (Add, ReLU) writes quint8 straight to HBM; in the down block the shortcut's requantised bytes never leave
//     the wave's registers (its unit covers the same pixels and channels as the wave's stem.3 unit);
//   * the 24 -> 48 block's first two convs (K = 216 and 24) start their accumulators at 1.5 * 2^23, as the layer-1 kernel does (qbnn_w16.hip).
// Same arithmetic and epilogue formulas as block_chain_pp_kernel / block_down_ws_kernel: bit-identical results.
// What was measured on the way (profiles/r05_stamp_c48_w16.txt; sources: tools/experiments/r05_c48_variants.hip.txt): staggered s_setprio,
// a row-after-row MFMA / epilogue interleave and a barrier-free ready-flag form of the chain kernel -- equal, 12 % slower, 6 % slower.
#include "qbnn_host.h"

// Diagnostic build only (-DQBNN_C48_STAMP, scratch library; tools/stamp_c48.py): s_memtime of every wave of workgroup 0 at the phase
// boundaries of one work item of the chain kernel, written to a debug buffer that nothing else reads.  The shipped library contains none of this.
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
  static constexpr int SPR = 5, KS = 15;       // k-steps per kernel row / per conv (144-byte kernel rows padded to 160)
  static constexpr int ONES_REG = 12;          // accumulator register of tile row 24 (lanes 0..31): the window sum
};
struct W48 { static constexpr int NT = 2, KS = T48::KS; };      // one 48 -> 48 conv's packed weights: two halves x 15 fragment tiles (dma_conv)
constexpr int C48_WHALF = T48::KS * 1024, C48_WCONV = 2 * C48_WHALF;
constexpr int C48_TILES = C48_G * T48::TILE_BYTES + 64;          // + slack: the last k-step of a tile row over-reads 16 bytes
constexpr int C48_IMG = T48::H * T48::H * T48::CH;               // bytes of one 16 x 16 x 48 image in HBM
constexpr int c48_lds() { return 3 * C48_TILES + 2 * C48_WCONV + 2 * T48::CH * 4; }      // X (two buffers), T, both convs' weights, biases

// tile offset of pixel (img, oh, col) -- interior coordinates
__device__ __forceinline__ int px48(int img, int oh, int col) { return img * T48::TILE_BYTES + (oh + 1) * T48::PITCH + (col + 1) * T48::PIXB; }

// stem.0's epilogue: ReLU-fused requantisation, centred bytes into the T tile
struct EpiT48 {
  uint8_t* tt; float vhi;
  __device__ __forceinline__ uint32_t load(int, int) const { return 0u; }
  __device__ __forceinline__ void store(int po, int, int, int, int c0, float v0, float v1, float v2, float v3, uint32_t) const {
    *reinterpret_cast<uint32_t*>(tt + po + c0) = pack_rne_u8(v0, v1, v2, v3, vhi);
  }
};
// stem.3's epilogue: requantise, quantized::add with the residual, ReLU, quint8 straight to HBM (EpiTileResInPlace's arithmetic, EpiResToGlobal's
// store).  RES_REG = false: the residual is the centred block input in the X tile (any sign); true: the shortcut's quint8 bytes, handed in from the
// wave's registers (the down block: EpiDense<HAS_RES>'s arithmetic).
template <bool RES_REG>
struct EpiOut48 {
  const uint8_t* xt; uint8_t* y; int n_ok; QConv p; QAdd a;      // y: this item's first image in the output tensor; n_ok: images of the item that exist
  __device__ __forceinline__ uint32_t load(int po, int c0) const { return RES_REG ? 0u : *reinterpret_cast<const uint32_t*>(xt + po + c0); }
  __device__ __forceinline__ void store(int, int img, int oh, int col, int c0, float v0, float v1, float v2, float v3, uint32_t rqu) const {
    const int rq = (int)rqu;
    const float vv[4] = {v0, v1, v2, v3};
    float rf[4];
    if constexpr (RES_REG) { rf[0] = (float)(rqu & 0xffu); rf[1] = (float)((rqu >> 8) & 0xffu); rf[2] = (float)((rqu >> 16) & 0xffu); rf[3] = (float)(rqu >> 24); }
    else { rf[0] = (float)((rq << 24) >> 24); rf[1] = (float)((rq << 16) >> 24); rf[2] = (float)((rq << 8) >> 24); rf[3] = (float)(rq >> 24); }
    float t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float da = __builtin_fmaf(p.s_y, __builtin_rintf(med3f(vv[i], p.vlo, p.vhi)), p.dl_y);
      const float db = __builtin_fmaf(a.s_r, rf[i], RES_REG ? a.nzs_r : a.dl_r);
      t[i] = (da + db) * a.inv_s_o;
    }
    if (img < n_ok)
      *reinterpret_cast<uint32_t*>(y + (int64_t)img * C48_IMG + (oh * T48::H + col) * T48::CH + c0) = pack_rne_u8(t[0], t[1], t[2], t[3], a.vhi) + (uint32_t)a.z_o * 0x01010101u;
  }
};

// requantise one accumulator tile: 32 pixels (row oh of both images) x the 24 channels of `half`; `resreg`: the residual dwords (else epi.load)
template <class Epi>
__device__ __forceinline__ void epilogue48(const v16i& acc, const float4 (&b4)[3], const QConv& p, const Epi& epi, int half, int img, int oh, int col, int h,
                                           const uint32_t* resreg = nullptr) {
  const int zwr = __mul24(p.z_w, half_lo_bcast(acc[T48::ONES_REG]));      // |R| <= 432 * 127, |z_w| <= 128: a 24-bit product
  const int po = px48(img, oh, col);
  uint32_t pre[3];
#pragma unroll
  for (int g4 = 0; g4 < 3; ++g4) pre[g4] = resreg ? resreg[g4] : epi.load(po, 24 * half + 8 * g4 + 4 * h);
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

// One 3x3 / stride-1 48 -> 48 conv for this wave: channel half `half`, output rows 2 rp and 2 rp + 1 of both images, over a T48-shaped tile.
// `w` holds the half's 15 weight fragments; after the MFMAs are issued it is refilled from `wnext` (the next conv's half: no barrier protects or
// needs it).  `mid()` runs between the MFMAs and the epilogues (the next item's tile write: LDS stores beside the matrix pipe's tail).
// `scr`: the residual dwords of the two rows in registers (the down block), else the epilogue functor loads them.
template <class Epi, class Mid>
__device__ __forceinline__ void conv48_pair(const uint8_t* tile, v4i (&w)[T48::KS], const uint8_t* wnext, const float* bias_lds, const QConv& p,
                                            const Epi& epi, int half, int rp, int lane, Mid mid, const uint32_t (*scr)[3] C48_STAMP_ARGS) {
  int l_ = lane;
  asm volatile("" : "+v"(l_));       // per-lane offsets are recomputed per phase (hoisted out of the item loop they are spilled)
  const int r = l_ & 31, h = l_ >> 5, img = r >> 4, col = r & 15, oh0 = 2 * rp;
  // tile row oh0 + j is input row oh0 - 1 + j, tile column `col` is input column col - 1: the 144-byte window of (row, col) starts there
  const uint8_t* base = tile + img * T48::TILE_BYTES + oh0 * T48::PITCH + col * T48::PIXB + 16 * h;
  const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  v16i acc0 = zero16, acc1 = zero16;
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
  mid();
  C48_STAMP();
  // (the bias table is read HERE: its address is made opaque at this point -- typed as float, the reads are otherwise hoisted above the tile
  //  stores of the phase before, a barrier earlier, and their 12 registers parked in scratch across it)
  int boff = 24 * half + 4 * h;
  asm volatile("" : "+v"(boff));
  float4 b4[3];
#pragma unroll
  for (int g4 = 0; g4 < 3; ++g4) b4[g4] = *reinterpret_cast<const float4*>(bias_lds + boff + 8 * g4);
  epilogue48(acc0, b4, p, epi, half, img, oh0, col, h, scr ? scr[0] : nullptr);
  C48_STAMP();
  epilogue48(acc1, b4, p, epi, half, img, oh0 + 1, col, h, scr ? scr[1] : nullptr);
  C48_STAMP();
}

// ================================================================================================================================
// identity block, 48 channels
// ================================================================================================================================
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
    conv48_pair(xt, w, wl + C48_WCONV + half * C48_WHALF, bias_lds, a.blk[0].a, EpiT48{tt, a.blk[0].a.vhi}, half, rp, lane, [] {}, nullptr C48_STAMP_PASS);
    lds_barrier();               // T complete
    C48_STAMP();
    conv48_pair(tt, w, nullptr, bias_lds + T48::CH, a.blk[0].b,
                EpiOut48<false>{xt, a.y + (int64_t)s * a.y_ss + (int64_t)img0 * C48_IMG, a.B - img0, a.blk[0].b, a.blk[0].add}, half, rp, lane,
                [&] {            // the other X buffer (last read as item it - 1's residual, two barriers ago) <- item it + 1; then item it + 2 into registers
                  if (it + 1 < count) write_x(xn);
                  fetch(it + 2 < count ? item + 2 : item);
                }, nullptr C48_STAMP_PASS);
    C48_STAMP();
  }
}

// ================================================================================================================================
// down-sampling block 24 -> 48 (32 x 32 -> 16 x 16)
//   LDS: X = the two images' centred 34 x 34 x 24 tiles (zero halo), T = stem.0's centred output (the chain kernel's 18 x 18 x 48 tile pair),
//        the three convs' weights: stem.0 as N24 + TAIL fragments (2 halves x 7 k-steps), the shortcut as N24 (2 x 1), stem.3 as N24 (2 x 15).
//   Per item:  barrier | per output row of the wave: shortcut (1 MFMA -> quint8 dwords that STAY in registers), stem.0 (7 MFMAs -> T) |
//              barrier | stem.3 (30 MFMAs for the row pair), the next item's X tile written behind its MFMAs (X is not read in this phase),
//              epilogue: + shortcut registers, ReLU -> HBM.
// ================================================================================================================================
#ifndef QBNN_D24_B128
#define QBNN_D24_B128 1
#endif
#ifndef QBNN_X24_PAD
#define QBNN_X24_PAD (QBNN_D24_B128 ? 160 : 0)      // image stride: the two images' ds_read_b128 lane groups on disjoint banks (tools/lds_conflicts.py: 8 -> 4 cycles)
#endif
struct X24 {                                   // 32 x 32 x 24 map with a one-pixel zero halo (the layer-1 kernel's tile)
  static constexpr int H = 32, CH = 24, TW = 34, PIXB = 24, PITCH = TW * PIXB, TILE_BYTES = TW * TW * PIXB, STRIDE = TILE_BYTES + QBNN_X24_PAD;      // (image to image)
};
constexpr int D24_KSA = 7, D24_WA_HALF = D24_KSA * 1024, D24_WS_HALF = 1024;
struct WA24 { static constexpr int NT = 2, KS = D24_KSA; };
struct WS24 { static constexpr int NT = 2, KS = 1; };
constexpr int D24_XT = C48_G * X24::STRIDE + 2 * X24::PITCH + 64;      // + slack: the tail fragment of the last row pair reads one row past the tile
constexpr int D24_IMG_IN = X24::H * X24::H * X24::CH;
constexpr int d24_lds() { return D24_XT + C48_TILES + 2 * D24_WA_HALF + 2 * D24_WS_HALF + C48_WCONV + 3 * T48::CH * 4; }
#define QBNN_MAGIC_BITS48 0x4B400000

// 1.5 * 2^23 + z_w R for an accumulator started at 1.5 * 2^23: as_float(acc) - that = (float)(sum - z_w R) exactly (qbnn_w16.hip: epilogue24, MAGIC)
__device__ __forceinline__ float magic_c(const v16i& acc, const QConv& p) {
  return __builtin_fmaf((float)p.z_w, __int_as_float(half_lo_bcast(acc[T48::ONES_REG])) - QBNN_MAGIC, QBNN_MAGIC);
}

template <int NM>
__global__ __launch_bounds__(C48_THREADS) void down24_w16_kernel(const ArgsArr<DownArgs, NM> all) {
  const DownArgs a = args_of(all, blockIdx.y);
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* xt = smem;                                       // centred block input, two 34 x 34 x 24 tiles
  uint8_t* tt = smem + D24_XT;                              // centred stem.0 output (T48 tiles)
  uint8_t* wa = tt + C48_TILES;                             // stem.0:   [half][7 fragment tiles]
  uint8_t* ws = wa + 2 * D24_WA_HALF;                       // shortcut: [half][1]
  uint8_t* wb = ws + 2 * D24_WS_HALF;                       // stem.3:   [half][15]
  float* bias_lds = reinterpret_cast<float*>(wb + C48_WCONV);      // [3][48]: s, a, b
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave & 1, rp = wave >> 1;

  const int groups = (a.B + C48_G - 1) / C48_G;
  int begin, count;
  item_range(a.n_samples * groups, blockIdx.x, gridDim.x, begin, count);
  if (count <= 0) return;

  zero_halo<X24::TW, X24::PIXB, X24::STRIDE, C48_G, C48_THREADS>(xt, tid);
  zero_halo<T48::TW, T48::PIXB, T48::TILE_BYTES, C48_G, C48_THREADS>(tt, tid);
  load_bias<T48::CH, C48_THREADS>(bias_lds, a.s.bias, tid);
  load_bias<T48::CH, C48_THREADS>(bias_lds + T48::CH, a.a.bias, tid);
  load_bias<T48::CH, C48_THREADS>(bias_lds + 2 * T48::CH, a.b.bias, tid);

  // the item's input: 2 x 24576 bytes = 3072 16-byte chunks over 1024 threads (an image row is 768 bytes = 48 chunks), fetched one item ahead
  constexpr int CPI = D24_IMG_IN / 16, RCH = X24::H * X24::CH / 16, PER_T = C48_G * CPI / C48_THREADS;
  static_assert(C48_G * CPI == PER_T * C48_THREADS, "whole chunks per thread");
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
      const int gi = img0 + g < a.B ? img0 + g : a.B - 1;
      pre[j] = *reinterpret_cast<const v4i*>(xs + (int64_t)gi * D24_IMG_IN + (int64_t)rem * 16);
    }
  };
  auto write_x = [&]() {
    const uint32_t z4 = (uint32_t)a.z_in * 0x01010101u;
    int t_ = tid;
    asm volatile("" : "+v"(t_));
#pragma unroll
    for (int j = 0; j < PER_T; ++j) {
      const int i = t_ + j * C48_THREADS;
      const int g = i / CPI, rem = i - g * CPI, row = rem / RCH, within = rem - row * RCH;
      const v4i v = pre[j];
      uint8_t* d = xt + g * X24::STRIDE + (row + 1) * X24::PITCH + X24::PIXB + within * 16;      // (24-byte pixels: 8-byte aligned)
      *reinterpret_cast<v2i*>(d) = v2i{(int)sub_bytes(v.x, z4), (int)sub_bytes(v.y, z4)};
      *reinterpret_cast<v2i*>(d + 8) = v2i{(int)sub_bytes(v.z, z4), (int)sub_bytes(v.w, z4)};
    }
  };
  fetch(begin);
  write_x();
  fetch(count > 1 ? begin + 1 : begin);

  int cur_s = -1;
  for (int it = 0; it < count; ++it) {
    const int item = begin + it;
    const int s = item / groups, img0 = (item - s * groups) * C48_G;
    if (s != cur_s) {            // workgroup-uniform; a few times per launch
      __syncthreads();
      int l_ = lane;
      asm volatile("" : "+v"(l_));
      dma_conv<WA24, C48_WAVES>(wa, a.a.w + (int64_t)s * a.a.w_ss, wave, l_);
      dma_conv<WS24, C48_WAVES>(ws, a.s.w + (int64_t)s * a.s.w_ss, wave, l_);
      dma_conv<W48, C48_WAVES>(wb, a.b.w + (int64_t)s * a.b.w_ss, wave, l_);
      dma_barrier();
      cur_s = s;
    }
    uint32_t scr[2][3];          // the shortcut's quint8 output of this wave's two rows (its channel half): the residual of stem.3's epilogue
    v4i wbr[T48::KS];
    lds_barrier();               // X complete (written during the previous item's stem.3), T free
    {
      // ---- phase 1: per output row, shortcut (1x1 / s2, K = 24 -> one k-step) and stem.0 (3x3 / s2, K = 216 -> 7 k-steps, TAIL form)
      int l_ = lane;
      asm volatile("" : "+v"(l_));
      const int r = l_ & 31, h = l_ >> 5, img = r >> 4, col = r & 15;
      v4i war[D24_KSA];
#pragma unroll
      for (int ks = 0; ks < D24_KSA; ++ks) war[ks] = *reinterpret_cast<const v4i*>(wa + half * D24_WA_HALF + l_ * 16 + ks * 1024);
      const v4i wsr = *reinterpret_cast<const v4i*>(ws + half * D24_WS_HALF + l_ * 16);
      int m = QBNN_MAGIC_BITS48;
      asm volatile("" : "+v"(m));            // the start block is built here, per phase (kept in scratch across the item loop otherwise: qbnn_w16.hip)
      const v16i mg = v16i{m, m, m, m, m, m, m, m, m, m, m, m, m, m, m, m};
      // output pixel (oh, col) reads input rows 2 oh - 1 .. 2 oh + 1 = tile rows 2 oh .. 2 oh + 2, columns 2 col - 1 .. = tile columns 2 col ..
      const uint8_t* px0 = xt + img * X24::STRIDE + (2 * col) * X24::PIXB;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int oh = 2 * rp + i;
        const uint8_t* base = px0 + (2 * oh) * X24::PITCH + 16 * h;
        const uint8_t* tbase = px0 + (2 * oh + (h ? 2 : 0)) * X24::PITCH + 64;      // tails: k-half 0 = rows 0, 1 of the window; k-half 1 = row 2 (, 3: unused)
        {                      // shortcut: the centre pixel = tile (2 oh + 1, 2 col + 1): its 24 channels + 8 bytes of the next pixel (weights 0)
          const uint8_t* cp = base + X24::PITCH + X24::PIXB;
          const v2i c0 = *reinterpret_cast<const v2i*>(cp), c1 = *reinterpret_cast<const v2i*>(cp + 8);
          const v16i accs = __builtin_amdgcn_mfma_i32_32x32x32_i8(wsr, v4i{c0.x, c0.y, c1.x, c1.y}, mg, 0, 0, 0);
          // its epilogue (no ReLU): EpiDense<HAS_RES = false>'s arithmetic; the dwords stay in registers.  (One conv after the other, each with its
          // bias table read just before: both accumulators + both tables beside stem.0's weights do not fit 128 VGPRs.)
          float4 bs[3];
#pragma unroll
          for (int g4 = 0; g4 < 3; ++g4) bs[g4] = *reinterpret_cast<const float4*>(bias_lds + 24 * half + 8 * g4 + 4 * h);
          const float cm = magic_c(accs, a.s);
          const float zy = (float)a.s.z_y;
#pragma unroll
          for (int g4 = 0; g4 < 3; ++g4) {
            float v[4];
            const float bb[4] = {bs[g4].x, bs[g4].y, bs[g4].z, bs[g4].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = med3f(__builtin_fmaf(bb[q], a.s.rcp, __int_as_float(accs[4 * g4 + q]) - cm) * a.s.mult, a.s.vlo, a.s.vhi);
            scr[i][g4] = pack_low_bytes((v[0] + QBNN_MAGIC) + zy, (v[1] + QBNN_MAGIC) + zy, (v[2] + QBNN_MAGIC) + zy, (v[3] + QBNN_MAGIC) + zy);
            // (pinned: the dwords are first USED behind the next barrier, and the optimiser otherwise sinks this whole epilogue there -- with
            //  the shortcut's 16 accumulators and its bias table parked in scratch across the barrier)
            asm volatile("" : "+v"(scr[i][g4]));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        v16i acc = mg;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
#if QBNN_D24_B128
            // the window starts at an even tile column: 48 col + 16 h + 32 t bytes into a 16-byte-aligned row -- ONE ds_read_b128 (4 - 8 LDS cycles per
            // wave) where two 8-byte halves compile to ds_read2_b64 (16; tools/lds_pattern_bench.hip, profiles/r05_lds_patterns.txt)
            const v4i xf = *reinterpret_cast<const v4i*>(base + kh * X24::PITCH + t * 32);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(war[kh * 2 + t], xf, acc, 0, 0, 0);
#else
            const v2i lo = *reinterpret_cast<const v2i*>(base + kh * X24::PITCH + t * 32), hi = *reinterpret_cast<const v2i*>(base + kh * X24::PITCH + t * 32 + 8);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(war[kh * 2 + t], v4i{lo.x, lo.y, hi.x, hi.y}, acc, 0, 0, 0);
#endif
          }
        {
          const v2i lo = *reinterpret_cast<const v2i*>(tbase), hi = *reinterpret_cast<const v2i*>(tbase + X24::PITCH);
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(war[6], v4i{lo.x, lo.y, hi.x, hi.y}, acc, 0, 0, 0);
        }
        {                      // stem.0 epilogue (ReLU-fused): centred bytes into T
          float4 ba[3];
#pragma unroll
          for (int g4 = 0; g4 < 3; ++g4) ba[g4] = *reinterpret_cast<const float4*>(bias_lds + T48::CH + 24 * half + 8 * g4 + 4 * h);
          const float cm = magic_c(acc, a.a);
          const int po = px48(img, oh, col);
#pragma unroll
          for (int g4 = 0; g4 < 3; ++g4) {
            float v[4];
            const float bb[4] = {ba[g4].x, ba[g4].y, ba[g4].z, ba[g4].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = __builtin_fmaf(bb[q], a.a.rcp, __int_as_float(acc[4 * g4 + q]) - cm) * a.a.mult;
            *reinterpret_cast<uint32_t*>(tt + po + 24 * half + 8 * g4 + 4 * h) = pack_rne_u8(v[0], v[1], v[2], v[3], a.a.vhi);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_sched_barrier(0);     // (keeps the stem.3 weight reads below behind the row loop: hoisted to the phase's top they are spilled)
#pragma unroll
      for (int ks = 0; ks < T48::KS; ++ks) wbr[ks] = *reinterpret_cast<const v4i*>(wb + half * C48_WHALF + l_ * 16 + ks * 1024);
    }
    lds_barrier();               // T complete; X has been read for the last time in this item
#ifdef QBNN_C48_STAMP
    const bool st_on = false;
    int st_k = 0;
#endif
    conv48_pair(tt, wbr, nullptr, bias_lds + 2 * T48::CH, a.b,
                EpiOut48<true>{nullptr, a.y + (int64_t)s * a.y_ss + (int64_t)img0 * C48_IMG, a.B - img0, a.b, a.add}, half, rp, lane,
                [&] {            // X is not read in this phase: item it + 1's tile is written behind stem.3's MFMAs, item it + 2 goes into registers
                  if (it + 1 < count) write_x();      // (requested only here, under the epilogues, and written after them: 0.431 against 0.411 ms)
                  fetch(it + 2 < count ? item + 2 : item);
                }, scr C48_STAMP_PASS);
  }
}

template <int NM, class A, class K>
int launch_items(K kernel, const char* what, int lds, const A* arr, int n, hipStream_t st, std::atomic<uint64_t>& attr) {
  static_assert(sizeof(ArgsArr<A, NM>) <= 3840, "kernel arguments are limited to 4 KiB (incl. the hidden ones)");
  if (int rc_attr = ensure_dyn_lds((const void*)kernel, attr, lds)) return rc_attr;
  ArgsArr<A, NM> all;
  memset(&all, 0, sizeof(all));
  int items = 0;
  for (int i = 0; i < n; ++i) {
    all.m[i] = arr[i];
    const int it = arr[i].n_samples * ((arr[i].B + C48_G - 1) / C48_G);
    items = it > items ? it : items;
  }
  const int per = 256 / n > 0 ? 256 / n : 1;
  const int gx = items < per ? (items > 0 ? items : 1) : per;
  hipLaunchKernelGGL(kernel, dim3(gx, n), dim3(C48_THREADS), lds, st, all);
  return check_launch(what);
}
template <class A, class K>
int launch_items_dev(K kernel, const char* what, int lds, const A* dev, int n, int items, hipStream_t st, std::atomic<uint64_t>& attr) {
  if (int rc_attr = ensure_dyn_lds((const void*)kernel, attr, lds)) return rc_attr;
  const int per = 256 / n > 0 ? 256 / n : 1;
  const int gx = items < per ? (items > 0 ? items : 1) : per;
  hipLaunchKernelGGL(kernel, dim3(gx, n), dim3(C48_THREADS), lds, st, ArgsArr<A, 0>{dev});
  return check_launch(what);
}

}  // namespace

// entry points for qbnn_blocks.hip (declared in qbnn_host.h)
int qbnn_launch_chain48_w16(const ChainArgs<1>* arr, int n, hipStream_t st) {
  static_assert(c48_lds() <= 160 * 1024, "LDS budget");
  static std::atomic<uint64_t> attr1{0}, attrN{0};
  if (n < 1 || n > QBNN_FUSED_CALLS) return fail(QBNN_E_INVALID, "qbnn_launch_chain48_w16: 1 to 8 argument blocks per launch%s");
  if (n == 1) return launch_items<1>(chain48_w16_kernel<1>, "qbnn_block_chain_i8_mc (48 channels, N24 layout)", c48_lds(), arr, 1, st, attr1);
  return launch_items<QBNN_FUSED_CALLS>(chain48_w16_kernel<QBNN_FUSED_CALLS>, "qbnn_block_chain_i8_multi (48 channels, N24 layout)", c48_lds(), arr, n, st, attrN);
}
int qbnn_launch_chain48_w16_dev(const ChainArgs<1>* dev, int n, int items, hipStream_t st) {
  static std::atomic<uint64_t> attr{0};
  return launch_items_dev(chain48_w16_kernel<0>, "qbnn_block_chain_i8_multi_launch (48 channels, N24 layout)", c48_lds(), dev, n, items, st, attr);
}
int qbnn_launch_down24_w16(const DownArgs* arr, int n, hipStream_t st) {
  static_assert(d24_lds() <= 160 * 1024, "LDS budget");
  static std::atomic<uint64_t> attr1{0}, attrN{0};
  if (n < 1 || n > QBNN_FUSED_CALLS) return fail(QBNN_E_INVALID, "qbnn_launch_down24_w16: 1 to 8 argument blocks per launch%s");
  if (n == 1) return launch_items<1>(down24_w16_kernel<1>, "qbnn_block_down_i8_mc (24 -> 48, N24 layout)", d24_lds(), arr, 1, st, attr1);
  return launch_items<QBNN_FUSED_CALLS>(down24_w16_kernel<QBNN_FUSED_CALLS>, "qbnn_block_down_i8_multi (24 -> 48, N24 layout)", d24_lds(), arr, n, st, attrN);
}
int qbnn_launch_down24_w16_dev(const DownArgs* dev, int n, int items, hipStream_t st) {
  static std::atomic<uint64_t> attr{0};
  return launch_items_dev(down24_w16_kernel<0>, "qbnn_block_down_i8_multi_launch (24 -> 48, N24 layout)", d24_lds(), dev, n, items, st, attr);
}
