// libqbnn_hip.so -- register-stationary conv kernels for the wide layers (96 / 192 channels, 8x8 / 4x4 maps).
//
// The fused wide-block kernels of qbnn_blocks.hip stream a block's sampled weights (162 / 663 KiB per MC sample) through an LDS
// ring once per work item of 8 / 16 images: ~4 GB of L2 -> LDS traffic per 100-sample step at B = 256, slab barriers in every
// K loop, and 200+ VGPRs per wave (two waves per SIMD, matrix and vector phases serialised).  Here ONE conv runs per launch
// and its weights never move after they are loaded:
//   * a workgroup is 12 waves (three per SIMD, <= 168 VGPRs); wave = (output-channel tile n, pixel lane ml).  The wave keeps the
//     27 weight fragments of its 32 output channels (a 96 -> 96 3x3 conv: K = 864 = 27 k-steps) in registers for as long as the
//     workgroup stays on one MC sample -- workgroups walk contiguous (sample, image group) ranges, so that is 1-2 loads per launch;
//   * the images of a work item (8 at 8x8) are a dense LDS tile, pixel pitch C + 16 bytes (conflict-free ds_read_b128 pixel
//     fragments); zero padding is a per-lane address choice (a tap outside the map reads a line of zeros).  The next item's tile
//     arrives by LDS-DMA (global_load_lds) while the current one is multiplied; one pass centres it (x - z_x) and leaves each
//     pixel's channel sum in a table (the window sum R of the zero-point correction is then 9 table reads per output pixel);
//   * per 32-pixel tile a wave issues its 27 MFMAs and requantises the 32 x 32 result straight to global memory (one dword per
//     lane and channel group: the four consecutive channels a lane owns) -- no staging, no barrier; waves only meet at the two
//     barriers of a tile swap, so their matrix and vector phases drift apart and overlap across the SIMD's three waves.
// The block's intermediate tensor (stem.0's output) goes through L2 / HBM between the two launches (6 KB per image at 8x8x96).
// Arithmetic and packed weight layout are those of conv_i8_kernel (EpiDense): bit-identical results; entered through the same
// C ABI call (qbnn_conv2d_i8_mc) for the geometries it covers.
#include "qbnn_host.h"

constexpr int RS_WAVES = 12, RS_THREADS = 64 * RS_WAVES;
#ifndef QBNN_RS_DEPTH
#define QBNN_RS_DEPTH 6
#endif
constexpr int RS_DEPTH = QBNN_RS_DEPTH;      // pixel fragments in flight per wave

template <int CH_, int HO_, int G_>
struct RSCfg {
  static constexpr int CH = CH_, HO = HO_, G = G_;
  static constexpr int NT = CH / 32;                   // output-channel tiles = wave roles
  static constexpr int SPT = CH / 32;                  // k-steps per tap
  static constexpr int KS = 9 * SPT;                   // k-steps of the conv (all held by one wave)
  // Pixel pitch of the LDS tile: C channel bytes + 48.  The pad keeps the 32 pixels of a ds_read_b128 fragment on distinct banks
  // (pitch 36 / 60 dwords) and carries the pixel's channel sum S(p) twice, at byte C (read by the lanes of k-half 0) and C + 16
  // (k-half 1, whose fragment address is 16 bytes further): the window sum R is then 9 reads off the fragment addresses.
  static constexpr int PIXB = CH + 48;
  static constexpr int CPP = PIXB / 16;                // 16-byte chunks per padded pixel
  static constexpr int IMG_PX = HO * HO;
  static constexpr int NPX = G * IMG_PX;               // pixels per work item
  static constexpr int TILE = NPX * PIXB;
  // Zero region in front of the tiles: a tap outside the map selects its start instead of the lane's window origin, and the tap's
  // (compile-time) offset is added to either -- so it spans the largest tap offset plus one pixel, and the window origin of
  // the first pixels (which lies before their tile) stays inside the workgroup's LDS.
  static constexpr int ZR = ((2 * HO + 3) * PIXB + 15) / 16 * 16;
  static constexpr int MT = NPX / 32;                  // 32-pixel tiles per item
  static constexpr int ML = RS_WAVES / NT;             // pixel lanes
  static constexpr int TPW = MT / ML;                  // tiles per wave and item
  static constexpr int NPIECE = NPX * CPP / 64;        // 1 KiB LDS-DMA pieces per tile
  static constexpr int LDS = ZR + 2 * TILE + CH * 4;
  static_assert(RS_WAVES % NT == 0 && MT % ML == 0 && (NPX * CPP) % 64 == 0 && NPX <= RS_THREADS && (CH % 32) == 0, "RS geometry");
  static_assert((32 % HO) == 0 && (IMG_PX % 32) == 0 || (32 % IMG_PX) == 0, "a 32-pixel tile is whole rows of one image, or whole images");
  static_assert(TILE % 16 == 0 && ((ML * TPW * 32) % IMG_PX) == 0, "tile alignment");
  static_assert(KS * 4 <= 112, "the wave's weight fragments must fit beside the accumulator at three waves per SIMD");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

// one LDS-DMA wave instruction: 16 bytes per lane from `src` (per-lane address) to dst + 16 * lane (dst wave-uniform).
// (A __device__ function: the builtin does not exist in the host pass, which silently drops a __global__ template that names it.)
static __device__ __forceinline__ void lds_dma16(const uint8_t* src, uint8_t* dst) {
  __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

static __device__ __forceinline__ int dot16(const v4i& c) {
  int d = __builtin_amdgcn_sdot4(c.x, 0x01010101, 0, false);
  d = __builtin_amdgcn_sdot4(c.y, 0x01010101, d, false);
  d = __builtin_amdgcn_sdot4(c.z, 0x01010101, d, false);
  return __builtin_amdgcn_sdot4(c.w, 0x01010101, d, false);
}

// MODE 0: ReLU-fused conv (lower clamp bound 0: v_cvt_pk_u8_f32 rounds and saturates), 1: plain conv, 2: conv + quantized::add + ReLU
template <class R, int MODE>
__global__ __launch_bounds__(RS_THREADS) void rs_conv3x3_kernel(const ConvArgs a, int n_samples) {
  constexpr bool HAS_RES = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* tiles = smem + R::ZR;                                           // zero region, then two dense tiles
  float* bias_lds = reinterpret_cast<float*>(tiles + 2 * R::TILE);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = wave % R::NT, ml = wave / R::NT;
  const int r = lane & 31, h = lane >> 5;

  const int groups = (a.B + R::G - 1) / R::G;
  int begin, count;
  item_range(n_samples * groups, blockIdx.x, gridDim.x, begin, count);
  if (count <= 0) return;

  for (int i = tid; i < R::ZR / 4; i += RS_THREADS) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
  load_bias<R::CH, RS_THREADS>(bias_lds, a.p.bias, tid);

  // Which of the 9 taps of this lane's pixel lie inside the map.  A wave's tile t covers pixels 32 (ml TPW + t) + r of the item:
  // their position inside the image repeats with period IMG_PX / 32 tiles (1 where a tile is one or more whole images).
  constexpr int NPAR = R::IMG_PX > 32 ? R::IMG_PX / 32 : 1;
  bool tap_ok[NPAR][9];
#pragma unroll
  for (int par = 0; par < NPAR; ++par) {
    const int rem = (par * 32 + r) % R::IMG_PX, oh = rem / R::HO, ow = rem % R::HO;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
      tap_ok[par][tap] = (unsigned)(oh + tap / 3 - 1) < (unsigned)R::HO && (unsigned)(ow + tap % 3 - 1) < (unsigned)R::HO;
  }

  // raw quint8 images of one item -> LDS tile (pitch PIXB), up to 1 KiB per wave instruction (pad chunks and images beyond the
  // batch are not fetched: the centring pass writes the former's channel sums and zeroes the latter)
  auto dma_tile = [&](int item, uint8_t* dst) {
    const int s = item / groups, img0 = (item - s * groups) * R::G;
    const int valid_px = (a.B - img0 < R::G ? a.B - img0 : R::G) * R::IMG_PX;
    const uint8_t* xs = a.x + (int64_t)s * a.x_ss + (int64_t)img0 * R::IMG_PX * R::CH;
    int l_ = lane;
    asm volatile("" : "+v"(l_));
    for (int piece = wave; piece < R::NPIECE; piece += RS_WAVES) {
      const int c = piece * 64 + l_;
      const int px = c / R::CPP, sub = c - px * R::CPP;
      if (sub < R::CH / 16 && px < valid_px) lds_dma16(xs + px * R::CH + sub * 16, dst + piece * 1024);
    }
  };

  v4i w[R::KS];
  int cur = 0, cur_s = -1;
  dma_tile(begin, tiles);
  for (int it = 0; it < count; ++it) {
    const int item = begin + it;
    const int s = item / groups, img0 = (item - s * groups) * R::G;
    const int valid_px = (a.B - img0 < R::G ? a.B - img0 : R::G) * R::IMG_PX;
    if (s != cur_s) {                       // this wave's 32 output channels of the sample's weights: registers, until the sample changes
      const int8_t* wq = a.p.w + (int64_t)s * a.p.w_ss + ((int64_t)n * R::KS * 64 + lane) * 16;
#pragma unroll
      for (int ks = 0; ks < R::KS; ++ks) w[ks] = *reinterpret_cast<const v4i*>(wq + (int64_t)ks * 1024);
      cur_s = s;
    }
    uint8_t* tile = tiles + cur * R::TILE;
    dma_barrier();                          // vmcnt(0) + barrier: the raw tile has landed (and everyone has left the other buffer)
    if (tid < R::NPX) {                     // centre the tile in place (x' = x - z_x; images beyond the batch: zeros), channel sums into the pad
      const uint32_t z4 = (uint32_t)a.p.z_x * 0x01010101u;
      uint8_t* q = tile + tid * R::PIXB;
      const bool ok = tid < valid_px;
      int sum = 0;
#pragma unroll
      for (int j = 0; j < R::CH / 16; ++j) {
        const v4i v = *reinterpret_cast<const v4i*>(q + 16 * j);
        const v4i c = ok ? v4i{(int)sub_bytes(v.x, z4), (int)sub_bytes(v.y, z4), (int)sub_bytes(v.z, z4), (int)sub_bytes(v.w, z4)} : v4i{0, 0, 0, 0};
        *reinterpret_cast<v4i*>(q + 16 * j) = c;
        sum += dot16(c);
      }
      *reinterpret_cast<int*>(q + R::CH) = sum;
      *reinterpret_cast<int*>(q + R::CH + 16) = sum;
    }
    lds_barrier();
    if (it + 1 < count) dma_tile(item + 1, tiles + (cur ^ 1) * R::TILE);        // in flight under this item's MFMAs
    uint8_t* ys = a.y + (int64_t)s * a.y_ss + (int64_t)img0 * R::IMG_PX * R::CH;
    const uint8_t* rs = HAS_RES ? a.res + (int64_t)s * a.res_ss + (int64_t)img0 * R::IMG_PX * R::CH : nullptr;
    const uint8_t* zsel = smem + 16 * h;
#pragma unroll
    for (int t = 0; t < R::TPW; ++t) {
      constexpr int dummy = 0; (void)dummy;
      const int par = (ml * R::TPW + t) % NPAR;               // (ml TPW is a multiple of NPAR: the parity is t's, a compile-time value)
      const int m = (ml * R::TPW + t) * 32 + r;              // this lane's pixel of the item
      const bool live = m < valid_px;
      // window origin of the pixel: tap (kh, kw) is the pixel (kh HO + kw) further
      const uint8_t* worg = tile + (m - R::HO - 1) * R::PIXB + 16 * h;
      uint32_t resq[4];
      const uint8_t* rp = rs + (int64_t)m * R::CH + 32 * n + 4 * h;
      if (HAS_RES) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) resq[g4] = live ? *reinterpret_cast<const uint32_t*>(rp + 8 * g4) : 0u;
      }
      v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      int Rsum = 0;
      // The wave's MFMAs wait on one LDS fragment each (the weights are registers), so the fragment reads run RS_DEPTH k-steps
      // ahead of the MFMA that consumes them: an LDS round trip is ~100+ cycles under load, an MFMA 32.
      const uint8_t* tb[9];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) tb[tap] = (tap_ok[t % NPAR][tap] ? worg : zsel) + ((tap / 3) * R::HO + tap % 3) * R::PIXB;
      v4i xr[RS_DEPTH];
#pragma unroll
      for (int i = 0; i < RS_DEPTH && i < R::KS; ++i) xr[i] = *reinterpret_cast<const v4i*>(tb[i / R::SPT] + 32 * (i % R::SPT));
#pragma unroll
      for (int i = 0; i < R::KS; ++i) {
        const v4i xc = xr[i % RS_DEPTH];
        if (i + RS_DEPTH < R::KS) xr[i % RS_DEPTH] = *reinterpret_cast<const v4i*>(tb[(i + RS_DEPTH) / R::SPT] + 32 * ((i + RS_DEPTH) % R::SPT));
        if (i % R::SPT == 0) Rsum += *reinterpret_cast<const int*>(tb[i / R::SPT] + R::CH);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[i], xc, acc, 0, 0, 0);
      }
      (void)par;
      // FBGEMM requantisation (+ quantized::add + ReLU), as EpiDense; one dword per channel group straight to global memory
      const int zwr = a.p.z_w * Rsum;
      uint8_t* yo = ys + (int64_t)m * R::CH + 32 * n + 4 * h;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 bb = *reinterpret_cast<const float4*>(bias_lds + 32 * n + 8 * g4 + 4 * h);
        float v[4];
        v[0] = __builtin_fmaf(bb.x, a.p.rcp, (float)(acc[4 * g4 + 0] - zwr)) * a.p.mult;
        v[1] = __builtin_fmaf(bb.y, a.p.rcp, (float)(acc[4 * g4 + 1] - zwr)) * a.p.mult;
        v[2] = __builtin_fmaf(bb.z, a.p.rcp, (float)(acc[4 * g4 + 2] - zwr)) * a.p.mult;
        v[3] = __builtin_fmaf(bb.w, a.p.rcp, (float)(acc[4 * g4 + 3] - zwr)) * a.p.mult;
        uint32_t o;
        if (MODE == 0) {                            // ReLU-fused conv: non-negative centred bytes, then + z_y bytewise
          o = pack_rne_u8(v[0], v[1], v[2], v[3], a.p.vhi) + (uint32_t)a.p.z_y * 0x01010101u;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = med3f(v[i], a.p.vlo, a.p.vhi);
          if (!HAS_RES) {
            const float zy = (float)a.p.z_y;       // round with the (even) magic constant first, then add z_y exactly
            o = pack_low_bytes((v[0] + QBNN_MAGIC) + zy, (v[1] + QBNN_MAGIC) + zy, (v[2] + QBNN_MAGIC) + zy, (v[3] + QBNN_MAGIC) + zy);
          } else {
            const uint32_t rq = resq[g4];
            const float rf[4] = {(float)(rq & 0xffu), (float)((rq >> 8) & 0xffu), (float)((rq >> 16) & 0xffu), (float)(rq >> 24)};
            float tt[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float da = __builtin_fmaf(a.p.s_y, __builtin_rintf(v[i]), a.p.dl_y);      // centred conv output integer, exactly
              const float db = __builtin_fmaf(a.a.s_r, rf[i], a.a.nzs_r);
              tt[i] = (da + db) * a.a.inv_s_o;
            }
            o = pack_rne_u8(tt[0], tt[1], tt[2], tt[3], a.a.vhi) + (uint32_t)a.a.z_o * 0x01010101u;      // bytes <= a_hi - z_o: no carry
          }
        }
        if (live) *reinterpret_cast<uint32_t*>(yo + 8 * g4) = o;
      }
    }
    cur ^= 1;
  }
}

template <class R>
static int launch_rs(const ConvArgs& a, int n_samples, bool has_res, hipStream_t st) {
  const int groups = (a.B + R::G - 1) / R::G;
  const int n_items = n_samples * groups;
  const int grid = n_items < 256 ? n_items : 256;
  if (has_res) {
    static std::atomic<uint64_t> attr_r{0};
    if (int rc = ensure_dyn_lds((const void*)rs_conv3x3_kernel<R, 2>, attr_r, R::LDS)) return rc;
    hipLaunchKernelGGL((rs_conv3x3_kernel<R, 2>), dim3(grid), dim3(RS_THREADS), R::LDS, st, a, n_samples);
  } else if (a.p.vlo == 0.0f) {
    static std::atomic<uint64_t> attr_n{0};
    if (int rc = ensure_dyn_lds((const void*)rs_conv3x3_kernel<R, 0>, attr_n, R::LDS)) return rc;
    hipLaunchKernelGGL((rs_conv3x3_kernel<R, 0>), dim3(grid), dim3(RS_THREADS), R::LDS, st, a, n_samples);
  } else {
    static std::atomic<uint64_t> attr_p{0};
    if (int rc = ensure_dyn_lds((const void*)rs_conv3x3_kernel<R, 1>, attr_p, R::LDS)) return rc;
    hipLaunchKernelGGL((rs_conv3x3_kernel<R, 1>), dim3(grid), dim3(RS_THREADS), R::LDS, st, a, n_samples);
  }
  return check_launch("qbnn_conv2d_i8_mc");
}

// entry point for qbnn_kernels.hip's dispatch (declared in qbnn_host.h): 3x3 / stride 1 / pad 1 convs at 96 channels on 8x8 maps
int qbnn_launch_rs_conv(const ConvArgs& a, int cin, int hin, int n_samples, bool has_res, hipStream_t st) {
  if (cin == 96 && hin == 8) return launch_rs<RSCfg<96, 8, 8>>(a, n_samples, has_res, st);
  return fail(QBNN_E_INVALID, "qbnn_launch_rs_conv: unsupported geometry%s C=%ld H=%ld", "", cin, hin);
}
