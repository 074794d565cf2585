// libqbnn_hip.so -- ring form of the WIDE down-sampling BasicBlocks (reference models_bbb.py:146-183 with stride 2:
// shortcut 1x1/s2 conv, stem.0 3x3/s2 ConvReLU, stem.3 3x3 conv, Add, ReLU) at 48 -> 96 channels (16 x 16 -> 8 x 8) and
// 96 -> 192 (8 x 8 -> 4 x 4).  Same arithmetic, epilogue functors and argument block as block_down_ws_kernel (qbnn_blocks.hip).
//
// Why a second form (round 4): block_down_ws_kernel streams every weight fragment of these blocks per WAVE from L2 through the
// vector L1 -- one pixel tile per wave, so 1 KiB of weights per MFMA, 1.0 / 2.0 MB per work item through a 64 B/clk L1 (counters:
// issue-stalled 36 / 47 % of wave-cycles, MFMA busy 23 / 24 %).  Here the block's weights reach the CU ONCE per work item: the 8 waves
// DMA them (global_load_lds) into an LDS slab ring and read their fragments with ds_read_b128, as the identity chains of these widths do.
// A two-slab ring (one slab in flight) was measured first and is DMA-latency-bound here: these items have 12 MFMAs per wave and slab
// against an L2 -> LDS latency of 2.5 - 3 k cycles under load (one 30 KiB slab in flight = 10 B/clk per CU; profiles/r04_stamp_down_ring.txt).
// So the ring is NBUF = 4 slabs deep with THREE in flight (72 KiB per CU, the guide's figure for the full L2 -> LDS rate):
//   * the DMA is issued from inline assembly -- the compiler otherwise puts `s_waitcnt vmcnt(0)` in front of every LDS access that
//     follows a global_load_lds it cannot prove disjoint (all of the epilogues' bias reads / tile writes), draining the ring each time;
//   * a wave waits for ITS share of slab q with `s_waitcnt vmcnt((NBUF - 2) * C)` (C = its DMA instructions per slab: counters retire in
//     order, and any other vector-memory operation in between only makes the wait stricter), then the workgroup barrier publishes
//     the slab and frees the buffer of slab q - 1, into which slab q + NBUF - 1 is requested.
//
// LDS: X tile (centred block input, zero halo on the top / left only -- a stride-2, pad-1 window never leaves the map at the
//        bottom / right; kernel rows stay contiguous, so k-steps may straddle taps: Cin = 48), rows padded by 16 B at 48 channels (bank conflicts
//        of the stride-2 fragment reads: 4-way -> 2-way; tools/lds_conflicts.py)
//      T tile (centred stem.0 output, dense [M][COUT + 16] + a zero line that taps outside the map read) and SC (shortcut output =
//        residual = block output staging, quint8 [M][COUT + 8]) both ALIAS X: stem.0 and the shortcut accumulate first (two accumulator
//        sets), a barrier marks the last read of X, then both epilogues run; the next item's X is written after SC's read-out
//      four weight slabs of 24 KiB, biases.
// Phases per work item (M = MFMA K loop over the ring, E = requantising epilogue):
//      M_a M_s | barrier | E_a(-> T) E_s(-> SC)   M_b E_b(SC += , ReLU)   | barrier, read-out, barrier, next X
// Window sums (sampled weights have a non-zero zero point: sum x'(W - z_w) = acc - z_w R) from per-pixel channel sums kept beside the tiles,
// as in the wide identity blocks: S_X (written with the X tile by the thread that moves the pixel: one v_dot4 per dword, one 16-bit store
// per pixel) serves stem.0 (its 3x3 / s2 window) and the shortcut (the centre pixel); S_T (one plain store per pixel and channel block from stem.0's epilogue) serves stem.3.  The first version of this
// kernel took them by 4 v_dot4 per k-step on the pixel fragments inside the K loop: 15 % of the M phases by ablation
// (profiles/r04_stamp_down_ring.txt).
#include "qbnn_host.h"

#ifndef QBNN_DOWN_PD
#define QBNN_DOWN_PD 3      // k-steps between a fragment's LDS request and its MFMAs
#endif

#ifdef QBNN_STAMP      // diagnostic build only (tools/stamp_ring.py): per-phase s_memtime sums; the shipped library has none of this
QBNN_EXPORT void qbnn_debug_stamp_buffer_ring(void* p) { hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_dev_ptr), &p, sizeof(p)); }
QBNN_EXPORT void qbnn_debug_read_inner_ring(unsigned long long* host4) {
  hipMemcpyFromSymbol(host4, HIP_SYMBOL(g_inner), 32);
  unsigned long long z[4] = {0, 0, 0, 0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_inner), z, 32);
}
#endif

namespace {

//                      CIN COUT HIN  G  SLK (k-steps per weight slab)  XROWPAD (bytes behind every X row)
template <int CIN_, int COUT_, int HIN_, int G_, int SLK_, int XROWPAD_>
struct DRCfg {
  static constexpr int CIN = CIN_, COUT = COUT_, HIN = HIN_, HO = HIN_ / 2, G = G_, SLK = SLK_;
  static constexpr int XTW = HIN + 1;                       // top / left halo
  static constexpr int XROW = XTW * CIN + XROWPAD_;         // row pitch: bank conflicts of the stride-2 fragment reads (tools/lds_conflicts.py)
  static constexpr int XIMG = XTW * XROW;
  static constexpr int X_BYTES = G * XIMG + ((3 * CIN) % 32 ? 32 : 0);      // the last k-step of a ragged kernel row over-reads < 32 bytes
  static constexpr int RB_A = 3 * CIN, SPR_A = (RB_A + 31) / 32, KS_A = 3 * SPR_A;      // stem.0: 3x3 / s2 on X
  static constexpr int KS_S = (CIN + 31) / 32;                                          // shortcut: 1x1 / s2 on X
  static constexpr int SPT_B = COUT / 32, KS_B = 9 * SPT_B;                             // stem.3: 3x3 / s1 on T
  static constexpr int NT = COUT / 32, NB = 3, NBLKS = NT / NB;
  static constexpr int M = G * HO * HO, MT = M / 32;
  static constexpr int PIXB_T = COUT + 16;                  // 112 / 208 bytes: conflict-free ds_read_b128 of 32 consecutive pixels
  static constexpr int T_BYTES = M * PIXB_T + PIXB_T;       // + the zero line
  static constexpr int SCP = COUT + 8, SC_BYTES = M * SCP;
  static constexpr int SC_OFF = T_BYTES;                    // T at byte 0 of the X region, SC behind it
  // the weight ring: NBUF slabs of SLK k-steps x NT channel tiles (1 KiB fragments); a slab never spans two convs, a conv's last
  // slab may be short (its spare fragments are loaded twice: every wave issues the same number of DMA instructions per slab)
  static constexpr int NBUF = 4;
  static constexpr int NF = NT * SLK, DMA_PER_WAVE = NF / 8;
  static constexpr int SLABB = NF * 1024;
  static constexpr int NS_A = (KS_A + SLK - 1) / SLK, NS_S = (KS_S + SLK - 1) / SLK, NS_B = (KS_B + SLK - 1) / SLK, NSI = NS_A + NS_S + NS_B;
  static constexpr int NPX = G * HIN * HIN;                 // S_X: int16 [NPX]
  static constexpr int ST_OFF = SC_OFF + SC_BYTES;          // S_T [NBLKS][M] ints: in the X region behind SC (X is dead while T lives)
  static constexpr int LDS = X_BYTES + NBUF * SLABB + 3 * COUT * 4 + NPX * 2;
  static constexpr int NHALO = XTW * CIN / 16 + HIN * (CIN / 16);                              // ... of one image's halo
  static_assert(MT * NBLKS == 8, "one (pixel tile, channel half) pass per wave");
  static_assert(NF % 8 == 0, "every wave issues the same number of DMA instructions per slab");
  static_assert(XROW % 16 == 0 && XIMG % 16 == 0 && CIN % 16 == 0 && COUT % 96 == 0 && M % 32 == 0 && T_BYTES % 16 == 0, "alignment");
  static_assert(ST_OFF % 16 == 0 && ST_OFF + (NBLKS * M + HO + 1) * 4 <= G * XIMG, "T, SC and the S_T table alias the X tile");
  static_assert(HIN % 2 == 0 && CIN * 128 < 16384, "S_X fields: column parity = pixel parity; |channel sum| < 2^14");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};
using DR48 = DRCfg<48, 96, 16, 4, 8, 16>;     // 24 KiB slabs (NT = 3); X reads 2-way conflicted (4-way without the row pad)
using DR96 = DRCfg<96, 192, 8, 8, 4, 0>;      // 24 KiB slabs (NT = 6); X reads 2-way conflicted -- a 16-byte row pad makes them conflict-free,
                                              // but then the fourth slab buffer no longer fits (164,000 B), and three slabs in flight matter more
// epilogue blocking (MB = 1 pixel tile x NB = 3 channel tiles per wave): conv_epi_phase reads MB / NB / NBLKS / COUT of these
using E48 = ConvCfg<48, 96, 3, 2, 16, 1, 4, 1, 3, false>;
using E96 = ConvCfg<96, 192, 3, 2, 8, 1, 8, 1, 3, false>;

// One conv's M phase: its KS k-steps as ONE software-pipelined stream (fragments are requested PD k-steps ahead of their MFMAs -- with
// three MFMAs per k-step and wave, one step ahead exposed the LDS latency at every step: 625 cycles per k-step against 96 of MFMAs,
// profiles/r04_stamp_down_ring.txt), fully unrolled, so every tile offset is an immediate.  Where the stream of requests enters a new
// weight slab the ring advances IN the stream: wait for the slab, barrier, request slab + NBUF - 1 -- the MFMAs of the previous slab's
// last steps are still to issue, so the pipeline does not drain at slab boundaries.  This wave accumulates its pixel tile against the
// channel tiles nblk * 3 .. + 2.  step(ks) -> address of k-step ks's pixel fragment; issue(q, buffer) requests flat slab q.
// init() -> the accumulators' start value -z_w R (R = this pixel's window sum, from the channel-sum tables: the epilogue then has one vector
// instruction fewer per output); called once the conv's first ring barrier has passed -- the tables' writers lie before it.
template <class D, int KS, class StepFn, class IssueFn, class InitFn>
__device__ __forceinline__ void ring_mfma(StepFn step, WeightRing& rg, ConvAccMN<1, 3>& A, int nblk, int wave, int lane, IssueFn issue, InitFn init) {
  constexpr int SLK = D::SLK, PD = QBNN_DOWN_PD;
  struct Frag { v4i w[3]; v4i x; };
  Frag f[PD + 1];
  const uint8_t* wl = nullptr;
  auto advance = [&]() {
    wait_vmcnt<(D::NBUF - 2) * D::DMA_PER_WAVE>();      // this wave's share of the slab has landed ...
    lds_barrier();                                      // ... everyone's has; everyone has read the previous slab's last fragments
    issue(rg.pnext, rg.pbuf);
    ++rg.pnext;
    rg.pbuf = rg.pbuf + 1 == D::NBUF ? 0 : rg.pbuf + 1;
    wl = rg.base + rg.cbuf * D::SLABB + ((nblk * 3) * SLK * 64 + lane) * 16;
    rg.cbuf = rg.cbuf + 1 == D::NBUF ? 0 : rg.cbuf + 1;
  };
  auto load = [&](Frag& fr, int ks) {
    const int j = ks % SLK;
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) fr.w[nb] = *reinterpret_cast<const v4i*>(wl + (nb * SLK + j) * 1024);
    fr.x = *reinterpret_cast<const v4i*>(step(ks));
  };
  auto mfma = [&](const Frag& fr) {
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) A.acc[0][nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fr.w[nb], fr.x, A.acc[0][nb], 0, 0, 0);
  };
#pragma unroll
  for (int p = 0; p < PD && p < KS; ++p) {
    if (p % SLK == 0) advance();
    if (p == 0) {
      const int a0 = init();
#pragma unroll
      for (int nb = 0; nb < 3; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) A.acc[0][nb][i] = a0;
    }
    load(f[p % (PD + 1)], p);
  }
#pragma unroll
  for (int j = 0; j < KS; ++j) {
    const int p = j + PD;
    if (p < KS) {
      if (p % SLK == 0) advance();
      load(f[p % (PD + 1)], p);
    }
    mfma(f[j % (PD + 1)]);
    __builtin_amdgcn_sched_barrier(0);                  // keep the steps apart: the request distance is the point
  }
}

// epilogue over accumulators that already hold sum x'(W - z_w)
template <class EC, class Epi>
__device__ __forceinline__ void epi_presub(const float* bias_lds, const QConv& p, Epi& epi, ConvAcc<EC>& A, int wave, int lane) {
  auto ld = [&](int, int, int, int po, int c0) { return epi.load(po, c0); };
  auto none = [](int) {};
  conv_epi_phase_with<EC, Epi, decltype(ld), decltype(none), true>(bias_lds, p, epi, A, wave, lane, ld, none);
}

// DROP (conv_resnet_mc, round 5): a quantised channel dropout behind every conv -- dr.d = stem.3 (behind stem.0), stem.6 (behind the second
// conv: the Add's first operand), shortcut.2 (behind the shortcut conv: its second) -- applied in the epilogues (EpiDenseTileDrop / EpiDenseDrop of
// block_down_ws_kernel's DROP form) from ONE-BIT mask tables: a Bernoulli mask has two quantised values, 3 dropouts x G images x COUT channels are
// 144 / 576 bytes per work item.  At 48 -> 96 they have LDS of their own and are drawn at the top of the item; the 96 -> 192 configuration
// uses all 160 KiB, so there they sit in the tail of the X region, which is dead once the two first convs' MFMAs are done (T, SC and S_T
// take its head): drawn behind that barrier, published by one more.
template <class D> struct DRMask {
  using MT = MaskTab<D::COUT, true>;
  static constexpr int MTB = MT::bytes(D::G);
  static constexpr bool IN_X = D::LDS + 3 * MTB > 160 * 1024;
  static constexpr int X_TAIL = (D::ST_OFF + (D::NBLKS * D::M + D::HO + 1) * 4 + 15) / 16 * 16;      // first free byte of the X region while T / SC / S_T live
  static_assert(!IN_X || X_TAIL + 3 * MTB <= D::G * D::XIMG, "the mask tables fit behind S_T");
  static constexpr int LDS = D::LDS + (IN_X ? 0 : 3 * MTB);
};

template <class D, class EC, int NM, bool DROP = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void block_down_ring_kernel(const ArgsArr<DownArgs, NM> all, const DropSet<DROP ? 3 : 0> dr) {
  const DownArgs a = args_of(all, blockIdx.y);
  static_assert(!DROP || NM == 1, "dropout variants are single-call");
  using DM = DRMask<D>;
  constexpr int NTHR = 512;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* xt = smem;
  uint8_t* tt = smem;                                        // T and SC alias X (see the header comment)
  uint8_t* zline = tt + D::M * D::PIXB_T;
  uint8_t* sc = smem + D::SC_OFF;
  uint8_t* rbase = smem + D::X_BYTES;
  float* bias_lds = reinterpret_cast<float*>(rbase + D::NBUF * D::SLABB);               // [3][COUT]: s, a, b
  int16_t* sx16 = reinterpret_cast<int16_t*>(bias_lds + 3 * D::COUT);                   // S_X: channel sums of the X tile's interior pixels
  int* stt = reinterpret_cast<int*>(smem + D::ST_OFF);                                  // S_T: ... of the T tile, per channel block
  uint8_t* mtab = DM::IN_X ? smem + DM::X_TAIL : reinterpret_cast<uint8_t*>(sx16 + D::NPX);      // DROP: three one-bit mask tables
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mblk = wave / D::NBLKS, nblk = wave - mblk * D::NBLKS;

  constexpr int CPP = D::CIN / 16, PPT = D::NPX / NTHR, PER_T = PPT * CPP;              // a thread moves whole pixels: PPT pixels of CPP 16-byte chunks
  static_assert(D::NPX % NTHR == 0, "whole pixels per thread");
  constexpr int IMG_IN = D::HIN * D::HIN * D::CIN, IMG_OUT = D::HO * D::HO * D::COUT, U8 = D::COUT / 8;
  constexpr int NOUT = (D::M * U8 + NTHR - 1) / NTHR;
  const int groups = (a.B + D::G - 1) / D::G;
  const ItemWalk walk(a.n_samples * groups, blockIdx.x, gridDim.x);      // interleaved per XCD: a sample's weights stay in ONE L2
  const int count = walk.count;

  load_bias<D::COUT, NTHR>(bias_lds, a.s.bias, tid);
  load_bias<D::COUT, NTHR>(bias_lds + D::COUT, a.a.bias, tid);
  load_bias<D::COUT, NTHR>(bias_lds + 2 * D::COUT, a.b.bias, tid);
  if (count <= 0) return;

  // A thread moves WHOLE pixels (pixel t + j * 512 of the item, CPP consecutive 16-byte chunks): the pixel's channel sum is then thread-local --
  // one 16-bit store into S_X, no atomics -- and the lanes' tile writes sit 48 / 96 bytes apart (conflict-free / 2-way).
  v4i pre[PER_T];
  auto fetch = [&](int item) {
    const int s = item / groups, img0 = (item - s * groups) * D::G;
    const uint8_t* xs = a.x + (int64_t)s * a.x_ss + (int64_t)img0 * IMG_IN;            // an item's images are contiguous in HBM
    const int valid = (a.B - img0 < D::G ? a.B - img0 : D::G) * D::HIN * D::HIN;
    int t = tid;
    asm volatile("" : "+v"(t));         // per-thread addresses are recomputed here, not hoisted out of the item loop (spills)
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int px = t + j * NTHR;
      const uint8_t* p = xs + (px < valid ? (int64_t)px * D::CIN : 0);
#pragma unroll
      for (int c = 0; c < CPP; ++c) pre[j * CPP + c] = *reinterpret_cast<const v4i*>(p + 16 * c);
    }
  };
  // registers -> centred X interior + S_X, and the halo zeros (T / SC, which share these bytes, have overwritten them)
  auto write_tile = [&](int item) {
    const int s = item / groups, img0 = (item - s * groups) * D::G;
    const int valid = (a.B - img0 < D::G ? a.B - img0 : D::G) * D::HIN * D::HIN;
    const uint32_t z4 = (uint32_t)a.z_in * 0x01010101u;
    int t = tid;
    asm volatile("" : "+v"(t));
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int px = t + j * NTHR;
      const int g = px / (D::HIN * D::HIN), rem = px - g * (D::HIN * D::HIN), row = rem / D::HIN, col = rem - row * D::HIN;
      uint8_t* dst = xt + g * D::XIMG + (row + 1) * D::XROW + (col + 1) * D::CIN;
      int sum = 0;
#pragma unroll
      for (int c = 0; c < CPP; ++c) {
        const v4i v = pre[j * CPP + c];
        const v4i q = px < valid ? v4i{(int)sub_bytes(v.x, z4), (int)sub_bytes(v.y, z4), (int)sub_bytes(v.z, z4), (int)sub_bytes(v.w, z4)} : v4i{0, 0, 0, 0};
        *reinterpret_cast<v4i*>(dst + 16 * c) = q;
        sum = __builtin_amdgcn_sdot4(q.x, 0x01010101, sum, false);
        sum = __builtin_amdgcn_sdot4(q.y, 0x01010101, sum, false);
        sum = __builtin_amdgcn_sdot4(q.z, 0x01010101, sum, false);
        sum = __builtin_amdgcn_sdot4(q.w, 0x01010101, sum, false);
      }
      sx16[px] = (int16_t)sum;
    }
    constexpr int TOP = D::XTW * D::CIN / 16, CW = D::CIN / 16;
    for (int i = t; i < D::G * D::NHALO; i += NTHR) {
      const int g = i / D::NHALO, q = i - g * D::NHALO;
      const int off = q < TOP ? q * 16 : (1 + (q - TOP) / CW) * D::XROW + ((q - TOP) % CW) * 16;
      *reinterpret_cast<v4i*>(xt + g * D::XIMG + off) = v4i{0, 0, 0, 0};
    }
  };
  // request flat slab q (item q / NSI of this workgroup's walk; beyond its last item: that item's slabs again -- harmless, and every
  // wave keeps issuing DMA_PER_WAVE instructions per slab, which is what the vmcnt accounting counts on) into ring buffer `buf`
  const uint32_t lane16 = (uint32_t)lane * 16u;      // (dma16_s: the per-lane part of every fragment address)
  auto issue = [&](int q, int buf) {
    int itx = q / D::NSI;
    const int loc = q - itx * D::NSI;
    itx = itx < count ? itx : count - 1;
    const int s = walk.item(itx) / groups;
    const int8_t* wq; int KS, slab;
    if (loc < D::NS_A) { wq = a.a.w + (int64_t)s * a.a.w_ss; KS = D::KS_A; slab = loc; }
    else if (loc < D::NS_A + D::NS_S) { wq = a.s.w + (int64_t)s * a.s.w_ss; KS = D::KS_S; slab = loc - D::NS_A; }
    else { wq = a.b.w + (int64_t)s * a.b.w_ss; KS = D::KS_B; slab = loc - D::NS_A - D::NS_S; }
    const uint32_t dst = lds_addr_of(rbase + buf * D::SLABB);
#pragma unroll
    for (int k = 0; k < D::DMA_PER_WAVE; ++k) {
      const int f = wave + 8 * k;
      const int nt = f / D::SLK, u = f - nt * D::SLK;
      int ks = slab * D::SLK + u;
      ks = ks < KS ? ks : KS - 1;
      dma16_s(wq + (int64_t)(nt * KS + ks) * 1024, lane16, dst + f * 1024);
    }
  };

  fetch(walk.item(0));
  write_tile(walk.item(0));
  WeightRing rg{rbase, 0, D::NBUF - 1, D::NBUF - 1};
#pragma unroll
  for (int q = 0; q < D::NBUF - 1; ++q) issue(q, q);
  ConvAccMN<1, 3> A, S;
  QBNN_STAMP_DECL
  for (int it = 0; it < count; ++it) {
    QBNN_STAMP_START();
    const int item = walk.item(it);
    const int s = item / groups, img0 = (item - s * groups) * D::G;
    const bool more = it + 1 < count;
    const int next = more ? walk.item(it + 1) : item;
    // this lane's output pixel (recomputed per item from an opaque lane index: held across the loop the addresses below spill)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int r = ln & 31, h = ln >> 5;
    const int m = mblk * 32 + r;
    const int g = m / (D::HO * D::HO), rem = m - g * (D::HO * D::HO), oh = rem / D::HO, ow = rem - oh * D::HO;
    const uint8_t* xlane = xt + g * D::XIMG + (2 * oh) * D::XROW + (2 * ow) * D::CIN + 16 * h;      // tap (0, 0) of this pixel's 3x3 / s2 window
    if constexpr (DROP && !DM::IN_X) {      // this item's masks (their readers lie behind the barrier that ends the first two convs' MFMAs)
#pragma unroll
      for (int d = 0; d < 3; ++d) fill_mask_tab<D::G, D::COUT, true, NTHR>(mtab + d * DM::MTB, dr.d[d], s, img0, a.B, tid);
    }
    // window sums of stem.0 (3x3 / s2: rows 2 oh - 1 .. 2 oh + 1, columns likewise; only the top / left can leave the map) and of the
    // shortcut (the centre pixel) from S_X.  The centre pixel (2 oh, 2 ow) has an even index: with sb = the dword that holds it, row kh's
    // three pixels are the high half of sb[.. - 1] and both halves of sb[..] -- two reads at immediate offsets per kernel row; a read in
    // front of the table is masked out.
    const int* sb = reinterpret_cast<const int*>(sx16) + (((g * D::HIN + 2 * oh) * D::HIN + 2 * ow) >> 1);
    // ---- stem.0: M over X (3x3 / s2)
    ring_mfma<D, D::KS_A>(
        [&](int ks) {
          const int kh = ks / D::SPR_A, t = ks - kh * D::SPR_A;
          return xlane + kh * D::XROW + t * 32;
        },
        rg, A, nblk, wave, lane, issue,
        [&] {
          int ra = 0;
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const int u = sb[(kh - 1) * (D::HIN / 2) - 1], v = sb[(kh - 1) * (D::HIN / 2)];
            const int row = (ow > 0 ? u >> 16 : 0) + (int)(int16_t)v + (v >> 16);
            ra += (kh > 0 || oh > 0) ? row : 0;
          }
          return -a.a.z_w * ra;
        });
    QBNN_STAMP_AT(0);
    // ---- shortcut: M over the centre taps of X (1x1 / s2)
    ring_mfma<D, D::KS_S>(
        [&](int ks) { return xlane + D::XROW + D::CIN + ks * 32; },
        rg, S, nblk, wave, lane, issue, [&] { return -a.s.z_w * (int)(int16_t)sb[0]; });
    QBNN_STAMP_AT(1);
    lds_barrier();                       // every wave has read X for the last time: T and SC may overwrite it
    QBNN_STAMP_AT(2);
    if constexpr (DROP && DM::IN_X) {
#pragma unroll
      for (int d = 0; d < 3; ++d) fill_mask_tab<D::G, D::COUT, true, NTHR>(mtab + d * DM::MTB, dr.d[d], s, img0, a.B, tid);
      lds_barrier();
    }
    constexpr int IMG_PX = D::HO * D::HO;
    if constexpr (DROP) {
      {
        EpiDenseTileDrop<D::PIXB_T, D::COUT, IMG_PX, true> epi{tt, a.a, dr.d[0], {mtab, dr.d[0].mq1}, 0};
        epi_presub<EC>(bias_lds + D::COUT, a.a, epi, A, wave, lane);
        const int v = half_sum(epi.csum);
        if (lane < 32) stt[nblk * D::M + mblk * 32 + lane] = v;
      }
      {
        EpiDenseDrop<D::COUT, IMG_PX, false, D::SCP, true> epi{sc, a.s, a.add, dr.d[2], {mtab + 2 * DM::MTB, dr.d[2].mq1}};
        epi_presub<EC>(bias_lds, a.s, epi, S, wave, lane);
      }
    } else {
      {
        EpiDenseTile<D::PIXB_T> epi{tt, a.a, 0};
        epi_presub<EC>(bias_lds + D::COUT, a.a, epi, A, wave, lane);
        const int v = half_sum(epi.csum);                            // channel sum of this wave's 96 channels of T, per pixel
        if (lane < 32) stt[nblk * D::M + mblk * 32 + lane] = v;
      }
      {
        EpiDense<D::COUT, false, D::SCP> epi{sc, a.s, a.add};
        epi_presub<EC>(bias_lds, a.s, epi, S, wave, lane);
      }
    }
    for (int i = tid; i < D::PIXB_T / 4; i += NTHR) reinterpret_cast<uint32_t*>(zline)[i] = 0u;      // (X shared these bytes)
    QBNN_STAMP_AT(3);
    // ---- stem.3: M over T (3x3 / s1, dense tile: taps outside the map read the zero line); E: + SC, ReLU -> SC in place
    {
      int vmask = 0;                   // bit tap = that tap of this pixel's window lies inside the map
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int kh = tap / 3, kw = tap - 3 * kh;
        if ((unsigned)(oh + kh - 1) < (unsigned)D::HO && (unsigned)(ow + kw - 1) < (unsigned)D::HO) vmask |= 1 << tap;
      }
      const uint8_t* tlane = tt + m * D::PIXB_T + 16 * h;
      const uint8_t* zl = zline + 16 * h;
      ring_mfma<D, D::KS_B>(
          [&](int ks) {
            const int tap = ks / D::SPT_B, sub = ks - tap * D::SPT_B, kh = tap / 3, kw = tap - 3 * kh;
            return ((vmask >> tap) & 1 ? tlane + ((kh - 1) * D::HO + (kw - 1)) * D::PIXB_T : zl) + sub * 32;
          },
          rg, A, nblk, wave, lane, issue,
          [&] {                          // window sum from S_T: reads at immediate offsets from the pixel's own entry; one outside the map is masked out (it lands inside the X region)
            int rb = 0;
            const int* sp = stt + m;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
              int sv = sp[(tap / 3 - 1) * D::HO + (tap % 3 - 1)];
              if constexpr (D::NBLKS == 2) sv += sp[D::M + (tap / 3 - 1) * D::HO + (tap % 3 - 1)];
              rb += (vmask >> tap) & 1 ? sv : 0;
            }
            return -a.b.z_w * rb;
          });
    }
    QBNN_STAMP_AT(4);
    fetch(next);                         // the next item's input: in flight during this epilogue and the read-out
    if constexpr (DROP) {
      EpiDenseDrop<D::COUT, IMG_PX, true, D::SCP, true> epi{sc, a.b, a.add, dr.d[1], {mtab + DM::MTB, dr.d[1].mq1}};
      epi_presub<EC>(bias_lds + 2 * D::COUT, a.b, epi, A, wave, lane);
    } else {
      EpiDense<D::COUT, true, D::SCP> epi{sc, a.b, a.add};
      epi_presub<EC>(bias_lds + 2 * D::COUT, a.b, epi, A, wave, lane);
    }
    QBNN_STAMP_AT(5);
    lds_barrier();
    QBNN_STAMP_AT(6);
    // ---- read-out: SC -> registers; barrier (SC shares its bytes with X); next item's input -> X; registers -> HBM (the item's
    //      output block is contiguous)
    {
      uint8_t* ys = a.y + (int64_t)s * a.y_ss + (int64_t)img0 * IMG_OUT;
      int t = tid;
      asm volatile("" : "+v"(t));
      v2i outv[NOUT];
#pragma unroll
      for (int j = 0; j < NOUT; ++j) {
        const int i = t + j * NTHR;
        const int px = i / U8, within = i - px * U8;
        if (i < D::M * U8) outv[j] = *reinterpret_cast<const v2i*>(sc + px * D::SCP + within * 8);
      }
      lds_barrier();
      write_tile(next);                // unconditional (the last item rewrites its own input): an unconsumed prefetch costs a vmcnt(0) guard
#pragma unroll
      for (int j = 0; j < NOUT; ++j) {
        const int i = t + j * NTHR;
        if (i < D::M * U8 && img0 + (i * 8) / IMG_OUT < a.B) *reinterpret_cast<v2i*>(ys + (int64_t)i * 8) = outv[j];
      }
    }
    QBNN_STAMP_AT(7);
  }
  wait_vmcnt<0>();                       // the ring's tail requests land before the workgroup's LDS is handed on
#ifdef QBNN_STAMP
  if (g_stamp_dev && (tid & 63) == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(g_stamp_dev + wave * 8 + i, st_acc[i]);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// 16-wave form with ROLES (round 5; single-call launches without dropout; QBNN_DOWN_R16=0 for the A/B).  The 8-wave kernel above holds two accumulator
// sets per wave (stem.0 + shortcut: 157 / 181 VGPRs, two waves per SIMD), and two thirds of its work item are vector instructions issued at the
// two-wave rate (2.9 - 4.8 cycles each; four waves: 2.0 - 3.4, profiles/r05_issue_bench2.txt).  Here a workgroup has 16 waves of ONE accumulator set
// (<= 128 VGPRs, four per SIMD):
//   waves 0 - 7  ("A"): stem.0 for pass w (pixel tile x the pass's three channel tiles), its epilogue -> T; stem.3 for channel tiles 0, 1 of the pass
//   waves 8 - 15 ("S"): the shortcut for pass w - 8, its epilogue -> SC;                                     stem.3 for channel tile 2 of the pass
// so both first epilogues (48 outputs per lane each) run on all 16 waves, and stem.3's (48 per lane and pass) splits 32 : 16.  The weight ring is the
// 8-wave kernel's: waves 0 - 7 issue the DMA and keep its vmcnt accounting; every wave passes every ring barrier (a wave without MFMAs in a conv just
// walks that conv's advances), so the ring state stays in step.  Same tiles, tables, epilogue functors and bits.
template <class D, int KS, int NBW, int PD, class StepFn, class IssueFn, class InitFn>
__device__ __forceinline__ void ring_mfma_role(StepFn step, WeightRing& rg, ConvAccMN<1, 3>& A, int nt0, bool issuer, int lane, IssueFn issue, InitFn init) {
  constexpr int SLK = D::SLK;
  struct Frag { v4i w[NBW]; v4i x; };
  Frag f[PD + 1];
  const uint8_t* wl = nullptr;
  auto advance = [&]() {
    if (issuer) wait_vmcnt<(D::NBUF - 2) * D::DMA_PER_WAVE>();
    lds_barrier();
    if (issuer) issue(rg.pnext, rg.pbuf);
    ++rg.pnext;
    rg.pbuf = rg.pbuf + 1 == D::NBUF ? 0 : rg.pbuf + 1;
    wl = rg.base + rg.cbuf * D::SLABB + (nt0 * SLK * 64 + lane) * 16;
    rg.cbuf = rg.cbuf + 1 == D::NBUF ? 0 : rg.cbuf + 1;
  };
  auto load = [&](Frag& fr, int ks) {
    const int j = ks % SLK;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) fr.w[nb] = *reinterpret_cast<const v4i*>(wl + (nb * SLK + j) * 1024);
    fr.x = *reinterpret_cast<const v4i*>(step(ks));
  };
  auto mfma = [&](const Frag& fr) {
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) A.acc[0][nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fr.w[nb], fr.x, A.acc[0][nb], 0, 0, 0);
  };
#pragma unroll
  for (int p = 0; p < PD && p < KS; ++p) {
    if (p % SLK == 0) advance();
    if (p == 0) {
      const int a0 = init();
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) A.acc[0][nb][i] = a0;
    }
    load(f[p % (PD + 1)], p);
  }
#pragma unroll
  for (int j = 0; j < KS; ++j) {
    const int p = j + PD;
    if (p < KS) {
      if (p % SLK == 0) advance();
      load(f[p % (PD + 1)], p);
    }
    mfma(f[j % (PD + 1)]);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// a conv this wave has no MFMAs in: its ring advances only (the barriers, and the DMA requests if the wave is an issuer)
template <class D, int KS, class IssueFn>
__device__ __forceinline__ void ring_follow(WeightRing& rg, bool issuer, IssueFn issue) {
#pragma unroll
  for (int p = 0; p < KS; p += D::SLK) {
    if (issuer) wait_vmcnt<(D::NBUF - 2) * D::DMA_PER_WAVE>();
    lds_barrier();
    if (issuer) issue(rg.pnext, rg.pbuf);
    ++rg.pnext;
    rg.pbuf = rg.pbuf + 1 == D::NBUF ? 0 : rg.pbuf + 1;
    rg.cbuf = rg.cbuf + 1 == D::NBUF ? 0 : rg.cbuf + 1;
  }
}
// conv_epi_phase_with's PRESUB form over NBW of the pass's three channel tiles, starting at tile n0 (accumulator slots 0 .. NBW - 1)
template <class C, int NBW, class Epi>
__device__ __forceinline__ void epi_presub_sub(const float* bias_lds, const QConv& p, Epi& epi, ConvAccMN<1, 3>& A, int pass, int lane, int n0) {
  const int r = lane & 31, h = lane >> 5;
  const int mblk = pass / C::NBLKS, nblk = pass - mblk * C::NBLKS;
  const int po = epi.pixel(mblk * 32 + r);
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int cb = (nblk * C::NB + n0 + nb) * 32 + 4 * h;
    float4 b4[4];
    uint32_t pre[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) b4[g4] = *reinterpret_cast<const float4*>(bias_lds + cb + 8 * g4);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) pre[g4] = epi.load(po, cb + 8 * g4);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float4 bb = b4[g4];
      const float v0 = __builtin_fmaf(bb.x, p.rcp, (float)A.acc[0][nb][4 * g4 + 0]) * p.mult;
      const float v1 = __builtin_fmaf(bb.y, p.rcp, (float)A.acc[0][nb][4 * g4 + 1]) * p.mult;
      const float v2 = __builtin_fmaf(bb.z, p.rcp, (float)A.acc[0][nb][4 * g4 + 2]) * p.mult;
      const float v3 = __builtin_fmaf(bb.w, p.rcp, (float)A.acc[0][nb][4 * g4 + 3]) * p.mult;
      epi.store(po, cb + 8 * g4, v0, v1, v2, v3, pre[g4]);
    }
  }
}

#ifndef QBNN_DOWN_R16_PD
#define QBNN_DOWN_R16_PD 2
#endif

template <class D, class EC>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4)))
void block_down_ring16_kernel(const ArgsArr<DownArgs, 1> all) {
  const DownArgs a = args_of(all, 0);
  constexpr int NTHR = 1024, PD = QBNN_DOWN_R16_PD;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* xt = smem;
  uint8_t* tt = smem;
  uint8_t* zline = tt + D::M * D::PIXB_T;
  uint8_t* sc = smem + D::SC_OFF;
  uint8_t* rbase = smem + D::X_BYTES;
  float* bias_lds = reinterpret_cast<float*>(rbase + D::NBUF * D::SLABB);
  int16_t* sx16 = reinterpret_cast<int16_t*>(bias_lds + 3 * D::COUT);
  int* stt = reinterpret_cast<int*>(smem + D::ST_OFF);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool role_a = wave < 8;                  // also: the waves that feed the weight ring
  const int pass = wave & 7;
  const int mblk = pass / D::NBLKS, nblk = pass - mblk * D::NBLKS;

  constexpr int CPP = D::CIN / 16;
  // input traffic: TPP adjacent lanes share a pixel and move CPT consecutive 16-byte chunks of it each (the pixel's channel sum: thread-local v_dot4
  // chain + one DPP exchange between the pair, ONE 16-bit store into S_X)
  constexpr int TPP = NTHR > D::NPX ? NTHR / D::NPX : 1, PPT = NTHR > D::NPX ? 1 : D::NPX / NTHR, CPT = CPP / TPP, PER_T = PPT * CPT, PXS = NTHR / TPP;
  static_assert(CPP % TPP == 0 && (TPP == 1 || TPP == 2) && PPT * PXS == D::NPX, "every thread moves the same share of whole pixels");
  constexpr int IMG_IN = D::HIN * D::HIN * D::CIN, IMG_OUT = D::HO * D::HO * D::COUT, U8 = D::COUT / 8;
  constexpr int NOUT = (D::M * U8 + NTHR - 1) / NTHR;
  const int groups = (a.B + D::G - 1) / D::G;
  const ItemWalk walk(a.n_samples * groups, blockIdx.x, gridDim.x);
  const int count = walk.count;

  load_bias<D::COUT, NTHR>(bias_lds, a.s.bias, tid);
  load_bias<D::COUT, NTHR>(bias_lds + D::COUT, a.a.bias, tid);
  load_bias<D::COUT, NTHR>(bias_lds + 2 * D::COUT, a.b.bias, tid);
  if (count <= 0) return;

  v4i pre[PER_T];
  auto fetch = [&](int item) {
    const int s = item / groups, img0 = (item - s * groups) * D::G;
    const uint8_t* xs = a.x + (int64_t)s * a.x_ss + (int64_t)img0 * IMG_IN;
    const int valid = (a.B - img0 < D::G ? a.B - img0 : D::G) * D::HIN * D::HIN;
    int t = tid;
    asm volatile("" : "+v"(t));
    const int px0 = t / TPP, part = t - px0 * TPP;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int px = px0 + j * PXS;
      const uint8_t* p = xs + (px < valid ? (int64_t)px * D::CIN : 0) + part * (CPT * 16);
#pragma unroll
      for (int c = 0; c < CPT; ++c) pre[j * CPT + c] = *reinterpret_cast<const v4i*>(p + 16 * c);
    }
  };
  auto write_tile = [&](int item) {
    const int s = item / groups, img0 = (item - s * groups) * D::G;
    const int valid = (a.B - img0 < D::G ? a.B - img0 : D::G) * D::HIN * D::HIN;
    const uint32_t z4 = (uint32_t)a.z_in * 0x01010101u;
    int t = tid;
    asm volatile("" : "+v"(t));
    const int px0 = t / TPP, part = t - px0 * TPP;
    {
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        const int px = px0 + j * PXS;
        const int g = px / (D::HIN * D::HIN), rem = px - g * (D::HIN * D::HIN), row = rem / D::HIN, col = rem - row * D::HIN;
        uint8_t* dst = xt + g * D::XIMG + (row + 1) * D::XROW + (col + 1) * D::CIN + part * (CPT * 16);
        int sum = 0;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
          const v4i v = pre[j * CPT + c];
          const v4i q = px < valid ? v4i{(int)sub_bytes(v.x, z4), (int)sub_bytes(v.y, z4), (int)sub_bytes(v.z, z4), (int)sub_bytes(v.w, z4)} : v4i{0, 0, 0, 0};
          *reinterpret_cast<v4i*>(dst + 16 * c) = q;
          sum = __builtin_amdgcn_sdot4(q.x, 0x01010101, sum, false);
          sum = __builtin_amdgcn_sdot4(q.y, 0x01010101, sum, false);
          sum = __builtin_amdgcn_sdot4(q.z, 0x01010101, sum, false);
          sum = __builtin_amdgcn_sdot4(q.w, 0x01010101, sum, false);
        }
        if constexpr (TPP == 2) sum += __builtin_amdgcn_update_dpp(0, sum, 0xB1, 0xf, 0xf, false);      // quad_perm [1, 0, 3, 2]: the pixel's other half
        if (part == 0) sx16[px] = (int16_t)sum;
      }
    }
    constexpr int TOP = D::XTW * D::CIN / 16, CW = D::CIN / 16;
    for (int i = t; i < D::G * D::NHALO; i += NTHR) {
      const int g = i / D::NHALO, q = i - g * D::NHALO;
      const int off = q < TOP ? q * 16 : (1 + (q - TOP) / CW) * D::XROW + ((q - TOP) % CW) * 16;
      *reinterpret_cast<v4i*>(xt + g * D::XIMG + off) = v4i{0, 0, 0, 0};
    }
  };
  const uint32_t lane16 = (uint32_t)lane * 16u;
  auto issue = [&](int q, int buf) {          // (called by waves 0 - 7 only: `wave` is the issuing wave's index among the eight)
    int itx = q / D::NSI;
    const int loc = q - itx * D::NSI;
    itx = itx < count ? itx : count - 1;
    const int s = walk.item(itx) / groups;
    const int8_t* wq; int KS, slab;
    if (loc < D::NS_A) { wq = a.a.w + (int64_t)s * a.a.w_ss; KS = D::KS_A; slab = loc; }
    else if (loc < D::NS_A + D::NS_S) { wq = a.s.w + (int64_t)s * a.s.w_ss; KS = D::KS_S; slab = loc - D::NS_A; }
    else { wq = a.b.w + (int64_t)s * a.b.w_ss; KS = D::KS_B; slab = loc - D::NS_A - D::NS_S; }
    const uint32_t dst = lds_addr_of(rbase + buf * D::SLABB);
#pragma unroll
    for (int k = 0; k < D::DMA_PER_WAVE; ++k) {
      const int f = wave + 8 * k;
      const int nt = f / D::SLK, u = f - nt * D::SLK;
      int ks = slab * D::SLK + u;
      ks = ks < KS ? ks : KS - 1;
      dma16_s(wq + (int64_t)(nt * KS + ks) * 1024, lane16, dst + f * 1024);
    }
  };

  fetch(walk.item(0));
  write_tile(walk.item(0));
  WeightRing rg{rbase, 0, D::NBUF - 1, D::NBUF - 1};
  if (role_a) {
#pragma unroll
    for (int q = 0; q < D::NBUF - 1; ++q) issue(q, q);
  }
  ConvAccMN<1, 3> A;
  for (int it = 0; it < count; ++it) {
    const int item = walk.item(it);
    const int s = item / groups, img0 = (item - s * groups) * D::G;
    const bool more = it + 1 < count;
    const int next = more ? walk.item(it + 1) : item;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int r = ln & 31, h = ln >> 5;
    const int m = mblk * 32 + r;
    const int g = m / (D::HO * D::HO), rem = m - g * (D::HO * D::HO), oh = rem / D::HO, ow = rem - oh * D::HO;
    const uint8_t* xlane = xt + g * D::XIMG + (2 * oh) * D::XROW + (2 * ow) * D::CIN + 16 * h;
    const int* sb = reinterpret_cast<const int*>(sx16) + (((g * D::HIN + 2 * oh) * D::HIN + 2 * ow) >> 1);
    // ---- stem.0 (A waves) | shortcut (S waves): both over X
    if (role_a) {
      ring_mfma_role<D, D::KS_A, 3, PD>(
          [&](int ks) {
            const int kh = ks / D::SPR_A, t = ks - kh * D::SPR_A;
            return xlane + kh * D::XROW + t * 32;
          },
          rg, A, nblk * 3, true, lane, issue,
          [&] {
            int ra = 0;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
              const int u = sb[(kh - 1) * (D::HIN / 2) - 1], v = sb[(kh - 1) * (D::HIN / 2)];
              const int row = (ow > 0 ? u >> 16 : 0) + (int)(int16_t)v + (v >> 16);
              ra += (kh > 0 || oh > 0) ? row : 0;
            }
            return -a.a.z_w * ra;
          });
      ring_follow<D, D::KS_S>(rg, true, issue);
    } else {
      ring_follow<D, D::KS_A>(rg, false, issue);
      ring_mfma_role<D, D::KS_S, 3, PD>(
          [&](int ks) { return xlane + D::XROW + D::CIN + ks * 32; },
          rg, A, nblk * 3, false, lane, issue, [&] { return -a.s.z_w * (int)(int16_t)sb[0]; });
    }
    lds_barrier();                       // every wave has read X for the last time: T and SC may overwrite it
    if (role_a) {
      EpiDenseTile<D::PIXB_T> epi{tt, a.a, 0};
      epi_presub<EC>(bias_lds + D::COUT, a.a, epi, A, pass, lane);
      const int v = half_sum(epi.csum);
      if (lane < 32) stt[nblk * D::M + mblk * 32 + lane] = v;
    } else {
      EpiDense<D::COUT, false, D::SCP> epi{sc, a.s, a.add};
      epi_presub<EC>(bias_lds, a.s, epi, A, pass, lane);
    }
    for (int i = tid; i < D::PIXB_T / 4; i += NTHR) reinterpret_cast<uint32_t*>(zline)[i] = 0u;
    // ---- stem.3: M over T; A waves channel tiles 0, 1 of the pass, S waves tile 2
    {
      int vmask = 0;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int kh = tap / 3, kw = tap - 3 * kh;
        if ((unsigned)(oh + kh - 1) < (unsigned)D::HO && (unsigned)(ow + kw - 1) < (unsigned)D::HO) vmask |= 1 << tap;
      }
      const uint8_t* tlane = tt + m * D::PIXB_T + 16 * h;
      const uint8_t* zl = zline + 16 * h;
      auto step_b = [&](int ks) {
        const int tap = ks / D::SPT_B, sub = ks - tap * D::SPT_B, kh = tap / 3, kw = tap - 3 * kh;
        return ((vmask >> tap) & 1 ? tlane + ((kh - 1) * D::HO + (kw - 1)) * D::PIXB_T : zl) + sub * 32;
      };
      auto init_b = [&] {
        int rb = 0;
        const int* sp = stt + m;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          int sv = sp[(tap / 3 - 1) * D::HO + (tap % 3 - 1)];
          if constexpr (D::NBLKS == 2) sv += sp[D::M + (tap / 3 - 1) * D::HO + (tap % 3 - 1)];
          rb += (vmask >> tap) & 1 ? sv : 0;
        }
        return -a.b.z_w * rb;
      };
      if (role_a) ring_mfma_role<D, D::KS_B, 2, PD>(step_b, rg, A, nblk * 3, true, lane, issue, init_b);
      else ring_mfma_role<D, D::KS_B, 1, PD>(step_b, rg, A, nblk * 3 + 2, false, lane, issue, init_b);
    }
    fetch(next);
    {
      EpiDense<D::COUT, true, D::SCP> epi{sc, a.b, a.add};
      if (role_a) epi_presub_sub<EC, 2>(bias_lds + 2 * D::COUT, a.b, epi, A, pass, lane, 0);
      else epi_presub_sub<EC, 1>(bias_lds + 2 * D::COUT, a.b, epi, A, pass, lane, 2);
    }
    lds_barrier();
    {
      uint8_t* ys = a.y + (int64_t)s * a.y_ss + (int64_t)img0 * IMG_OUT;
      int t = tid;
      asm volatile("" : "+v"(t));
      v2i outv[NOUT];
#pragma unroll
      for (int j = 0; j < NOUT; ++j) {
        const int i = t + j * NTHR;
        const int px = i / U8, within = i - px * U8;
        if (i < D::M * U8) outv[j] = *reinterpret_cast<const v2i*>(sc + px * D::SCP + within * 8);
      }
      lds_barrier();
      write_tile(next);
#pragma unroll
      for (int j = 0; j < NOUT; ++j) {
        const int i = t + j * NTHR;
        if (i < D::M * U8 && img0 + (i * 8) / IMG_OUT < a.B) *reinterpret_cast<v2i*>(ys + (int64_t)i * 8) = outv[j];
      }
    }
  }
  wait_vmcnt<0>();
}

template <class D, class EC>
int launch_r16(const DownArgs& a, hipStream_t st) {
  static std::atomic<uint64_t> attr{0};
  if (int rc = ensure_dyn_lds((const void*)block_down_ring16_kernel<D, EC>, attr, D::LDS)) return rc;
  const int items = a.n_samples * ((a.B + D::G - 1) / D::G);
  ArgsArr<DownArgs, 1> one;
  one.m[0] = a;
  hipLaunchKernelGGL((block_down_ring16_kernel<D, EC>), dim3(items < 256 ? (items > 0 ? items : 1) : 256), dim3(1024), D::LDS, st, one);
  return check_launch("qbnn_block_down_i8_mc");
}

template <class D, class EC>
int launch_by_value(const DownArgs* arr, int n, hipStream_t st) {
  static std::atomic<uint64_t> attr1{0}, attrN{0};
  int items = 0;
  for (int i = 0; i < n; ++i) { const int it = arr[i].n_samples * ((arr[i].B + D::G - 1) / D::G); items = it > items ? it : items; }
  if (n == 1) {
    if (int rc = ensure_dyn_lds((const void*)block_down_ring_kernel<D, EC, 1>, attr1, D::LDS)) return rc;
    ArgsArr<DownArgs, 1> one;
    one.m[0] = arr[0];
    hipLaunchKernelGGL((block_down_ring_kernel<D, EC, 1>), dim3(items < 256 ? (items > 0 ? items : 1) : 256), dim3(512), D::LDS, st, one, DropSet<0>{});
    return check_launch("qbnn_block_down_i8_mc");
  }
  static_assert(sizeof(ArgsArr<DownArgs, QBNN_FUSED_CALLS>) <= 3840, "kernel arguments are limited to 4 KiB (incl. the hidden ones)");
  if (int rc = ensure_dyn_lds((const void*)block_down_ring_kernel<D, EC, QBNN_FUSED_CALLS>, attrN, D::LDS)) return rc;
  ArgsArr<DownArgs, QBNN_FUSED_CALLS> all;
  memset(&all, 0, sizeof(all));
  for (int i = 0; i < n; ++i) all.m[i] = arr[i];
  const int per = 256 / n > 0 ? 256 / n : 1;
  hipLaunchKernelGGL((block_down_ring_kernel<D, EC, QBNN_FUSED_CALLS>), dim3(items < per ? (items > 0 ? items : 1) : per, n), dim3(512), D::LDS, st, all, DropSet<0>{});
  return check_launch("qbnn_block_down_i8_multi");
}

template <class D, class EC>
int launch_dev(const DownArgs* dev, int n, int items, hipStream_t st) {
  static std::atomic<uint64_t> attr{0};
  if (int rc = ensure_dyn_lds((const void*)block_down_ring_kernel<D, EC, 0>, attr, D::LDS)) return rc;
  const int per = 256 / n > 0 ? 256 / n : 1;
  hipLaunchKernelGGL((block_down_ring_kernel<D, EC, 0>), dim3(items < per ? (items > 0 ? items : 1) : per, n), dim3(512), D::LDS, st, ArgsArr<DownArgs, 0>{dev}, DropSet<0>{});
  return check_launch("qbnn_block_down_i8_multi_launch");
}

template <class D, class EC>
int launch_drop(const DownArgs& a, const DropSet<3>& dr, hipStream_t st) {
  constexpr int LDS = DRMask<D>::LDS;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static std::atomic<uint64_t> attr{0};
  if (int rc = ensure_dyn_lds((const void*)block_down_ring_kernel<D, EC, 1, true>, attr, LDS)) return rc;
  const int items = a.n_samples * ((a.B + D::G - 1) / D::G);
  ArgsArr<DownArgs, 1> one;
  one.m[0] = a;
  hipLaunchKernelGGL((block_down_ring_kernel<D, EC, 1, true>), dim3(items < 256 ? (items > 0 ? items : 1) : 256), dim3(512), LDS, st, one, dr);
  return check_launch("qbnn_block_down_drop_i8_mc");
}

}  // namespace

int qbnn_launch_block_down_ring_drop(const DownArgs& a, const DropSet<3>& dr, int Cin, hipStream_t st) {
  if (Cin == 48) return launch_drop<DR48, E48>(a, dr, st);
  if (Cin == 96) return launch_drop<DR96, E96>(a, dr, st);
  return fail(QBNN_E_INVALID, "qbnn_block_down_drop (ring): 48 -> 96 and 96 -> 192 channels only%s");
}

int qbnn_launch_block_down_ring(const DownArgs* arr, int n, int Cin, hipStream_t st) {
  if (n <= 0 || n > QBNN_FUSED_CALLS) return fail(QBNN_E_INVALID, "qbnn_block_down (ring): 1 .. 8 argument blocks per launch%s");
  static const int r16 = [] { const char* e = getenv("QBNN_DOWN_R16"); return e ? atoi(e) : 3; }();      // bit 0: 48 -> 96, bit 1: 96 -> 192 (measured on one box: 0.324 -> 0.305 and 0.255 -> 0.245 ms)
  if (n == 1 && Cin == 48 && (r16 & 1)) return launch_r16<DR48, E48>(arr[0], st);
  if (n == 1 && Cin == 96 && (r16 & 2)) return launch_r16<DR96, E96>(arr[0], st);
  if (Cin == 48) return launch_by_value<DR48, E48>(arr, n, st);
  if (Cin == 96) return launch_by_value<DR96, E96>(arr, n, st);
  return fail(QBNN_E_INVALID, "qbnn_block_down (ring): 48 -> 96 and 96 -> 192 channels only%s");
}

int qbnn_launch_block_down_ring_dev(const DownArgs* dev, int n, int items, int Cin, hipStream_t st) {
  if (Cin == 48) return launch_dev<DR48, E48>(dev, n, items, st);
  if (Cin == 96) return launch_dev<DR96, E96>(dev, n, items, st);
  return fail(QBNN_E_INVALID, "qbnn_block_down (ring): 48 -> 96 and 96 -> 192 channels only%s");
}
