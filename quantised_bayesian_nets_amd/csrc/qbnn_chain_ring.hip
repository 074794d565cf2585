// libqbnn_hip.so -- the WIDE identity BasicBlocks (reference models_bbb.py:146-183, stride 1: stem.0 3x3 ConvReLU, stem.3 3x3 conv, Add,
// ReLU) at 96 channels (8 x 8 maps) and 192 channels (4 x 4) with the K loop of qbnn_down_ring.hip.  Tile, epilogues, channel-sum tables
// and argument block are those of round 3's block_chain_ald_kernel (out of the build since round 6: tools/experiments/r03_block_chain_ald_kernel.hip.txt).
//
// What changed (round 4): block_chain_ald_kernel ran a conv's K loop slab by slab over a TWO-slab ring -- `s_waitcnt vmcnt(0)`, barrier,
// request the next slab, then the slab's k-steps with their LDS fragment reads one step ahead: at every slab the software pipeline restarts
// from an empty state and the one slab in flight has to land within one slab's worth of MFMAs (stamps: 15.4 k / 35.2 k cycles per M phase
// against 10.4 k / 20.7 k of MFMAs at 96 / 192 channels).  Here, as in the ring form of the down-sampling blocks:
//   * the ring is NBUF = 4 slabs of 24 KiB with three in flight, requested from inline assembly with the wave's own vmcnt accounting;
//   * a conv's k-steps are ONE fully unrolled stream whose fragments (3 weight + MB pixel fragments per step) are requested PD k-steps
//     ahead; the ring advance (wait, barrier, next request) sits inside the stream, so the pipeline never drains inside a conv;
//   * both convs' slabs of an item form one flat sequence (a slab never spans two convs; a conv's last slab may be short).
// Phases per work item:  M_a | barrier | E_a (-> T over X, channel sums)  M_b (residual + next input requested in its last slab) | barrier |
//                        E_b (+ residual from global memory, ReLU -> tile) | barrier | read-out, next X over the tile, stores
#include "qbnn_host.h"

#ifdef QBNN_STAMP      // diagnostic build only (tools/stamp_ring.py): per-phase s_memtime sums; the shipped library has none of this
QBNN_EXPORT void qbnn_debug_stamp_buffer_chain_ring(void* p) { hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_dev_ptr), &p, sizeof(p)); }
#endif

namespace {

template <class C_, int NBUF_ = 4>
struct CRCfg {
  using C = C_;
  static constexpr int NT = C::NT, NB = 3, MB = C::MB, NBLKS = NT / NB, KS = C::KS, SPT = C::SPT, HO = C::HO, PIXB = C::PIXB;
  static constexpr int IMG_PX = HO * HO, M = C::G * IMG_PX;
  static constexpr int TILE = M * PIXB;                       // dense [image][oh][ow][C + 16], followed by the zero line (PIXB bytes)
  static constexpr int NBUF = NBUF_;
  static constexpr int NF = 24, SLK = NF / NT, DMA_PER_WAVE = NF / 8, SLABB = NF * 1024;
  static constexpr int NS = (KS + SLK - 1) / SLK, NSI = 2 * NS;
  static constexpr int LDS = TILE + PIXB + NBUF * SLABB + 2 * C::COUT * 4 + 2 * M * 4 + 64;      // (+ slack: a masked window-sum read may lie HO + 1 entries behind the tables)
  static_assert(C::NB == 3 && NT % 3 == 0 && NF % NT == 0, "three channel tiles per wave; 24 fragments per slab");
  static_assert(C::NPASS == 8, "one (MB pixel tiles, channel block) pass per wave");
  static_assert(C::CIN == C::COUT && C::CIN % 32 == 0 && C::PADB > 0 && C::STRIDE == 1 && C::KSZ == 3, "wide identity BasicBlock on a dense padded tile");
  static_assert((TILE + PIXB) % 16 == 0, "ring alignment");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

// One conv's M phase over the dense tile: see the header comment and ring_mfma in qbnn_down_ring.hip.  vm[mb]: bit `tap` = that tap of this
// lane's pixel in M-tile mb lies inside the map; tl[mb]: the pixel's own bytes (k-half included); zl: the zero line.  EXTRA: vector-memory
// instructions the caller issued between the previous conv's last ring request and this call (they are younger than the first slabs this
// conv waits for, and vmcnt counts them).  tail() runs right after
// the conv's last ring advance -- the place for global loads that should be in flight during the last slab (nothing is requested from the
// ring after it, so the compiler's own vmcnt waits for those loads are not made stricter by DMA traffic it cannot see).
template <class D, int PD, int EXTRA, class IssueFn, class TailFn>
__device__ __forceinline__ void ring_mfma_dense(const uint8_t* const (&tl)[D::MB], const int (&vm)[D::MB], const uint8_t* zl, const int* tab, int z_w,
                                                int mblk, WeightRing& rg, ConvAccMN<D::MB, 3>& A, int nblk, int lane, IssueFn issue, TailFn tail) {
  constexpr int SLK = D::SLK, KS = D::KS, MB = D::MB;
  // The accumulators start at -z_w R(p) (sampled weights have a non-zero zero point: sum x'(W - z_w) = acc - z_w R), R = the window sum of
  // the centred tile bytes from the per-pixel channel sums in `tab`: the pixel's own entry and its 8 neighbours at immediate offsets (an
  // entry outside the map is read -- it lies inside the workgroup's LDS -- and masked out).  The table is complete once the conv's first
  // ring barrier has passed (its writers: the tile write / the previous conv's epilogue), hence init() in the stream below.
  auto init = [&]() {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int* sp = tab + (mblk * MB + mb) * 32 + (lane & 31);
      int R = 0;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int v = sp[(tap / 3 - 1) * D::HO + (tap % 3 - 1)];
        R += (vm[mb] >> tap) & 1 ? v : 0;
      }
      const int a0 = -z_w * R;
#pragma unroll
      for (int nb = 0; nb < 3; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) A.acc[mb][nb][i] = a0;
    }
  };
  struct Frag { v4i w[3]; v4i x[MB]; };
  Frag f[PD + 1];
  const uint8_t* wl = nullptr;
  auto advance = [&](auto extra) {
    // this wave's share of the slab has landed: every vector-memory operation it issued AFTER that slab's may still be in flight -- the
    // next NBUF - 2 slabs and, for the conv's first NBUF - 1 advances, the EXTRA loads the caller issued right in front of the conv
    wait_vmcnt<(D::NBUF - 2) * D::DMA_PER_WAVE + decltype(extra)::value>();
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
    const int tap = ks / D::SPT, sub = ks - tap * D::SPT, kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
      fr.x[mb] = *reinterpret_cast<const v4i*>(((vm[mb] >> tap) & 1 ? tl[mb] + ((kh - 1) * D::HO + (kw - 1)) * D::PIXB : zl) + sub * 32);
  };
  auto mfma = [&](const Frag& fr) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < 3; ++nb) A.acc[mb][nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fr.w[nb], fr.x[mb], A.acc[mb][nb], 0, 0, 0);
  };
  constexpr int LAST = (D::NS - 1) * SLK;               // first k-step of the conv's last slab
  auto request = [&](int p) {
    if (p % SLK == 0) {
      if (p / SLK < D::NBUF - 1) advance(std::integral_constant<int, EXTRA>{});
      else advance(std::integral_constant<int, 0>{});
      if (p == 0) init();
      if (p == LAST) tail();
    }
    load(f[p % (PD + 1)], p);
  };
#pragma unroll
  for (int p = 0; p < PD && p < KS; ++p) request(p);
#pragma unroll
  for (int j = 0; j < KS; ++j) {
    if (j + PD < KS) request(j + PD);
    mfma(f[j % (PD + 1)]);
    __builtin_amdgcn_sched_barrier(0);                  // keep the steps apart: the request distance is the point
  }
}

// DROP (conv_resnet_mc, mcdropout/models_mc.py:116-160): a quantised channel dropout behind each conv -- dr.d[0] behind stem.0 (stem.3 of the
// reference's Sequential), dr.d[1] behind stem.3 (the Add's first operand); one bit per (image, channel) in LDS (a Bernoulli mask has two quantised
// values), applied by the epilogues to the centred integer they hold.
template <class C, int PD, int NM, bool DROP = false, int NBUF = 4>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(NBUF == 2 ? 4 : 2, NBUF == 2 ? 4 : 2)))
void block_chain_ring_kernel(const ArgsArr<ChainArgs<1>, NM> all, const DropSet<DROP ? 2 : 0> dr) {
  using D = CRCfg<C, NBUF>;
  static_assert(!DROP || NM == 1, "dropout variants are single-call");
  constexpr int MTB = MaskTab<C::COUT, true>::bytes(C::G);
  const ChainArgs<1> a = args_of(all, blockIdx.y);
  constexpr int NTHR = 512;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* xt = smem;                                                        // dense tile + zero line
  uint8_t* rbase = smem + D::TILE + D::PIXB;                                 // four weight slabs
  float* bias_lds = reinterpret_cast<float*>(rbase + D::NBUF * D::SLABB);    // [2][COUT]
  int* sx = reinterpret_cast<int*>(bias_lds + 2 * C::COUT);                  // channel sums of the X tile  [M]
  int* stab = sx + D::M;                                                     // ... of the T tile
  uint8_t* mtab = reinterpret_cast<uint8_t*>(stab + D::M) + 64;              // DROP: bit tables of the two dropouts (behind the tables' slack)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mblk = wave / D::NBLKS, nblk = wave - mblk * D::NBLKS;
  const BlockParams& bp = a.blk[0];

  constexpr int IMG_PX = D::IMG_PX;
  constexpr int CPP = C::CIN / 16;                                           // 16-byte chunks per pixel
  // tile traffic: TPP adjacent lanes share a pixel, each moves CPT consecutive 16-byte chunks of it -- the pixel's channel sum is then a
  // thread-local v_dot4 chain (+ a DPP exchange between the TPP lanes) and ONE plain store into the table, no LDS atomics
  constexpr int TPP = NTHR / D::M, CPT = CPP / TPP, PER_T = CPT;
  static_assert(TPP * D::M == NTHR && CPT * TPP == CPP && (TPP == 1 || TPP == 2 || TPP == 4), "every thread moves the same share of one pixel");
  const int groups = (a.B + C::G - 1) / C::G;
  const ItemWalk walk(a.n_samples * groups, blockIdx.x, gridDim.x);          // interleaved per XCD: a sample's weights stay in ONE L2
  const int count = walk.count;

  for (int i = tid; i < D::PIXB / 4; i += NTHR) reinterpret_cast<uint32_t*>(xt + D::TILE)[i] = 0u;
  for (int i = tid; i < D::M; i += NTHR) stab[i] = 0;
  load_bias<C::COUT, NTHR>(bias_lds, bp.a.bias, tid);
  load_bias<C::COUT, NTHR>(bias_lds + C::COUT, bp.b.bias, tid);
  if (count <= 0) return;
  auto dot16 = [](const v4i& c, int d) {
    d = __builtin_amdgcn_sdot4(c.x, 0x01010101, d, false);
    d = __builtin_amdgcn_sdot4(c.y, 0x01010101, d, false);
    d = __builtin_amdgcn_sdot4(c.z, 0x01010101, d, false);
    return __builtin_amdgcn_sdot4(c.w, 0x01010101, d, false);
  };
  auto pixel_sum = [](int d) {       // over the TPP lanes of a pixel (adjacent lanes of one quad)
    if constexpr (TPP >= 2) d += __builtin_amdgcn_update_dpp(0, d, 0xB1, 0xf, 0xf, false);      // quad_perm [1, 0, 3, 2]
    if constexpr (TPP == 4) d += __builtin_amdgcn_update_dpp(0, d, 0x4E, 0xf, 0xf, false);      // quad_perm [2, 3, 0, 1]
    return d;
  };

  // an item's images are contiguous in HBM: chunk i of the item is byte 16 i of that block
  v4i pre[PER_T];
  auto fetch = [&](int item) {
    const int s = item / groups, img0 = (item - s * groups) * C::G;
    const uint8_t* xs = a.x + (int64_t)s * a.x_ss + (int64_t)img0 * IMG_PX * C::CIN;
    const int valid = (a.B - img0 < C::G ? a.B - img0 : C::G) * IMG_PX;
    int t = tid;
    asm volatile("" : "+v"(t));         // per-thread addresses are recomputed here, not hoisted out of the item loop (spills)
    const int px = t / TPP, part = t - px * TPP;
    const uint8_t* p = xs + (px < valid ? (int64_t)px * C::CIN + part * (CPT * 16) : 0);
#pragma unroll
    for (int j = 0; j < CPT; ++j) pre[j] = *reinterpret_cast<const v4i*>(p + 16 * j);
  };
  auto write_tile = [&](int item) {
    const int s = item / groups, img0 = (item - s * groups) * C::G;
    const int valid = (a.B - img0 < C::G ? a.B - img0 : C::G) * IMG_PX;
    const uint32_t z4 = (uint32_t)a.z_in * 0x01010101u;
    int t = tid;
    asm volatile("" : "+v"(t));
    const int px = t / TPP, part = t - px * TPP;
    uint8_t* dst = xt + px * D::PIXB + part * (CPT * 16);
    int sum = 0;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const v4i v = pre[j];
      const v4i c = px < valid ? v4i{(int)sub_bytes(v.x, z4), (int)sub_bytes(v.y, z4), (int)sub_bytes(v.z, z4), (int)sub_bytes(v.w, z4)} : v4i{0, 0, 0, 0};
      *reinterpret_cast<v4i*>(dst + 16 * j) = c;
      sum = dot16(c, sum);
    }
    sum = pixel_sum(sum);
    if (part == 0) sx[px] = sum;
  };
  // request flat slab q (item q / NSI of this workgroup's walk; beyond its last item: that item's slabs again -- harmless, and every
  // wave keeps issuing DMA_PER_WAVE instructions per slab, which is what the vmcnt accounting counts on) into ring buffer `buf`
  const uint32_t lane16 = (uint32_t)lane * 16u;      // (dma16_s: the per-lane part of every fragment address)
  auto issue = [&](int q, int buf) {
    int itx = q / D::NSI;
    const int loc = q - itx * D::NSI;
    itx = itx < count ? itx : count - 1;
    const int s = walk.item(itx) / groups;
    const bool second = loc >= D::NS;
    const int slab = second ? loc - D::NS : loc;
    const int8_t* wq = second ? bp.b.w + (int64_t)s * bp.b.w_ss : bp.a.w + (int64_t)s * bp.a.w_ss;
    const uint32_t dst = lds_addr_of(rbase + buf * D::SLABB);
#pragma unroll
    for (int k = 0; k < D::DMA_PER_WAVE; ++k) {
      const int f = wave + 8 * k;
      const int nt = f / D::SLK, u = f - nt * D::SLK;
      int ks = slab * D::SLK + u;
      ks = ks < D::KS ? ks : D::KS - 1;
      dma16_s(wq + (int64_t)(nt * D::KS + ks) * 1024, lane16, dst + f * 1024);
    }
  };

  fetch(walk.item(0));
  write_tile(walk.item(0));
  __syncthreads();                                   // (the T table is zero before stem.0's first epilogue adds into it)
  WeightRing rg{rbase, 0, D::NBUF - 1, D::NBUF - 1};
#pragma unroll
  for (int q = 0; q < D::NBUF - 1; ++q) issue(q, q);
  ConvAcc<C> A;
  QBNN_STAMP_DECL
  for (int it = 0; it < count; ++it) {
    QBNN_STAMP_START();
    const int item = walk.item(it);
    const int s = item / groups, img0 = (item - s * groups) * C::G;
    const bool more = it + 1 < count;
    const int next = more ? walk.item(it + 1) : item;
    // this lane's pixels (recomputed per item from an opaque lane index: held across the loop the addresses spill)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int r = ln & 31, h = ln >> 5;
    const uint8_t* tl[D::MB];
    int vm[D::MB];
#pragma unroll
    for (int mb = 0; mb < D::MB; ++mb) {
      const int m = (mblk * D::MB + mb) * 32 + r;
      const int rem = m % IMG_PX, oh = rem / D::HO, ow = rem % D::HO;
      int v = 0;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int kh = tap / 3, kw = tap - 3 * kh;
        if ((unsigned)(oh + kh - 1) < (unsigned)D::HO && (unsigned)(ow + kw - 1) < (unsigned)D::HO) v |= 1 << tap;
      }
      vm[mb] = v;
      tl[mb] = xt + m * D::PIXB + 16 * h;
    }
    const uint8_t* zl = xt + D::TILE + 16 * h;
    if constexpr (DROP) {        // this item's masks: published by stem.0's ring barriers, first read in its epilogue (the previous item's last reader lies before two barriers)
      fill_mask_tab<C::G, C::COUT, true, NTHR>(mtab, dr.d[0], s, img0, a.B, tid);
      fill_mask_tab<C::G, C::COUT, true, NTHR>(mtab + MTB, dr.d[1], s, img0, a.B, tid);
    }
    // ---- stem.0: M over the X tile, then T over it
    ring_mfma_dense<D, PD, 0>(tl, vm, zl, sx, bp.a.z_w, mblk, rg, A, nblk, lane, issue, [] {});
    QBNN_STAMP_AT(0);
    lds_barrier();                                       // every wave has read its last X fragment
    QBNN_STAMP_AT(1);
    {
      // stem.0 epilogue: window sums from the X table; the T table collects the channel sums of what is written
      auto make_epi_a = [&]() {
        if constexpr (DROP) return EpiDenseTileDrop<C::PIXB, C::COUT, IMG_PX, true>{xt, bp.a, dr.d[0], {mtab, dr.d[0].mq1}, 0};
        else return EpiDenseTile<C::PIXB>{xt, bp.a, 0};
      };
      auto epi = make_epi_a();
      auto flush = [&](int mb) {
        const int v = half_sum(epi.csum);
        epi.csum = 0;
        if (lane < 32) __hip_atomic_fetch_add(&stab[(mblk * C::MB + mb) * 32 + lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      };
      auto no_res = [](int, int, int, int, int) { return 0u; };
      auto ahead = [&](int mb) { if (mb > 0) flush(mb - 1); };
      conv_epi_phase_with<C, decltype(epi), decltype(no_res), decltype(ahead), true>(bias_lds, bp.a, epi, A, wave, lane, no_res, ahead);
      flush(C::MB - 1);
    }
    QBNN_STAMP_AT(2);
    // ---- stem.3: M over T.  The residual (= the block input, quint8, from global memory: L2 / MALL) of the first M-tile is requested in front
    // of the conv, the second M-tile's in its last slab.  Vector-memory results return in order and every `s_waitcnt vmcnt` counts them:
    // as 12 dword loads per lane inside the M phase (block_chain_ald_kernel's form) the request queued behind the ring's slabs in flight and
    // the first M-tile's epilogue waited 4 - 8 k cycles for it; moved in front of the conv it made the ring's own waits stricter (+3 k
    // cycles) -- so a lane fetches THREE 16-byte chunks (the pixel's 96 channels of this wave's channel block = 6 chunks; the lane of k-half
    // h takes chunks 3h .. 3h + 2: the wave reads 32 x 96 contiguous bytes exactly once), the ring's waits count them (EXTRA = 3), and the
    // two halves trade dwords with v_permlane32_swap when the epilogue starts.
    const int valid_px = (a.B - img0 < C::G ? a.B - img0 : C::G) * IMG_PX;
    const uint8_t* resp = a.x + (int64_t)s * a.x_ss + (int64_t)img0 * IMG_PX * C::COUT;
    auto make_epi_b = [&]() {
      if constexpr (DROP) return EpiDenseTileResGlobalDrop<C::PIXB, C::COUT, IMG_PX, true>{xt, resp, valid_px, bp.b, bp.add, dr.d[1], {mtab + MTB, dr.d[1].mq1}};
      else return EpiDenseTileResGlobal<C::PIXB, C::COUT>{xt, resp, valid_px, bp.b, bp.add};
    };
    auto epi_b = make_epi_b();
    constexpr int RES_LOADS = 3;       // vector-memory instructions of one load_res(): the ring's waits of stem.3 count exactly these (EXTRA below)
    v4i rraw[2][RES_LOADS];
    auto load_res = [&](int mb) {
      const int m = (mblk * C::MB + mb) * 32 + r;
      const uint8_t* p = resp + (int64_t)(m < valid_px ? m : 0) * C::COUT + nblk * 96 + h * 48;      // (pixels beyond a ragged batch: results never leave the tile)
#pragma unroll
      for (int j = 0; j < RES_LOADS; ++j) rraw[mb & 1][j] = *reinterpret_cast<const v4i*>(p + 16 * j);
    };
    uint32_t resq[3][4];             // [nb][g4]: the dword (4 channels at 8 g4 + 4 h) of this lane's pixel, M-tile being requantised
    auto trade_res = [&](int mb) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {  // chunk k = bytes 16 k .. 16 k + 15 of the 96: channel tile k / 2, groups g4 = 2 (k % 2) and + 1, each as (k-half 0 dword, k-half 1 dword)
        const v4i c = rraw[mb & 1][j];
        const auto e = __builtin_amdgcn_permlane32_swap((unsigned)c.x, (unsigned)c.y, false, false);      // e[0]: chunk j, e[1]: chunk j + 3 -- this lane's half
        const auto o = __builtin_amdgcn_permlane32_swap((unsigned)c.z, (unsigned)c.w, false, false);
        resq[j / 2][2 * (j % 2)] = e[0]; resq[j / 2][2 * (j % 2) + 1] = o[0];
        resq[(j + 3) / 2][2 * ((j + 3) % 2)] = e[1]; resq[(j + 3) / 2][2 * ((j + 3) % 2) + 1] = o[1];
      }
    };
    load_res(0);
    ring_mfma_dense<D, PD, RES_LOADS>(tl, vm, zl, stab, bp.b.z_w, mblk, rg, A, nblk, lane, issue, [&] { if (C::MB > 1) load_res(1); });
    QBNN_STAMP_AT(3);
    lds_barrier();
    QBNN_STAMP_AT(4);
    {
      auto res_of = [&](int, int nb, int g4, int, int) { return resq[nb][g4]; };
      auto ahead = [&](int mb) {
#ifdef QBNN_STAMP_EB
        QBNN_STAMP_AT(4 + mb);
#endif
        trade_res(mb);
        if (mb + 1 == C::MB) fetch(next);      // the next item's input: behind the last residual request (vmcnt retires in order)
      };
      conv_epi_phase_with<C, decltype(epi_b), decltype(res_of), decltype(ahead), true>(bias_lds + C::COUT, bp.b, epi_b, A, wave, lane, res_of, ahead);
    }
#ifdef QBNN_STAMP_EB
    QBNN_STAMP_AT(6);
    lds_barrier();
#else
    QBNN_STAMP_AT(5);
    lds_barrier();
    QBNN_STAMP_AT(6);
#endif
    for (int i = tid; i < D::M; i += NTHR) stab[i] = 0;             // T table: every wave has gathered from it
    // ---- per 16-byte chunk: tile -> quint8 register, next item's input -> the same tile bytes, register -> HBM (the
    //      item's output block is contiguous).  The next input is written unconditionally (the last item rewrites
    //      itself): a prefetch left unconsumed on one path makes the compiler guard later reuses with vmcnt(0).
    // ---- read-out: tile -> registers (chunk i of the item's contiguous output block per thread: coalesced stores); barrier; next item's input
    //      over the same bytes (whole pixels per thread, see write_tile); registers -> HBM.  (One loop that reads and rewrites each cell in the
    //      per-pixel mapping needs no barrier, but its 16-byte stores land 96 / 192 bytes apart and slowed the next item's first conv by 8 %.)
    bool pooled_out = false;
    if constexpr (IMG_PX == 16 && !DROP) pooled_out = a.pool != 0;
    if (pooled_out) {
      // QBNN_BLOCK_POOL_OUT (the network's last block): the output leaves as AvgPool2d(4) of it -- one byte per (image, channel) instead of 16.
      // The tile holds q - z_o: q_pool = rne(sum / 16) + z_o (the head's formula, qbnn_misc.hip head_i8_kernel; a mean of values inside
      // [-z_o, a_hi - z_o] stays inside, so its clamps are no-ops).  One (image, 4 channels) column per thread.
      if constexpr (IMG_PX == 16) {
        constexpr int C4 = C::COUT / 4, NCOL = C::G * C4, NIT = (NCOL + NTHR - 1) / NTHR;
        int t = tid;
        asm volatile("" : "+v"(t));
        uint32_t pk[NIT];
#pragma unroll
        for (int it2 = 0; it2 < NIT; ++it2) {
          const int i = t + it2 * NTHR, g = i / C4, c4 = i - g * C4;
          pk[it2] = 0;
          if (i < NCOL) {
            int s4[4] = {0, 0, 0, 0};
#pragma unroll
            for (int p = 0; p < IMG_PX; ++p) {
              const int v = *reinterpret_cast<const int*>(xt + (g * IMG_PX + p) * D::PIXB + 4 * c4);
              s4[0] = __builtin_amdgcn_sdot4(v, 0x00000001, s4[0], false);
              s4[1] = __builtin_amdgcn_sdot4(v, 0x00000100, s4[1], false);
              s4[2] = __builtin_amdgcn_sdot4(v, 0x00010000, s4[2], false);
              s4[3] = __builtin_amdgcn_sdot4(v, 0x01000000, s4[3], false);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              int q = (int)__builtin_rintf((float)s4[j] * 0.0625f) + bp.add.z_o;
              q = q < 0 ? 0 : (q > 255 ? 255 : q);
              pk[it2] |= (uint32_t)q << (8 * j);
            }
          }
        }
        lds_barrier();
        write_tile(next);
#pragma unroll
        for (int it2 = 0; it2 < NIT; ++it2) {
          const int i = t + it2 * NTHR, g = i / C4, c4 = i - g * C4;
          if (i < NCOL && img0 + g < a.B) *reinterpret_cast<uint32_t*>(a.y + (int64_t)s * a.y_ss + (int64_t)(img0 + g) * C::COUT + 4 * c4) = pk[it2];
        }
      }
    } else {
      const uint32_t z4o = (uint32_t)bp.add.z_o * 0x01010101u;
      uint8_t* ys = a.y + (int64_t)s * a.y_ss + (int64_t)img0 * IMG_PX * C::COUT;
      const int valid = valid_px * CPP;
      int t = tid;
      asm volatile("" : "+v"(t));
      v4i outv[CPT];
#pragma unroll
      for (int j = 0; j < CPT; ++j) {
        const int i = t + j * NTHR, px = i / CPP, within = i - px * CPP;
        outv[j] = *reinterpret_cast<const v4i*>(xt + px * D::PIXB + within * 16);
      }
      lds_barrier();
      write_tile(next);                // unconditional (the last item rewrites its own input): an unconsumed prefetch costs a vmcnt(0) guard
#pragma unroll
      for (int j = 0; j < CPT; ++j) {
        const int i = t + j * NTHR;
        const v4i v = outv[j];
        if (i < valid)
          *reinterpret_cast<v4i*>(ys + (int64_t)i * 16) = v4i{(int)add_bytes(v.x, z4o), (int)add_bytes(v.y, z4o), (int)add_bytes(v.z, z4o), (int)add_bytes(v.w, z4o)};
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

#ifndef QBNN_CHAIN_PD
#define QBNN_CHAIN_PD 2
#endif

// A launch variant: the blocking (ConvCfg), the ring depth and the fragment request distance.
//   V1: one workgroup per CU -- four slabs (three in flight), two pixel tiles x three channel tiles per wave where the item has them, 151 - 225 VGPRs;
//   V2 (round 5): TWO workgroups per CU -- half the images per item, one pixel tile x three channel tiles per wave, a two-slab ring: 80 KiB of
//       LDS and <= 128 VGPRs per workgroup.  The two workgroups of a CU run out of phase (one's epilogues under the other's K loop: MFMA cycles
//       with a vector instruction beside them 4 / 2 % -> 21 / 13 %) and every SIMD holds four waves, where a vector instruction costs 2.0 - 3.4
//       cycles instead of 2.9 - 4.8 (profiles/r05_issue_bench2.txt).  Each item streams its weights for half as many images, so the L2 -> LDS
//       traffic doubles (3.8 B/clk per CU).  Measured (same box): chain 96 0.268 -> 0.251 ms, chain 192 0.236 -> 0.223 ms; QBNN_CHAIN_2WG=0 for the A/B.
template <class C_, int NBUF_, int PD_> struct CRVar {
  using C = C_;
  static constexpr int NBUF = NBUF_, PD = PD_, WGS = NBUF_ == 2 ? 512 : 256;      // workgroups that fill the chip
  using D = CRCfg<C_, NBUF_>;
  static_assert(NBUF_ != 2 || D::LDS <= 80 * 1024, "two workgroups per CU");
};

template <class V>
int launch_by_value(const ChainArgs<1>* arr, int n, hipStream_t st) {
  using C = typename V::C;
  using D = typename V::D;
  static std::atomic<uint64_t> attr1{0}, attrN{0};
  int items = 0;
  for (int i = 0; i < n; ++i) { const int it = arr[i].n_samples * ((arr[i].B + C::G - 1) / C::G); items = it > items ? it : items; }
  if (n == 1) {
    if (int rc = ensure_dyn_lds((const void*)block_chain_ring_kernel<C, V::PD, 1, false, V::NBUF>, attr1, D::LDS)) return rc;
    ArgsArr<ChainArgs<1>, 1> one;
    one.m[0] = arr[0];
    hipLaunchKernelGGL((block_chain_ring_kernel<C, V::PD, 1, false, V::NBUF>), dim3(items < V::WGS ? (items > 0 ? items : 1) : V::WGS), dim3(512), D::LDS, st, one, DropSet<0>{});
    return check_launch("qbnn_block_chain_i8_mc");
  }
  static_assert(sizeof(ArgsArr<ChainArgs<1>, QBNN_FUSED_CALLS>) <= 3840, "kernel arguments are limited to 4 KiB (incl. the hidden ones)");
  if (int rc = ensure_dyn_lds((const void*)block_chain_ring_kernel<C, V::PD, QBNN_FUSED_CALLS, false, V::NBUF>, attrN, D::LDS)) return rc;
  ArgsArr<ChainArgs<1>, QBNN_FUSED_CALLS> all;
  memset(&all, 0, sizeof(all));
  for (int i = 0; i < n; ++i) all.m[i] = arr[i];
  const int per = V::WGS / n > 0 ? V::WGS / n : 1;
  hipLaunchKernelGGL((block_chain_ring_kernel<C, V::PD, QBNN_FUSED_CALLS, false, V::NBUF>), dim3(items < per ? (items > 0 ? items : 1) : per, n), dim3(512), D::LDS, st, all, DropSet<0>{});
  return check_launch("qbnn_block_chain_i8_multi");
}

template <class V>
int launch_dev(const ChainArgs<1>* dev, int n, int B, int max_samples, hipStream_t st) {
  using C = typename V::C;
  using D = typename V::D;
  const int items = max_samples * ((B + C::G - 1) / C::G);
  static std::atomic<uint64_t> attr{0};
  if (int rc = ensure_dyn_lds((const void*)block_chain_ring_kernel<C, V::PD, 0, false, V::NBUF>, attr, D::LDS)) return rc;
  const int per = V::WGS / n > 0 ? V::WGS / n : 1;
  hipLaunchKernelGGL((block_chain_ring_kernel<C, V::PD, 0, false, V::NBUF>), dim3(items < per ? (items > 0 ? items : 1) : per, n), dim3(512), D::LDS, st, ArgsArr<ChainArgs<1>, 0>{dev}, DropSet<0>{});
  return check_launch("qbnn_block_chain_i8_multi_launch");
}

template <class V>
int launch_drop(const ChainArgs<1>& a, const DropSet<2>& dr, hipStream_t st) {
  using C = typename V::C;
  using D = typename V::D;
  constexpr int LDS = D::LDS + 2 * MaskTab<C::COUT, true>::bytes(C::G);
  static_assert(LDS <= (V::NBUF == 2 ? 80 : 160) * 1024, "LDS budget");
  static std::atomic<uint64_t> attr{0};
  if (int rc = ensure_dyn_lds((const void*)block_chain_ring_kernel<C, V::PD, 1, true, V::NBUF>, attr, LDS)) return rc;
  const int items = a.n_samples * ((a.B + C::G - 1) / C::G);
  ArgsArr<ChainArgs<1>, 1> one;
  one.m[0] = a;
  hipLaunchKernelGGL((block_chain_ring_kernel<C, V::PD, 1, true, V::NBUF>), dim3(items < V::WGS ? (items > 0 ? items : 1) : V::WGS), dim3(512), LDS, st, one, dr);
  return check_launch("qbnn_block_chain_drop_i8_mc");
}

using CR_96  = ConvCfg<96, 96, 3, 1, 8, 1, 8, 2, 3, true, 36, 16>;        // 8 images per item: 16 pixel tiles x 3 channel tiles
using CR_192 = ConvCfg<192, 192, 3, 1, 4, 1, 16, 2, 3, true, 36, 16>;     // 16 images per item: 8 x 6
using CR_192_G8 = ConvCfg<192, 192, 3, 1, 4, 1, 8, 1, 3, true, 36, 16>;   // 8 images per item: 4 x 6
using CR_96_G4 = ConvCfg<96, 96, 3, 1, 8, 1, 4, 1, 3, true, 36, 16>;      // 4 images per item: 8 x 3
using V1_96 = CRVar<CR_96, 4, QBNN_CHAIN_PD>;
using V1_192 = CRVar<CR_192, 4, QBNN_CHAIN_PD>;
using V1_192_G8 = CRVar<CR_192_G8, 4, QBNN_CHAIN_PD>;       // (ensemble members at B <= 256 with QBNN_CHAIN_2WG=0)
using V2_96 = CRVar<CR_96_G4, 2, 2>;                         // 128 VGPRs, no spill
using V2_192 = CRVar<CR_192_G8, 2, 1>;                       // 121 VGPRs (request distance 2: two spills, the same time)
using V2_96_DROP = CRVar<CR_96_G4, 2, 1>;                    // (the dropout form spills at distance 2)

bool two_per_cu() {
  static const bool on = [] { const char* e = getenv("QBNN_CHAIN_2WG"); return !(e && e[0] == '0'); }();
  return on;
}

}  // namespace

// `small_items`: the 192-channel block with 8 images per item (the caller's choice when 16-image items would leave CUs idle; V1 only -- V2's items are that size)
int qbnn_launch_block_chain_ring(const ChainArgs<1>* arr, int n, int Cc, bool small_items, hipStream_t st) {
  if (n <= 0 || n > QBNN_FUSED_CALLS) return fail(QBNN_E_INVALID, "qbnn_block_chain (ring): 1 .. 8 argument blocks per launch%s");
  const bool two = two_per_cu() && n == 1;       // side-by-side ensemble members: V1 (measured: 27.5 k against 27.2 k member forwards/s)
  if (Cc == 96) return two ? launch_by_value<V2_96>(arr, n, st) : launch_by_value<V1_96>(arr, n, st);
  if (Cc == 192) return two ? launch_by_value<V2_192>(arr, n, st) : small_items ? launch_by_value<V1_192_G8>(arr, n, st) : launch_by_value<V1_192>(arr, n, st);
  return fail(QBNN_E_INVALID, "qbnn_block_chain (ring): 96 and 192 channels only%s");
}

// (device-resident argument blocks: B and the largest sample count of the calls size the grid)
int qbnn_launch_block_chain_ring_dev(const ChainArgs<1>* dev, int n, int B, int max_samples, int Cc, bool small_items, hipStream_t st) {
  const bool two = two_per_cu() && n == 1;
  if (Cc == 96) return two ? launch_dev<V2_96>(dev, n, B, max_samples, st) : launch_dev<V1_96>(dev, n, B, max_samples, st);
  if (Cc == 192) return two ? launch_dev<V2_192>(dev, n, B, max_samples, st) : small_items ? launch_dev<V1_192_G8>(dev, n, B, max_samples, st) : launch_dev<V1_192>(dev, n, B, max_samples, st);
  return fail(QBNN_E_INVALID, "qbnn_block_chain (ring): 96 and 192 channels only%s");
}

int qbnn_launch_block_chain_ring_drop(const ChainArgs<1>& a, const DropSet<2>& dr, int Cc, hipStream_t st) {
  if (Cc == 96) return two_per_cu() ? launch_drop<V2_96_DROP>(a, dr, st) : launch_drop<V1_96>(a, dr, st);
  if (Cc == 192) return two_per_cu() ? launch_drop<V2_192>(a, dr, st) : launch_drop<V1_192>(a, dr, st);
  return fail(QBNN_E_INVALID, "qbnn_block_chain_drop (ring): 96 and 192 channels only%s");
}
