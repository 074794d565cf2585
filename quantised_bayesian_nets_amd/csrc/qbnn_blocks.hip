// libqbnn_hip.so -- fused BasicBlock kernels of the int8 path (persistent workgroups, activations stay in LDS between a
// block's convs) and their C ABI: qbnn_block_chain_i8_mc / qbnn_stem_chain_i8_mc / qbnn_block_down_i8_mc and the
// multi-call (ensemble) forms.  Shared device code: qbnn_conv.h.
#include "qbnn_host.h"
#include <mutex>
#include <unordered_map>
#include <vector>

#ifdef QBNN_STAMP
static unsigned long long* g_stamp_buf = nullptr;
QBNN_EXPORT void qbnn_debug_stamp_buffer(void* p) { g_stamp_buf = (unsigned long long*)p; hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_dev_ptr), &p, sizeof(p)); }
QBNN_EXPORT void qbnn_debug_read_inner(unsigned long long* host4) {
  hipMemcpyFromSymbol(host4, HIP_SYMBOL(g_inner), 32);
  unsigned long long z[4] = {0, 0, 0, 0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_inner), z, 32);
}
#endif

// =====================================================================================
// Fused down-sampling BasicBlock (models_bbb.py:146-183 with stride 2): shortcut 1x1/s2 conv, stem.0 3x3/s2 ConvReLU,
// stem.3 3x3 conv, Add, ReLU in one persistent kernel.
//   X tile (Cin, HIN): centred block input      --conv_s-->  SC: dense quint8 [M][COUT] (the residual operand)
//                                               --conv_a-->  T tile (COUT, HO): centred stem.0 output
//   T --conv_b--> + SC --> SC in place (block output, quint8) --> HBM
// =====================================================================================
// (DownArgs: qbnn_conv.h -- shared with the ring form of the wide blocks, qbnn_down_ring.hip)

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
using D96_a = ConvCfg<96, 192, 3, 2, 8, 1, 8, 1, 3, false>;

static int build_down_args(DownArgs& a, const uint8_t* x, int64_t x_ss, float s_x, int32_t z_x, int32_t B, int32_t a_hi, const qbnn_down_desc* d,
                           uint8_t* y, int64_t y_ss, int32_t n_samples, const qbnn_drop_desc* drops = nullptr) {
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
  if (drops) { c.s_x = drops[0].s_out; c.z_x = drops[0].z_m; }      // the second conv reads the dropped stem.0 output
  if ((rc = fill_qconv(a.b, d->blk.w_b, d->blk.w_b_sample_stride, d->blk.bias_b, &c))) return rc;
  c.s_r = d->s_s; c.z_r = d->z_s; c.s_o = d->blk.s_o; c.z_o = d->blk.z_o;
  if (drops) { c.s_r = drops[2].s_out; c.z_r = drops[2].z_m; }      // the Add's second operand: the dropped shortcut
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
    const int lay = calls[c0].desc->blk.w_layout;
    for (int i = 0; i < n; ++i)
      if (calls[c0 + i].desc->blk.w_layout != lay) return fail(QBNN_E_INVALID, "qbnn_block_down_i8_multi: one weight layout per call array%s");
    if (lay != QBNN_LAYOUT_MFMA32 && !(lay == QBNN_LAYOUT_MFMA32_N24 && Cin == 24 && H == 32))
      return fail(QBNN_E_INVALID, "qbnn_block_down_i8_multi: MFMA32 weights (MFMA32_N24 set: the 24 -> 48 block)%s");
    if (Cin == 24 && H == 32) rc = lay == QBNN_LAYOUT_MFMA32_N24 ? qbnn_launch_down24_w16(arr, n, st) : launch_block_down_ws_multi<D24_a, D24_s, D24_b, true>(arr, n, st);
    else if (Cin == 48 && H == 16) rc = qbnn_launch_block_down_ring(arr, n, 48, st);
    else if (Cin == 96 && H == 8) rc = qbnn_launch_block_down_ring(arr, n, 96, st);
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
  if (d->blk.w_layout == QBNN_LAYOUT_MFMA32_N24) {       // the 16-wave kernel (qbnn_c48.hip)
    if (Cin == 24 && H == 32) return qbnn_launch_down24_w16(&a, 1, st);
    return fail(QBNN_E_INVALID, "qbnn_block_down_i8_mc: the MFMA32_N24 layout set serves the 24 -> 48 block%s");
  }
  if (d->blk.w_layout != QBNN_LAYOUT_MFMA32) return fail(QBNN_E_INVALID, "qbnn_block_down_i8_mc: unknown weight layout%s");
  // (a ping-pong variant of this block -- phases W / M_a / E_sa / M_b / E_b on two 4-wave groups -- measured 15 % SLOWER
  //  than the weights-stationary kernel: five barrier intervals per image, each as long as the slower group's phase)
  if (Cin == 24 && H == 32) return launch_block_down_ws<D24_a, D24_s, D24_b, true>(a, st);
  // 48 -> 96 and 96 -> 192: the block's weights through the LDS slab ring (qbnn_down_ring.hip, round 4); QBNN_DOWN_RING=0: per-wave L2 streaming
  if (Cin == 48 && H == 16) return qbnn_launch_block_down_ring(&a, 1, 48, st);
  if (Cin == 96 && H == 8) return qbnn_launch_block_down_ring(&a, 1, 96, st);
  return fail(QBNN_E_INVALID, "qbnn_block_down_i8_mc: unsupported geometry%s Cin=%ld H=%ld", "", Cin, H);
}

// LDSW = true : weights-stationary as described above (contiguous item ranges).
// LDSW = false: the block's weights are too large for LDS -- every wave streams its fragments from L2 (conv_passes) and
//               the workgroups walk the items interleaved per XCD (ItemWalk), so that the workgroups sharing an L2
//               work on the same MC sample at a time and its weights stay hot there.  Same barrier / prefetch scheme.
// STEM = true (layer 1 only): the network's first conv (3 -> 24 channels, on the pre-gathered 27-tap patches) runs inside
//               the same kernel -- its output never goes to HBM (that tensor is the largest of the network: 629 MB per
//               100-sample step written and read back).  The item's input is then its image's patch block (32 KiB,
//               shared by all samples, L2-resident), staged in a dense LDS tile; conv0's epilogue writes the X tile.
// DROP (conv_resnet_mc: mcdropout/models_mc.py:116-160): a quantised channel dropout behind every conv (dr.d = [layers.3 when STEM],
//               then per block stem.3, stem.6), applied in the convs' epilogues from per-item mask tables in LDS.
template <int NBLK, bool STEM, bool DROP> constexpr int chain_ndrop() { return DROP ? 2 * NBLK + (STEM ? 1 : 0) : 0; }

template <class C, int NBLK, bool LDSW, bool STEM = false, int NM = 1, int NTHR_ = BLK_THREADS, bool DROP = false>
__global__ __launch_bounds__(NTHR_) void block_chain_ws_kernel(const ArgsArr<ChainArgs<NBLK>, NM> all, const DropSet<chain_ndrop<NBLK, STEM, DROP>()> dr) {
  const ChainArgs<NBLK> a = args_of(all, blockIdx.y);
  static_assert(!DROP || NM == 1, "dropout variants are single-call");
  using MT = MaskTab<C::COUT, false>;
  constexpr int MTB = MT::bytes(C::G), D0 = STEM ? 1 : 0;
  using C0 = ConvCfg<32, 24, 1, 1, 32, 0, 1, 4, 1>;      // layer 0 on the patch tensor: K = 27 -> 32, one k-step
  static_assert(!STEM || (LDSW && C::CIN == 24 && C::HIN == 32 && C::G == 1), "the fused stem feeds the 32x32x24 chain");
  static_assert(C::CIN == C::COUT && C::STRIDE == 1 && C::KSZ == 3 && C::HALO == 1, "identity BasicBlock geometry");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr int NTHR = NTHR_, NWV = NTHR_ / 64;
  constexpr int TILES = C::G * C::TILE_BYTES + C::TILE_SLACK;
  constexpr int WB = LDSW ? WConv<C>::BYTES : 0;
  uint8_t* xt = smem;
  uint8_t* tt = smem + TILES;
  uint8_t* wl = smem + 2 * TILES;                                            // [NBLK][2] whole convs
  float* bias_lds = reinterpret_cast<float*>(wl + 2 * NBLK * WB);            // [NBLK][2][COUT]
  uint8_t* im = reinterpret_cast<uint8_t*>(bias_lds + NBLK * 2 * C::COUT);   // STEM: patch tile [1024][32], stem weights, stem bias
  uint8_t* wl0 = im + C0::TILE_BYTES;
  float* bias0 = reinterpret_cast<float*>(wl0 + WConv<C0>::BYTES);
  uint8_t* mtab = STEM ? reinterpret_cast<uint8_t*>(bias0 + C0::COUT) : im;  // DROP: mask tables [dropout][G][COUT]
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
    if constexpr (DROP) {        // this item's masks (every wave has left the previous item's last epilogue: the barrier before the read-out)
#pragma unroll
      for (int d = 0; d < chain_ndrop<NBLK, STEM, DROP>(); ++d) fill_mask_tab<C::G, C::COUT, false, NTHR>(mtab + d * MTB, dr.d[d], s, img0, a.B, tid);
    }
    lds_barrier();
    QBNN_STAMP_AT(1);
    if constexpr (STEM) {        // layers.0 (ConvReLU2d): patch tile -> X tile, centred on its own zero point (= a.z_in)
      if constexpr (DROP) {
        EpiTileDrop<C::HO, C::PIXB, C::TILE_BYTES, C::COUT> epi{xt, a.stem, dr.d[0], {mtab, 0.f}};
        conv_core<C0, decltype(epi), NWV>(im, wl0, bias0, a.stem, epi, wave, lane);
      } else {
        EpiTile<C::HO, C::PIXB, C::TILE_BYTES> epi{xt, a.stem};
        conv_core<C0, decltype(epi), NWV>(im, wl0, bias0, a.stem, epi, wave, lane);
      }
      lds_barrier();
    }
#pragma unroll
    for (int k = 0; k < NBLK; ++k) {
      const BlockParams& bp = a.blk[k];
      auto conv_a = [&](auto& epi) {
        if constexpr (LDSW) conv_core<C, std::remove_reference_t<decltype(epi)>, NWV>(xt, wl + (2 * k) * WB, bias_lds + (2 * k) * C::COUT, bp.a, epi, wave, lane);
        else conv_passes<C, std::remove_reference_t<decltype(epi)>, NWV>(xt, bp.a.w + (int64_t)s * bp.a.w_ss, bias_lds + (2 * k) * C::COUT, bp.a, epi, wave, lane);
      };
      auto conv_b = [&](auto& epi) {
        if constexpr (LDSW) conv_core<C, std::remove_reference_t<decltype(epi)>, NWV>(tt, wl + (2 * k + 1) * WB, bias_lds + (2 * k + 1) * C::COUT, bp.b, epi, wave, lane);
        else conv_passes<C, std::remove_reference_t<decltype(epi)>, NWV>(tt, bp.b.w + (int64_t)s * bp.b.w_ss, bias_lds + (2 * k + 1) * C::COUT, bp.b, epi, wave, lane);
      };
      if constexpr (DROP) {
        EpiTileDrop<C::HO, C::PIXB, C::TILE_BYTES, C::COUT> epi{tt, bp.a, dr.d[D0 + 2 * k], {mtab + (D0 + 2 * k) * MTB, 0.f}};
        conv_a(epi);
      } else {
        EpiTile<C::HO, C::PIXB, C::TILE_BYTES> epi{tt, bp.a};
        conv_a(epi);
      }
      QBNN_STAMP_AT(2);
      lds_barrier();
      QBNN_STAMP_AT(3);
      if constexpr (DROP) {
        EpiTileResInPlaceDrop<C::HO, C::PIXB, C::TILE_BYTES, C::COUT> epi{xt, bp.b, bp.add, dr.d[D0 + 2 * k + 1], {mtab + (D0 + 2 * k + 1) * MTB, 0.f}};
        conv_b(epi);
      } else {
        EpiTileResInPlace<C::HO, C::PIXB, C::TILE_BYTES> epi{xt, bp.b, bp.add};
        conv_b(epi);
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

// (Round 3's two-slab form of the 96 / 192-channel identity block -- block_chain_ald_kernel, the QBNN_CHAIN_RING=0 path of rounds 4 - 5 -- left the build in
//  round 6: tools/experiments/r03_block_chain_ald_kernel.hip.txt.  Those blocks run on csrc/qbnn_chain_ring.hip.)

// grid of a fused multi-call launch: every call gets the same number of workgroups (<= its item count), 256 in total
static int fused_grid_x(int max_items, int n_calls) {
  const int per = 256 / n_calls > 0 ? 256 / n_calls : 1;
  return max_items < per ? (max_items > 0 ? max_items : 1) : per;
}


template <class C, int NBLK, bool LDSW = true, bool STEM = false, bool DROP = false> constexpr int chain_ws_lds() {
  return 2 * (C::G * C::TILE_BYTES + C::TILE_SLACK) + (LDSW ? 2 * NBLK * WConv<C>::BYTES : 0) + NBLK * 2 * C::COUT * 4 +
         (STEM ? 32 * 32 * 32 + 1024 + 24 * 4 : 0) + chain_ndrop<NBLK, STEM, DROP>() * MaskTab<C::COUT, false>::bytes(C::G);
}

template <class C, int NBLK, bool LDSW = true, bool STEM = false, int NTHR = BLK_THREADS>
static int launch_block_chain_ws(const ChainArgs<NBLK>& a, hipStream_t st) {
  constexpr int LDS = chain_ws_lds<C, NBLK, LDSW, STEM>();
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static std::atomic<uint64_t> attr{0};
  if (int rc_attr = ensure_dyn_lds((const void*)block_chain_ws_kernel<C, NBLK, LDSW, STEM, 1, NTHR>, attr, LDS)) return rc_attr;
  const int groups = (a.B + C::G - 1) / C::G;
  const int n_items = a.n_samples * groups;
  const int grid = n_items < 256 ? n_items : 256;
  ArgsArr<ChainArgs<NBLK>, 1> one;
  one.m[0] = a;
  hipLaunchKernelGGL((block_chain_ws_kernel<C, NBLK, LDSW, STEM, 1, NTHR>), dim3(grid), dim3(NTHR), LDS, st, one, DropSet<0>{});
  return check_launch("qbnn_block_chain_i8_mc");
}

template <class C, int NBLK, bool STEM>
static int launch_block_chain_ws_drop(const ChainArgs<NBLK>& a, const DropSet<chain_ndrop<NBLK, STEM, true>()>& dr, hipStream_t st) {
  constexpr int LDS = chain_ws_lds<C, NBLK, true, STEM, true>();
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static std::atomic<uint64_t> attr{0};
  if (int rc_attr = ensure_dyn_lds((const void*)block_chain_ws_kernel<C, NBLK, true, STEM, 1, BLK_THREADS, true>, attr, LDS)) return rc_attr;
  const int groups = (a.B + C::G - 1) / C::G;
  const int n_items = a.n_samples * groups;
  const int grid = n_items < 256 ? n_items : 256;
  ArgsArr<ChainArgs<NBLK>, 1> one;
  one.m[0] = a;
  hipLaunchKernelGGL((block_chain_ws_kernel<C, NBLK, true, STEM, 1, BLK_THREADS, true>), dim3(grid), dim3(BLK_THREADS), LDS, st, one, dr);
  return check_launch("qbnn_block_chain_drop_i8_mc");
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
  hipLaunchKernelGGL((block_chain_ws_kernel<C, NBLK, true, STEM, NM>), dim3(fused_grid_x(items, n), n), dim3(BLK_THREADS), LDS, st, all, DropSet<0>{});
  return check_launch("qbnn_block_chain_i8_multi");
}

template <class CB> struct DownSC {
  static constexpr int PITCH = CB::COUT + 8;
  static constexpr int BYTES = (CB::M * PITCH + 15) / 16 * 16;
};

// DROP (conv_resnet_mc): dr.d = stem.3 (behind conv_a), stem.6 (behind conv_b; the Add's first operand), shortcut.2 (behind conv_s;
// the Add's second operand).  MBITS: bit tables where LDS is short (96 -> 192).
template <class CA, class CS, class CB, bool LDSW, int NM = 1, bool DROP = false, bool MBITS = false>
__global__ __launch_bounds__(BLK_THREADS) void block_down_ws_kernel(const ArgsArr<DownArgs, NM> all, const DropSet<DROP ? 3 : 0> dr) {
  const DownArgs a = args_of(all, blockIdx.y);
  static_assert(!DROP || NM == 1, "dropout variants are single-call");
  using MT = MaskTab<CB::COUT, MBITS>;
  constexpr int MTB = MT::bytes(CB::G);
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
  uint8_t* mtab = reinterpret_cast<uint8_t*>(bias_lds + 3 * COUT);                        // DROP: three mask tables
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
    if constexpr (DROP) {
#pragma unroll
      for (int d = 0; d < 3; ++d) fill_mask_tab<CB::G, COUT, MBITS, BLK_THREADS>(mtab + d * MTB, dr.d[d], s, img0, a.B, tid);
    }
    lds_barrier();       // X complete; the previous item's SC has been read out by every thread
    QBNN_STAMP_AT(1);
    auto conv_s = [&](auto& epi) {
      if constexpr (LDSW) conv_core<CS, std::remove_reference_t<decltype(epi)>, BLK_WAVES>(xt, wl_s, bias_lds, a.s, epi, wave, lane);
      else conv_passes<CS, std::remove_reference_t<decltype(epi)>, BLK_WAVES>(xt, a.s.w + (int64_t)s * a.s.w_ss, bias_lds, a.s, epi, wave, lane);
    };
    auto conv_a = [&](auto& epi) {
      if constexpr (LDSW) conv_core<CA, std::remove_reference_t<decltype(epi)>, BLK_WAVES>(xt, wl_a, bias_lds + COUT, a.a, epi, wave, lane);
      else conv_passes<CA, std::remove_reference_t<decltype(epi)>, BLK_WAVES>(xt, a.a.w + (int64_t)s * a.a.w_ss, bias_lds + COUT, a.a, epi, wave, lane);
    };
    auto conv_b = [&](auto& epi) {
      if constexpr (LDSW) conv_core<CB, std::remove_reference_t<decltype(epi)>, BLK_WAVES>(tt, wl_b, bias_lds + 2 * COUT, a.b, epi, wave, lane);
      else conv_passes<CB, std::remove_reference_t<decltype(epi)>, BLK_WAVES>(tt, a.b.w + (int64_t)s * a.b.w_ss, bias_lds + 2 * COUT, a.b, epi, wave, lane);
    };
    constexpr int IMG_PIX = CB::HO * CB::HO;
    if constexpr (DROP) {
      EpiDenseDrop<COUT, IMG_PIX, false, SCP, MBITS> epi_s{sc, a.s, a.add, dr.d[2], {mtab + 2 * MTB, dr.d[2].mq1}};
      conv_s(epi_s);
      EpiTileDrop<CB::HIN, CB::PIXB, CB::TILE_BYTES, COUT, MBITS> epi_a{tt, a.a, dr.d[0], {mtab, dr.d[0].mq1}};
      conv_a(epi_a);
    } else {
      EpiDense<COUT, false, SCP> epi_s{sc, a.s, a.add};
      conv_s(epi_s);
      EpiTile<CB::HIN, CB::PIXB, CB::TILE_BYTES> epi_a{tt, a.a};
      conv_a(epi_a);
    }
    QBNN_STAMP_AT(2);
    lds_barrier();       // T and SC complete
    QBNN_STAMP_AT(3);
    if constexpr (DROP) {
      EpiDenseDrop<COUT, IMG_PIX, true, SCP, MBITS> epi_b{sc, a.b, a.add, dr.d[1], {mtab + MTB, dr.d[1].mq1}};
      conv_b(epi_b);
    } else {
      EpiDense<COUT, true, SCP> epi_b{sc, a.b, a.add};
      conv_b(epi_b);
    }
    QBNN_STAMP_AT(4);
    lds_barrier();
    QBNN_STAMP_AT(5);
    // (Round 3: conv_b's epilogue writing its dwords straight to HBM instead -- what the 16-wave layer-1 kernel does -- is SLOWER here:
    //  0.493 / 0.42 / 0.36 ms against 0.46 / 0.40 / 0.36 for the three blocks; the staged, coalesced read-out stays.)
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
  hipLaunchKernelGGL((block_down_ws_kernel<CA, CS, CB, LDSW, 1>), dim3(grid), dim3(BLK_THREADS), LDS, st, one, DropSet<0>{});
  return check_launch("qbnn_block_down_i8_mc");
}

template <class CA, class CS, class CB, bool LDSW, bool MBITS>
static int launch_block_down_ws_drop(const DownArgs& a, const DropSet<3>& dr, hipStream_t st) {
  constexpr int LDS = CA::G * CA::TILE_BYTES + CA::TILE_SLACK + CB::G * CB::TILE_BYTES + CB::TILE_SLACK + DownSC<CB>::BYTES +
                      (LDSW ? WConv<CS>::BYTES + WConv<CA>::BYTES + WConv<CB>::BYTES : 0) + 3 * CB::COUT * 4 + 3 * MaskTab<CB::COUT, MBITS>::bytes(CB::G);
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static std::atomic<uint64_t> attr{0};
  if (int rc_attr = ensure_dyn_lds((const void*)block_down_ws_kernel<CA, CS, CB, LDSW, 1, true, MBITS>, attr, LDS)) return rc_attr;
  const int groups = (a.B + CA::G - 1) / CA::G;
  const int n_items = a.n_samples * groups;
  const int grid = n_items < 256 ? n_items : 256;
  ArgsArr<DownArgs, 1> one;
  one.m[0] = a;
  hipLaunchKernelGGL((block_down_ws_kernel<CA, CS, CB, LDSW, 1, true, MBITS>), dim3(grid), dim3(BLK_THREADS), LDS, st, one, dr);
  return check_launch("qbnn_block_down_drop_i8_mc");
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
  hipLaunchKernelGGL((block_down_ws_kernel<CA, CS, CB, LDSW, QBNN_FUSED_CALLS>), dim3(fused_grid_x(items, n), n), dim3(BLK_THREADS), LDS, st, all, DropSet<0>{});
  return check_launch("qbnn_block_down_i8_multi");
}

//                          CIN COUT K  S  HIN HALO G  MB NB
using Blk_24  = ConvCfg<24, 24, 3, 1, 32, 1, 1, 4, 1>;
using Blk_48  = ConvCfg<48, 48, 3, 1, 16, 1, 2, 2, 2>;
using PP_48   = ConvCfg<48, 48, 3, 1, 16, 1, 1, 2, 2>;          // per wave group of the ping-pong kernel
// (A 16-wave instantiation of the 48-channel chain -- ConvCfg<48, 48, 3, 1, 16, 1, 2, 1, 2> at 1024 threads, 88 VGPRs, one pass per
//  wave and conv -- takes the same 0.389 ms as the 8-wave kernels (0.391): occupancy alone does not buy the overlap, round 3.)

// PostArgs of one dropout behind a conv whose output scale is s_conv (qbnn_conv2d_i8_post_mc's arithmetic; s_a / dl_a: the dropped
// tensor as the first operand of the block's Add)
static int fill_drop(PostArgs& o, const qbnn_drop_desc& q, float s_conv, int a_hi, uint64_t seed, uint32_t sample_begin) {
  if (q.z_m < 0 || q.z_m > 127 || !(q.s_m > 0.f) || !(q.s_out > 0.f)) return fail(QBNN_E_INVALID, "qbnn dropout: mask zero point must be in [0,127], scales positive%s");
  memset(&o, 0, sizeof(o));
  const int hi = a_hi < 255 ? a_hi : 255;
  o.keep = q.keep_prob; o.inv_sm = 1.0f / q.s_m; o.z_m = q.z_m;
  o.dmult = (float)((double)s_conv * (double)q.s_m / (double)q.s_m);       // ATen qmul: self_scale * other_scale / out_scale
  o.dlo = (float)(-q.z_m); o.dhi = (float)(hi - q.z_m);
  o.seed_lo = (uint32_t)seed; o.seed_hi = (uint32_t)(seed >> 32); o.layer_id = q.layer_id; o.sample_begin = sample_begin;
  o.mask_in = q.mask_in; o.nd = g_noise_dev;
  o.s_a = q.s_out;
  o.dl_a = fmaf(q.s_out, (float)q.z_m, (float)(-q.z_m) * q.s_out);
  float t = nearbyintf(1.0f * o.inv_sm);                                   // a kept element: quantize_per_tensor(1.0, s_m, z_m) - z_m
  t = t > 2147483520.f ? 2147483520.f : t;
  int m1 = q.z_m + (int)t;
  m1 = m1 < 0 ? 0 : (m1 > 255 ? 255 : m1);
  o.mq1 = (float)(m1 - q.z_m);
  return QBNN_OK;
}

template <int NBLK>
static int build_chain_args(ChainArgs<NBLK>& a, const uint8_t* x, int64_t x_ss, float s_x, int32_t z_x, int32_t B, int32_t a_hi,
                            const qbnn_block_desc* blk, uint8_t* y, int64_t y_ss, int32_t n_samples, const int8_t* stem_x, const QConv* stem,
                            const qbnn_drop_desc* drops = nullptr, bool pool_ok = false) {
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
    if (drops) { d.s_x = drops[2 * k].s_out; d.z_x = drops[2 * k].z_m; }      // the second conv reads the dropped stem.0 output
    if ((rc = fill_qconv(a.blk[k].b, b.w_b, b.w_b_sample_stride, b.bias_b, &d))) return rc;
    d.s_r = s_in; d.z_r = z_in; d.s_o = b.s_o; d.z_o = b.z_o;
    if ((rc = fill_qadd(a.blk[k].add, &d))) return rc;
    s_in = b.s_o; z_in = b.z_o;
    if ((b.flags & QBNN_BLOCK_POOL_OUT) && k != NBLK - 1) return fail(QBNN_E_INVALID, "qbnn_block_chain: QBNN_BLOCK_POOL_OUT belongs to the chain's last block%s");
  }
  a.pool = (blk[NBLK - 1].flags & QBNN_BLOCK_POOL_OUT) ? 1 : 0;
  // Only qbnn_block_chain_i8_mc's 4 x 4 x 192 single-block launch on the ring kernel writes the pooled [S][B][C] tensor (pool_ok); the multi-call,
  // prepared and dropout entry points run kernels that would write the full H x W x C map into a y the caller sized for the pooled one.
  if (a.pool && !pool_ok)
    return fail(QBNN_E_INVALID, "QBNN_BLOCK_POOL_OUT is served by qbnn_block_chain_i8_mc's 4x4x192 identity block only (one block per launch, ring kernel, MFMA32 weights)%s");
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
  const bool pool_ok = Cc == 192 && H == 4 && NBLK == 1 && !stem && blk[0].w_layout == QBNN_LAYOUT_MFMA32;
  if (int rc = build_chain_args<NBLK>(a, x, x_ss, s_x, z_x, B, a_hi, blk, y, y_ss, n_samples, stem_x, stem, nullptr, pool_ok)) return rc;
  if (stem) {
    if (Cc != 24 || H != 32) return fail(QBNN_E_INVALID, "qbnn_stem_chain_i8_mc: the fused stem feeds the 32x32x24 chain only%s");
    for (int k = 0; k < NBLK; ++k)
      if (blk[k].w_layout != blk[0].w_layout) return fail(QBNN_E_INVALID, "qbnn_stem_chain_i8_mc: one packed weight layout per launch%s");
    // the blocks' weights as QBNN_LAYOUT_MFMA32_TAIL fragments (7 k-steps): the 16-wave kernel; as MFMA32 (9 k-steps): the 8-wave kernel
    if (blk[0].w_layout == QBNN_LAYOUT_MFMA32_TAIL) {
      if constexpr (NBLK == 2) return qbnn_launch_stem_chain_w16(&a, 1, a_hi, st);
      else return fail(QBNN_E_INVALID, "qbnn_stem_chain_i8_mc: the MFMA32_TAIL layout serves layers.0 + two blocks (the 16-wave kernel)%s");
    }
    if (blk[0].w_layout != QBNN_LAYOUT_MFMA32) return fail(QBNN_E_INVALID, "qbnn_stem_chain_i8_mc: the 24-channel blocks take MFMA32 or MFMA32_TAIL weights%s");
    return launch_block_chain_ws<Blk_24, NBLK, true, true>(a, st);
  }
  for (int k = 0; k < NBLK; ++k)
    if (blk[k].w_layout != blk[0].w_layout || (blk[k].w_layout != QBNN_LAYOUT_MFMA32 && blk[k].w_layout != QBNN_LAYOUT_MFMA32_N24))
      return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_mc: one packed weight layout per launch (MFMA32 or MFMA32_N24)%s");
  if (blk[0].w_layout == QBNN_LAYOUT_MFMA32_N24) {       // the 16-wave 48-channel kernel (qbnn_c48.hip)
    if constexpr (NBLK == 1) { if (Cc == 48 && H == 16) return qbnn_launch_chain48_w16(&a, 1, st); }
    return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_mc: the MFMA32_N24 layout serves one 16x16x48 identity block per launch%s");
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
    if constexpr (NBLK == 1) return qbnn_launch_block_chain_ring(&a, 1, 96, false, st);
    else return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_mc: one block per launch at 96 channels (its weights stream through the LDS ring)%s");
  }
  if (Cc == 192 && H == 4) {
    if constexpr (NBLK == 1) return qbnn_launch_block_chain_ring(&a, 1, 192, false, st);
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

// ---- prepared multi-call launches: the argument blocks live in device memory (ArgsArr<A, 0>) ---------------------------------------
template <class C, int NBLK, bool STEM = false>
static int launch_block_chain_ws_dev(const ChainArgs<NBLK>* dev, int n, int items, hipStream_t st) {
  constexpr int LDS = chain_ws_lds<C, NBLK, true, STEM>();
  static std::atomic<uint64_t> attr{0};
  if (int rc_attr = ensure_dyn_lds((const void*)block_chain_ws_kernel<C, NBLK, true, STEM, 0>, attr, LDS)) return rc_attr;
  hipLaunchKernelGGL((block_chain_ws_kernel<C, NBLK, true, STEM, 0>), dim3(fused_grid_x(items, n), n), dim3(BLK_THREADS), LDS, st, ArgsArr<ChainArgs<NBLK>, 0>{dev},
                     DropSet<0>{});
  return check_launch("qbnn_block_chain_i8_multi_launch");
}
template <class CA, class CS, class CB, bool LDSW>
static int launch_block_down_ws_dev(const DownArgs* dev, int n, int items, hipStream_t st) {
  constexpr int LDS = CA::G * CA::TILE_BYTES + CA::TILE_SLACK + CB::G * CB::TILE_BYTES + CB::TILE_SLACK + DownSC<CB>::BYTES +
                      (LDSW ? WConv<CS>::BYTES + WConv<CA>::BYTES + WConv<CB>::BYTES : 0) + 3 * CB::COUT * 4;
  static std::atomic<uint64_t> attr{0};
  if (int rc_attr = ensure_dyn_lds((const void*)block_down_ws_kernel<CA, CS, CB, LDSW, 0>, attr, LDS)) return rc_attr;
  hipLaunchKernelGGL((block_down_ws_kernel<CA, CS, CB, LDSW, 0>), dim3(fused_grid_x(items, n), n), dim3(BLK_THREADS), LDS, st, ArgsArr<DownArgs, 0>{dev}, DropSet<0>{});
  return check_launch("qbnn_block_down_i8_multi_launch");
}

QBNN_EXPORT size_t qbnn_chain_multi_args_bytes(int32_t n_calls, int32_t n_blocks) {
  return (size_t)(n_calls > 0 ? n_calls : 0) * (n_blocks == 2 ? sizeof(ChainArgs<2>) : sizeof(ChainArgs<1>));
}
QBNN_EXPORT size_t qbnn_down_multi_args_bytes(int32_t n_calls) { return (size_t)(n_calls > 0 ? n_calls : 0) * sizeof(DownArgs); }

// The argument blocks go up ON the caller's stream (the destination comes from a stream-ordered allocator: a recycled block may still be
// read by earlier work of that stream, which a null-stream copy would not wait for) and the call returns when they have landed (the
// staging vector dies with the caller's frame).
// What a _multi_prepare call baked into the device-resident argument blocks, remembered per `dev_args` pointer so that _multi_launch can refuse a
// mismatching (a_hi, w_layout, n_calls, n_blocks / with_stem): those pick the kernel, and a kernel that reads the weights in another fragment
// layout than the sampler wrote gives silently wrong sums.  (A launch on a pointer this library never prepared is refused too.)
struct PreparedArgs { int32_t n_calls, a_hi, w_layout, n_blocks, with_stem, B; };
static std::mutex g_prepared_mu;
static std::unordered_map<const void*, PreparedArgs> g_prepared;
static void remember_prepared(const void* dev, const PreparedArgs& p) {
  std::lock_guard<std::mutex> lk(g_prepared_mu);
  if (g_prepared.size() > 4096) g_prepared.clear();      // (stale pointers of freed blocks: a re-prepare re-registers)
  g_prepared[dev] = p;
}
static int check_prepared(const void* dev, const PreparedArgs& want, const char* what) {
  std::lock_guard<std::mutex> lk(g_prepared_mu);
  auto it = g_prepared.find(dev);
  if (it == g_prepared.end()) return fail(QBNN_E_INVALID, "%s: dev_args was not written by the matching _multi_prepare call", what);
  const PreparedArgs& p = it->second;
  if (p.n_calls < want.n_calls || p.a_hi != want.a_hi || p.w_layout != want.w_layout || p.n_blocks != want.n_blocks || p.with_stem != want.with_stem || p.B != want.B)
    return fail(QBNN_E_INVALID, "%s: n_calls / B / a_hi / w_layout / n_blocks differ from what _multi_prepare baked into dev_args", what);
  return QBNN_OK;
}

static int upload_args(void* dev, const void* host, size_t bytes, const char* what, hipStream_t st) {
  if (hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return fail(QBNN_E_LAUNCH, "%s: copying the argument blocks to the device failed", what);
  return QBNN_OK;
}

QBNN_EXPORT int qbnn_block_chain_i8_multi_prepare(const qbnn_chain_call* calls, int32_t n_calls, int32_t with_stem, int32_t B, int32_t a_hi,
                                                  int32_t n_blocks, void* dev_args, void* stream) {
  if (!calls || n_calls <= 0 || B <= 0 || !dev_args) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi_prepare: bad argument%s");
  int rc = QBNN_OK;
  if (with_stem) {
    if (n_blocks != 2) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi_prepare: the fused stem feeds the two 32x32x24 blocks%s");
    std::vector<ChainArgs<2>> arr(n_calls);
    for (int i = 0; i < n_calls; ++i) {
      const qbnn_chain_call& k = calls[i];
      if (!k.im2col || !k.w0_packed || !k.blocks || !k.y || k.n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi_prepare: bad call entry%s");
      QConv stem;
      if ((rc = build_stem_qconv(stem, k.w0_packed, k.w0_sample_stride, k.bias0, k.s_in, k.s_w0, k.z_w0, k.s_y0, k.z_y0, a_hi))) return rc;
      if ((rc = build_chain_args<2>(arr[i], nullptr, 0, k.s_y0, k.z_y0, B, a_hi, k.blocks, k.y, k.y_sample_stride, k.n_samples, k.im2col, &stem))) return rc;
      for (int b = 0; b < 2; ++b)
        if (k.blocks[b].w_layout != calls[0].blocks[0].w_layout) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi_prepare: one weight layout per call array%s");
    }
    if ((rc = upload_args(dev_args, arr.data(), arr.size() * sizeof(arr[0]), "qbnn_block_chain_i8_multi_prepare", (hipStream_t)stream))) return rc;
    remember_prepared(dev_args, PreparedArgs{n_calls, a_hi, calls[0].blocks[0].w_layout, 2, 1, B});
    return QBNN_OK;
  }
  if (n_blocks != 1) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi_prepare: one block per call (two only behind the fused stem)%s");
  std::vector<ChainArgs<1>> arr(n_calls);
  for (int i = 0; i < n_calls; ++i) {
    const qbnn_chain_call& k = calls[i];
    if (!k.x || !k.blocks || !k.y || k.n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi_prepare: bad call entry%s");
    if ((rc = build_chain_args<1>(arr[i], k.x, k.x_sample_stride, k.s_x, k.z_x, B, a_hi, k.blocks, k.y, k.y_sample_stride, k.n_samples, nullptr, nullptr))) return rc;
    if (k.blocks[0].w_layout != calls[0].blocks[0].w_layout) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi_prepare: one weight layout per call array%s");
  }
  if ((rc = upload_args(dev_args, arr.data(), arr.size() * sizeof(arr[0]), "qbnn_block_chain_i8_multi_prepare", (hipStream_t)stream))) return rc;
  remember_prepared(dev_args, PreparedArgs{n_calls, a_hi, calls[0].blocks[0].w_layout, 1, 0, B});
  return QBNN_OK;
}

QBNN_EXPORT int qbnn_block_chain_i8_multi_launch(const void* dev_args, int32_t n_calls, int32_t with_stem, int32_t B, int32_t H, int32_t Cc,
                                                 int32_t a_hi, int32_t w_layout, int32_t n_blocks, int32_t max_samples, void* stream) {
  if (!dev_args || n_calls <= 0 || B <= 0 || max_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi_launch: bad argument%s");
  if (int rc = check_prepared(dev_args, PreparedArgs{n_calls, a_hi, w_layout, n_blocks, with_stem ? 1 : 0, B}, "qbnn_block_chain_i8_multi_launch")) return rc;
  hipStream_t st = (hipStream_t)stream;
  auto items = [&](int G) { return max_samples * ((B + G - 1) / G); };
  if (with_stem) {
    if (n_blocks != 2 || Cc != 24 || H != 32) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi_launch: the fused stem feeds the two 32x32x24 blocks only%s");
    if (w_layout == QBNN_LAYOUT_MFMA32_TAIL) return qbnn_launch_stem_chain_w16_dev(reinterpret_cast<const ChainArgs<2>*>(dev_args), n_calls, items(2), a_hi, st);
    if (w_layout != QBNN_LAYOUT_MFMA32) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi_launch: the 24-channel blocks take MFMA32 or MFMA32_TAIL weights%s");
    return launch_block_chain_ws_dev<Blk_24, 2, true>(reinterpret_cast<const ChainArgs<2>*>(dev_args), n_calls, items(Blk_24::G), st);      // MFMA32: the 8-wave kernel
  }
  if (n_blocks != 1) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi_launch: one block per call (two only behind the fused stem)%s");
  const ChainArgs<1>* dev = reinterpret_cast<const ChainArgs<1>*>(dev_args);
  if (w_layout == QBNN_LAYOUT_MFMA32_N24) {
    if (Cc == 48 && H == 16) return qbnn_launch_chain48_w16_dev(dev, n_calls, items(2), st);
    return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi_launch: the MFMA32_N24 layout serves the 16x16x48 identity block%s");
  }
  if (w_layout != QBNN_LAYOUT_MFMA32) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi_launch: unknown weight layout%s");
  if (Cc == 48 && H == 16) return launch_block_chain_ws_dev<Blk_48, 1>(dev, n_calls, items(Blk_48::G), st);
  if (Cc == 96 && H == 8) return qbnn_launch_block_chain_ring_dev(dev, n_calls, B, max_samples, 96, false, st);
  if (Cc == 192 && H == 4) {
    const bool small_items = ((B + 15) / 16) * n_calls * max_samples <= 128;
    return qbnn_launch_block_chain_ring_dev(dev, n_calls, B, max_samples, 192, small_items, st);
  }
  return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi_launch: unsupported geometry%s C=%ld H=%ld", "", Cc, H);
}

QBNN_EXPORT int qbnn_block_down_i8_multi_prepare(const qbnn_down_call* calls, int32_t n_calls, int32_t B, int32_t a_hi, void* dev_args, void* stream) {
  if (!calls || n_calls <= 0 || B <= 0 || !dev_args) return fail(QBNN_E_INVALID, "qbnn_block_down_i8_multi_prepare: bad argument%s");
  std::vector<DownArgs> arr(n_calls);
  for (int i = 0; i < n_calls; ++i) {
    const qbnn_down_call& k = calls[i];
    if (!k.x || !k.y || !k.desc || k.n_samples <= 0 || !k.desc->blk.w_a || !k.desc->blk.w_b || !k.desc->w_s)
      return fail(QBNN_E_INVALID, "qbnn_block_down_i8_multi_prepare: bad call entry%s");
    if (int rc = build_down_args(arr[i], k.x, k.x_sample_stride, k.s_x, k.z_x, B, a_hi, k.desc, k.y, k.y_sample_stride, k.n_samples)) return rc;
    if (k.desc->blk.w_layout != calls[0].desc->blk.w_layout) return fail(QBNN_E_INVALID, "qbnn_block_down_i8_multi_prepare: one weight layout per call array%s");
  }
  if (int rc = upload_args(dev_args, arr.data(), arr.size() * sizeof(arr[0]), "qbnn_block_down_i8_multi_prepare", (hipStream_t)stream)) return rc;
  remember_prepared(dev_args, PreparedArgs{n_calls, 0, calls[0].desc->blk.w_layout, 1, 0, B});      // (the down launch takes no a_hi: it is baked in)
  return QBNN_OK;
}

QBNN_EXPORT int qbnn_block_down_i8_multi_launch(const void* dev_args, int32_t n_calls, int32_t B, int32_t H, int32_t Cin, int32_t w_layout,
                                                int32_t max_samples, void* stream) {
  if (!dev_args || n_calls <= 0 || B <= 0 || max_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_block_down_i8_multi_launch: bad argument%s");
  if (int rc = check_prepared(dev_args, PreparedArgs{n_calls, 0, w_layout, 1, 0, B}, "qbnn_block_down_i8_multi_launch")) return rc;
  hipStream_t st = (hipStream_t)stream;
  const DownArgs* dev = reinterpret_cast<const DownArgs*>(dev_args);
  auto items = [&](int G) { return max_samples * ((B + G - 1) / G); };
  if (w_layout == QBNN_LAYOUT_MFMA32_N24) {
    if (Cin == 24 && H == 32) return qbnn_launch_down24_w16_dev(dev, n_calls, items(2), st);
    return fail(QBNN_E_INVALID, "qbnn_block_down_i8_multi_launch: the MFMA32_N24 layout set serves the 24 -> 48 block%s");
  }
  if (w_layout != QBNN_LAYOUT_MFMA32) return fail(QBNN_E_INVALID, "qbnn_block_down_i8_multi_launch: unknown weight layout%s");
  if (Cin == 24 && H == 32) return launch_block_down_ws_dev<D24_a, D24_s, D24_b, true>(dev, n_calls, items(D24_a::G), st);
  if (Cin == 48 && H == 16)
    return qbnn_launch_block_down_ring_dev(dev, n_calls, items(D48_a::G), 48, st);
  if (Cin == 96 && H == 8)
    return qbnn_launch_block_down_ring_dev(dev, n_calls, items(D96_a::G), 96, st);
  return fail(QBNN_E_INVALID, "qbnn_block_down_i8_multi_launch: unsupported geometry%s Cin=%ld H=%ld", "", Cin, H);
}

// ---- fused blocks with dropout (conv_resnet_mc) ------------------------------------------------------------------------------
template <int NBLK, bool STEM>
static int chain_drop_dispatch(const uint8_t* x, int64_t x_ss, float s_x, int32_t z_x, int32_t B, int32_t H, int32_t Cc, int32_t a_hi,
                               const qbnn_block_desc* blk, const qbnn_drop_desc* drop0, const qbnn_drop_desc* drops, uint8_t* y, int64_t y_ss,
                               int32_t n_samples, uint64_t seed, uint32_t sample_begin, hipStream_t st, const int8_t* stem_x, const QConv* stem,
                               float s_y0) {
  ChainArgs<NBLK> a;
  if (int rc = build_chain_args<NBLK>(a, x, x_ss, s_x, z_x, B, a_hi, blk, y, y_ss, n_samples, stem_x, stem, drops)) return rc;
  DropSet<chain_ndrop<NBLK, STEM, true>()> dr;
  constexpr int D0 = STEM ? 1 : 0;
  if constexpr (STEM) { if (int rc = fill_drop(dr.d[0], *drop0, s_y0, a_hi, seed, sample_begin)) return rc; }
  for (int k = 0; k < NBLK; ++k) {
    if (int rc = fill_drop(dr.d[D0 + 2 * k], drops[2 * k], blk[k].s_a, a_hi, seed, sample_begin)) return rc;
    if (int rc = fill_drop(dr.d[D0 + 2 * k + 1], drops[2 * k + 1], blk[k].s_b, a_hi, seed, sample_begin)) return rc;
  }
  if constexpr (STEM) {
    if (Cc != 24 || H != 32) return fail(QBNN_E_INVALID, "qbnn_stem_chain_drop_i8_mc: the fused stem feeds the 32x32x24 chain only%s");
    if (blk[0].w_layout == QBNN_LAYOUT_MFMA32_TAIL) {
      if constexpr (NBLK == 2) return qbnn_launch_stem_chain_w16_drop(a, dr, a_hi, st);
      else return fail(QBNN_E_INVALID, "qbnn_stem_chain_drop_i8_mc: the MFMA32_TAIL layout serves layers.0 + two blocks (the 16-wave kernel)%s");
    }
    if (blk[0].w_layout != QBNN_LAYOUT_MFMA32) return fail(QBNN_E_INVALID, "qbnn_stem_chain_drop_i8_mc: the 24-channel blocks take MFMA32 or MFMA32_TAIL weights%s");
    return launch_block_chain_ws_drop<Blk_24, NBLK, true>(a, dr, st);
  } else {
    if (blk[0].w_layout != QBNN_LAYOUT_MFMA32) return fail(QBNN_E_INVALID, "qbnn_block_chain_drop_i8_mc: these blocks take MFMA32 weights%s");
    if (Cc == 24 && H == 32) return launch_block_chain_ws_drop<Blk_24, NBLK, false>(a, dr, st);
    if (Cc == 48 && H == 16) {
      if constexpr (chain_ws_lds<Blk_48, NBLK, true, false, true>() <= 160 * 1024) return launch_block_chain_ws_drop<Blk_48, NBLK, false>(a, dr, st);
      else return fail(QBNN_E_INVALID, "qbnn_block_chain_drop_i8_mc: one 48-channel block per launch%s");
    }
    if constexpr (NBLK == 1) {
      if (Cc == 96 && H == 8) return qbnn_launch_block_chain_ring_drop(a, dr, 96, st);
      if (Cc == 192 && H == 4) return qbnn_launch_block_chain_ring_drop(a, dr, 192, st);
    }
    return fail(QBNN_E_INVALID, "qbnn_block_chain_drop_i8_mc: unsupported geometry%s C=%ld H=%ld", "", Cc, H);
  }
}

QBNN_EXPORT int qbnn_block_chain_drop_i8_mc(const uint8_t* x, int64_t x_ss, float s_x, int32_t z_x, int32_t B, int32_t H, int32_t Cc,
                                            int32_t a_hi, const qbnn_block_desc* host_blocks, const qbnn_drop_desc* drops, int32_t n_blocks,
                                            uint8_t* y, int64_t y_ss, int32_t n_samples, uint64_t seed, uint32_t sample_begin, void* stream) {
  if (!x || !y || !host_blocks || !drops || n_samples <= 0 || B <= 0) return fail(QBNN_E_INVALID, "qbnn_block_chain_drop_i8_mc: bad argument%s");
  for (int k = 0; k < n_blocks; ++k)
    if (!host_blocks[k].w_a || !host_blocks[k].w_b) return fail(QBNN_E_INVALID, "qbnn_block_chain_drop_i8_mc: NULL weights%s");
  hipStream_t st = (hipStream_t)stream;
  if (n_blocks == 1) return chain_drop_dispatch<1, false>(x, x_ss, s_x, z_x, B, H, Cc, a_hi, host_blocks, nullptr, drops, y, y_ss, n_samples, seed, sample_begin, st, nullptr, nullptr, 0.f);
  if (n_blocks == 2) return chain_drop_dispatch<2, false>(x, x_ss, s_x, z_x, B, H, Cc, a_hi, host_blocks, nullptr, drops, y, y_ss, n_samples, seed, sample_begin, st, nullptr, nullptr, 0.f);
  return fail(QBNN_E_INVALID, "qbnn_block_chain_drop_i8_mc: 1 or 2 blocks per launch%s");
}

QBNN_EXPORT int qbnn_stem_chain_drop_i8_mc(const int8_t* im2col, int32_t B, const int8_t* w0_packed, int64_t w0_ss, const float* bias0,
                                           float s_x, float s_w0, int32_t z_w0, float s_y0, int32_t z_y0, int32_t a_hi,
                                           const qbnn_drop_desc* drop0, const qbnn_block_desc* host_blocks, const qbnn_drop_desc* drops,
                                           int32_t n_blocks, uint8_t* y, int64_t y_ss, int32_t n_samples, uint64_t seed, uint32_t sample_begin,
                                           void* stream) {
  if (!im2col || !w0_packed || !y || !host_blocks || !drop0 || !drops || n_samples <= 0 || B <= 0)
    return fail(QBNN_E_INVALID, "qbnn_stem_chain_drop_i8_mc: bad argument%s");
  for (int k = 0; k < n_blocks; ++k)
    if (!host_blocks[k].w_a || !host_blocks[k].w_b) return fail(QBNN_E_INVALID, "qbnn_stem_chain_drop_i8_mc: NULL weights%s");
  QConv stem;
  if (int rc = build_stem_qconv(stem, w0_packed, w0_ss, bias0, s_x, s_w0, z_w0, s_y0, z_y0, a_hi)) return rc;
  hipStream_t st = (hipStream_t)stream;
  // the chain's input is the dropped conv0 output: scale drop0->s_out, zero point drop0->z_m
  if (n_blocks == 2) return chain_drop_dispatch<2, true>(nullptr, 0, drop0->s_out, drop0->z_m, B, 32, 24, a_hi, host_blocks, drop0, drops, y, y_ss, n_samples, seed, sample_begin, st, im2col, &stem, s_y0);
  if (n_blocks == 1) return chain_drop_dispatch<1, true>(nullptr, 0, drop0->s_out, drop0->z_m, B, 32, 24, a_hi, host_blocks, drop0, drops, y, y_ss, n_samples, seed, sample_begin, st, im2col, &stem, s_y0);
  return fail(QBNN_E_INVALID, "qbnn_stem_chain_drop_i8_mc: 1 or 2 blocks per launch%s");
}

QBNN_EXPORT int qbnn_block_down_drop_i8_mc(const uint8_t* x, int64_t x_ss, float s_x, int32_t z_x, int32_t B, int32_t H, int32_t Cin,
                                           int32_t a_hi, const qbnn_down_desc* d, const qbnn_drop_desc* drops, uint8_t* y, int64_t y_ss,
                                           int32_t n_samples, uint64_t seed, uint32_t sample_begin, void* stream) {
  if (!x || !y || !d || !drops || n_samples <= 0 || B <= 0 || !d->blk.w_a || !d->blk.w_b || !d->w_s)
    return fail(QBNN_E_INVALID, "qbnn_block_down_drop_i8_mc: bad argument%s");
  if (d->blk.w_layout != QBNN_LAYOUT_MFMA32) return fail(QBNN_E_INVALID, "qbnn_block_down_drop_i8_mc: the blocks with dropout take MFMA32 weights%s");
  DownArgs a;
  if (int rc = build_down_args(a, x, x_ss, s_x, z_x, B, a_hi, d, y, y_ss, n_samples, drops)) return rc;
  DropSet<3> dr;
  if (int rc = fill_drop(dr.d[0], drops[0], d->blk.s_a, a_hi, seed, sample_begin)) return rc;
  if (int rc = fill_drop(dr.d[1], drops[1], d->blk.s_b, a_hi, seed, sample_begin)) return rc;
  if (int rc = fill_drop(dr.d[2], drops[2], d->s_s, a_hi, seed, sample_begin)) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 24 && H == 32) return launch_block_down_ws_drop<D24_a, D24_s, D24_b, true, false>(a, dr, st);
  // 48 -> 96 and 96 -> 192: the ring form with one-bit mask tables (round 5); QBNN_DOWN_RING=0: the per-wave L2-streaming kernels
  if (Cin == 48 && H == 16) return qbnn_launch_block_down_ring_drop(a, dr, 48, st);
  if (Cin == 96 && H == 8) return qbnn_launch_block_down_ring_drop(a, dr, 96, st);
  return fail(QBNN_E_INVALID, "qbnn_block_down_drop_i8_mc: unsupported geometry%s Cin=%ld H=%ld", "", Cin, H);
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
      const int lay = calls[c0].blocks[0].w_layout;
      for (int i = 0; i < n; ++i)
        for (int k = 0; k < 2; ++k)
          if (calls[c0 + i].blocks[k].w_layout != lay) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi: one weight layout per call array%s");
      if (lay == QBNN_LAYOUT_MFMA32_TAIL) rc = qbnn_launch_stem_chain_w16(arr, n, a_hi, st);
      else if (lay == QBNN_LAYOUT_MFMA32) rc = launch_block_chain_ws_multi<Blk_24, 2, true, NM2>(arr, n, st);
      else return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi: the 24-channel blocks take MFMA32 or MFMA32_TAIL weights%s");
    } else {
      if (n_blocks != 1) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi: one block per call (two only behind the fused stem)%s");
      ChainArgs<1> arr[QBNN_FUSED_CALLS];
      for (int i = 0; i < n; ++i) {
        const qbnn_chain_call& k = calls[c0 + i];
        if (!k.x || !k.blocks || !k.y || k.n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi: bad call entry%s");
        if ((rc = build_chain_args<1>(arr[i], k.x, k.x_sample_stride, k.s_x, k.z_x, B, a_hi, k.blocks, k.y, k.y_sample_stride, k.n_samples, nullptr, nullptr))) return rc;
      }
      bool n24 = false;
      for (int i = 0; i < n; ++i) {
        const int lay = calls[c0 + i].blocks[0].w_layout;
        if (lay != QBNN_LAYOUT_MFMA32 && lay != QBNN_LAYOUT_MFMA32_N24) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi: unknown weight layout%s");
        if (i && (lay == QBNN_LAYOUT_MFMA32_N24) != n24) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi: one weight layout per call array%s");
        n24 = lay == QBNN_LAYOUT_MFMA32_N24;
      }
      if (n24 && !(Cc == 48 && H == 16)) return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi: the MFMA32_N24 layout serves the 16x16x48 identity block%s");
      if (Cc == 48 && H == 16) rc = n24 ? qbnn_launch_chain48_w16(arr, n, st) : launch_block_chain_ws_multi<Blk_48, 1, false, QBNN_FUSED_CALLS>(arr, n, st);
      else if (Cc == 96 && H == 8) rc = qbnn_launch_block_chain_ring(arr, n, 96, false, st);
      else if (Cc == 192 && H == 4) {
        const bool small_items = ((B + 15) / 16) * n <= 128;
        rc = qbnn_launch_block_chain_ring(arr, n, 192, small_items, st);
      }
      else return fail(QBNN_E_INVALID, "qbnn_block_chain_i8_multi: unsupported geometry%s C=%ld H=%ld", "", Cc, H);
    }
    if (rc) return rc;
    c0 += n;
  }
  return QBNN_OK;
}
