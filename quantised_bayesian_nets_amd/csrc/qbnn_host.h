// Host-side helpers shared by the translation units of libqbnn_hip.so (error reporting, launch checks, the qparam
// blocks of the conv kernels).  Internal: the public interface is include/qbnn.h.
#ifndef QBNN_HOST_H_
#define QBNN_HOST_H_
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <atomic>
#include <type_traits>

#include "../../include/qbnn.h"
#include "qbnn_common.h"
#include "qbnn_conv.h"

static inline int fail(int code, const char* fmt, const char* a = "", long b = 0, long c = 0) {
  char buf[512];
  snprintf(buf, sizeof(buf), fmt, a, b, c);
  return qbnn_fail_msg(code, buf);
}
static inline int check_launch(const char* what) { return qbnn_check_launch_msg(what); }
static inline int ensure_dyn_lds(const void* fn, std::atomic<uint64_t>& done, int bytes) { return qbnn_ensure_dyn_lds(fn, &done, bytes); }
#define g_noise_dev (qbnn_noise_dev())      // this thread's device noise source (qbnn_set_device_noise_source) or nullptr
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

static int fill_qconv(QConv& p, const int8_t* w, int64_t w_ss, const float* bias, const qbnn_conv_desc* d) {
  if (d->z_x < 0 || d->z_x > 127 || d->a_hi > 127 || d->a_hi < 1 || d->z_y < 0 || d->z_y > 127)
    return fail(QBNN_E_INVALID, "qbnn conv: activations must be <= 7 bit with zero points in [0,127] (reference quant_utils.py:120)%s");
  p.w = w; p.w_ss = w_ss; p.bias = d->has_bias ? bias : nullptr;
  p.z_x = d->z_x; p.z_w = d->z_w; p.z_y = d->z_y;
  const float atw = d->s_x * d->s_w;     // qconv.cpp GetQuantizationParams: float * float
  p.rcp = 1.0f / atw;                    // FBGEMM act_times_w_rcp
  p.mult = atw / d->s_y;                 // output_multiplier_float
  const int lo = d->relu ? d->z_y : 0, hi = d->a_hi < 255 ? d->a_hi : 255;
  p.vlo = (float)(lo - d->z_y); p.vhi = (float)(hi - d->z_y);
  p.s_y = d->s_y; p.nzs_y = (float)(-d->z_y) * d->s_y;
  p.dl_y = fmaf(d->s_y, (float)d->z_y, p.nzs_y);
  return QBNN_OK;
}

static int fill_qadd(QAdd& a, const qbnn_conv_desc* d) {
  if (d->z_o < 0 || d->z_o > 127 || d->z_r < 0 || d->z_r > 127)
    return fail(QBNN_E_INVALID, "qbnn conv: add zero points must be in [0,127]%s");
  a.s_r = d->s_r; a.nzs_r = (float)(-d->z_r) * d->s_r; a.z_r = d->z_r;
  a.dl_r = fmaf(d->s_r, (float)d->z_r, a.nzs_r);
  a.inv_s_o = 1.0f / d->s_o; a.z_o = d->z_o;
  a.vhi = (float)((d->a_hi < 255 ? d->a_hi : 255) - d->z_o);
  return QBNN_OK;
}

// 16-wave layer-1 kernel (qbnn_w16.hip): layers.0 + two identity blocks at 32 x 32 x 24, `n` <= 4 argument blocks in one grid
int qbnn_launch_stem_chain_w16(const ChainArgs<2>* arr, int n, int a_hi, hipStream_t st);
int qbnn_launch_stem_chain_w16_dev(const ChainArgs<2>* dev, int n, int items, int a_hi, hipStream_t st);               // ... any number of argument blocks in device memory
int qbnn_launch_stem_chain_w16_drop(const ChainArgs<2>& a, const DropSet<5>& dr, int a_hi, hipStream_t st);      // ... with the five dropouts of conv_resnet_mc
// 16-wave 48-channel identity block on QBNN_LAYOUT_MFMA32_N24 weights (qbnn_c48.hip, round 5): `n` <= QBNN_FUSED_CALLS argument blocks by
// value, or any number in device memory
int qbnn_launch_chain48_w16(const ChainArgs<1>* arr, int n, hipStream_t st);
int qbnn_launch_chain48_w16_dev(const ChainArgs<1>* dev, int n, int items, hipStream_t st);
// ... and the 24 -> 48 down-sampling block (stem.0 as MFMA32_N24_TAIL, shortcut and stem.3 as MFMA32_N24)
int qbnn_launch_down24_w16(const DownArgs* arr, int n, hipStream_t st);
int qbnn_launch_down24_w16_dev(const DownArgs* dev, int n, int items, hipStream_t st);
// Ring form of the wide down-sampling blocks (qbnn_down_ring.hip): 48 -> 96 at 16 x 16 and 96 -> 192 at 8 x 8; `n` argument blocks by value
// (n <= QBNN_FUSED_CALLS) or any number in device memory.  (Round 6: the only form -- the L2-streaming instantiations of rounds 1 - 3 left the build.)
int qbnn_launch_block_down_ring(const DownArgs* arr, int n, int Cin, hipStream_t st);
int qbnn_launch_block_down_ring_dev(const DownArgs* dev, int n, int items, int Cin, hipStream_t st);
int qbnn_launch_block_down_ring_drop(const DownArgs& a, const DropSet<3>& dr, int Cin, hipStream_t st);      // ... with the block's three dropouts (conv_resnet_mc)
// The wide identity blocks with the same K loop (qbnn_chain_ring.hip): 96 channels at 8 x 8, 192 at 4 x 4 (`small_items`: 8 instead of 16 images per
// work item).  (Round 6: the only form; round 3's block_chain_ald_kernel is tools/experiments/r03_block_chain_ald_kernel.hip.txt.)
int qbnn_launch_block_chain_ring(const ChainArgs<1>* arr, int n, int Cc, bool small_items, hipStream_t st);
int qbnn_launch_block_chain_ring_dev(const ChainArgs<1>* dev, int n, int B, int max_samples, int Cc, bool small_items, hipStream_t st);
int qbnn_launch_block_chain_ring_drop(const ChainArgs<1>& a, const DropSet<2>& dr, int Cc, hipStream_t st);      // ... with the block's two dropouts (conv_resnet_mc)
// QBNN_W16=0 selects the 8-wave kernels of qbnn_blocks.hip everywhere (A/B checks)
static inline bool qbnn_use_w16() {
  static const bool v = [] { const char* e = getenv("QBNN_W16"); return !(e && e[0] == '0'); }();
  return v;
}
#endif
