/*
 * qbnn.h -- C ABI of libqbnn_hip.so: the MI355X (gfx950) Monte-Carlo inference path for
 * quantised Bayesian networks.
 *
 * The reference (martinferianc/quantised-bayesian-nets) is pure Python on PyTorch: its
 * "FFI" for this path is the set of torch ops its layer classes call.  Each entry point
 * below names the reference call sites (file:line under the reference root) it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named host_*; the caller owns all buffers;
 *   - every call enqueues work on `stream` (a hipStream_t passed as void*) and returns
 *     without synchronising; the library keeps no mutable global state except the
 *     thread-local error string;
 *   - return value 0 = ok, negative = error (QBNN_E_*); text via qbnn_last_error();
 *   - activations: uint8 NHWC, one tensor per MC sample: [S][B][H][W][C]; a sample stride
 *     of 0 means "shared by all samples";
 *   - sampled weights: int8 in the MFMA-fragment-packed layout described at
 *     qbnn_packed_weight_bytes(), one slab per MC sample.
 */
#ifndef QBNN_H_
#define QBNN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version of this header: qbnn_version() of the loaded library must equal it (bumped whenever a prototype below changes;
 * 2 = round 5: qbnn_block_chain_i8_multi_launch takes a_hi and w_layout, qbnn_block_down_i8_multi_launch w_layout, qbnn_block_desc carries w_layout; round 4's `stream` argument
 * of the _multi_prepare calls). */
#define QBNN_ABI_VERSION 2

#define QBNN_OK 0
#define QBNN_E_INVALID (-1)     /* bad argument / unsupported shape   */
#define QBNN_E_LAUNCH (-2)      /* HIP launch error                   */

#define QBNN_LAYOUT_MFMA32 0    /* [NT][KS][64 lanes][16 B] fragments */
#define QBNN_LAYOUT_ROWMAJOR 1  /* [Cout][K]                          */
#define QBNN_LAYOUT_MFMA32_N24 2 /* fragments with 24 output channels (+ a ones row) per tile: the fused 48-channel kernels of round 5 */
#define QBNN_LAYOUT_MFMA32_TAIL 3 /* MFMA32 with the ragged ends of the kernel rows gathered into one k-step (72-byte rows: 7 k-steps instead of
                                   * 9): the 24-channel convs behind the fused stem (qbnn_stem_chain_i8_mc / _drop_ / the multi forms)        */
#define QBNN_LAYOUT_MFMA32_N24_TAIL 4 /* both of the previous two: stem.0 (24 -> 48, 3x3) of the first down-sampling block on the 16-wave kernel
                                       * (24 + 1 channel tiles AND the gathered tails k-step)                                                    */

/* qbnn_block_desc.flags.  POOL_OUT: the block's output leaves as AvgPool2d(H) of it -- y is [S][B][C] quint8 with the block output's (scale, zero
 * point): q = clamp(rne((sum - H*H z) / (H*H)) + z), the head's first step (models_bbb.py:209, :240: nn.AvgPool2d(4) behind the last BasicBlock), so
 * the 4 x 4 x 192 map never goes to HBM and qbnn_head_i8_mc runs on the pooled tensor with k = 1.  Served by the 4 x 4 x 192 identity block of
 * qbnn_block_chain_i8_mc (one block per launch, LDS-ring kernel, MFMA32 weights); any other geometry AND every other entry point (the _multi,
 * _multi_prepare and _drop_ forms) answers QBNN_E_INVALID. */
#define QBNN_BLOCK_POOL_OUT 1

/* Scalars of the int8 weight-sampling chain
 *   noise  = quantize_per_tensor(eps, NOISE_SCALE, 0, qint8)      conv_q.py:113-115, linear_q.py:86-88
 *   t      = mul_noise.mul(std, noise)                            conv_q.py:118,     linear_q.py:91
 *   weight = add_weight.add(weight, t)                            (same lines)
 *   weight = clamp_weight(weight, args)                           conv_q.py:119; src/utils.py:32-37
 * derived on the host exactly as ATen derives them (see DESIGN.md "Arithmetic contracts"). */
typedef struct qbnn_sample_params {
  float inv_noise_scale;   /* 1.0f / (float)NOISE_SCALE                                   */
  float mul_multiplier;    /* (float)((double)s_std * (double)NOISE_SCALE / (double)s_mul) */
  int32_t z_sigma;         /* std.q_zero_point()                                          */
  int32_t z_mul;           /* mul_noise.zero_point                                        */
  float s_w, nzs_w;        /* weight.q_scale(), (float)(-z_w) * s_w                       */
  float s_mul, nzs_mul;    /* mul_noise.scale,  (float)(-z_mul) * s_mul                   */
  float inv_s_add;         /* 1.0f / add_weight.scale                                     */
  int32_t z_add;           /* add_weight.zero_point                                       */
  int32_t w_lo, w_hi;      /* INT_BOUNDS[weight_precision], src/utils.py:19-20            */
} qbnn_sample_params;

/* Bytes of one sample's weights for a [cout][k] layer in `layout`.
 * QBNN_LAYOUT_MFMA32 (see csrc/qbnn_kernels.hip "Packed weight layout"): k = rows * krow, every kernel row of
 * krow = KW*Cin bytes is padded to a multiple of 32; a ragged cout (cout % 32 != 0) carries an extra all-ones row. */
size_t qbnn_packed_weight_bytes(int32_t cout, int32_t k, int32_t krow, int32_t layout);

/* Host helper: pack a logical int8 [cout][k] matrix (OHWI flattening of the reference's OIHW
 * weight: k = (kh*KW + kw)*Cin + c) into `layout`.  Pad entries are written as 0. */
int qbnn_pack_weights_host(const int8_t* host_src, int32_t cout, int32_t k, int32_t krow, int32_t layout, int8_t* host_dst);

/* Fused MC-batched weight sampler.  Replaces, per layer and per MC sample, the chain
 * normal_() -> quantize_per_tensor -> quantized::mul -> quantized::add -> clamp_weight
 * (conv_q.py:113-119, :198-205; linear_q.py:86-92, :160-167).
 *   mu_packed / sigma_packed : the layer's qint8 `weight` / `std` in `layout`
 *   the QUANTISED noise eps_q of element i (OHWI flat index) of sample s is drawn directly from its discrete distribution
 *     (P(k) of clamp(rne(N(0,1) / NOISE_SCALE), -128, 127)) with one Philox word and Walker's alias table (csrc/qbnn_eps_table.h):
 *     u = philox4x32_10(ctr = {i >> 2, layer_id, sample_begin + s, 0}, key = seed)[i & 3];  c = u >> 24;
 *     eps_q = ((u & 0xffffff) < thr[c] ? c : alias[c]) - 128
 *   unless eps_in != NULL: then the fp32 eps_in[s * cout * k + i] is quantised as the reference does (parity mode).
 *   w_out + s * w_sample_stride receives sample s in `layout`. */
int qbnn_sample_weights_i8(const int8_t* mu_packed, const int8_t* sigma_packed, int32_t cout, int32_t k, int32_t krow,
                           int32_t layout, const qbnn_sample_params* host_params, uint64_t seed,
                           uint32_t layer_id, uint32_t sample_begin, int32_t n_samples, const float* eps_in,
                           int8_t* w_out, int64_t w_sample_stride, void* stream);

/* Captured-graph mode.  Kernel arguments are frozen when a launch is captured into a HIP graph, but every MC evaluation wants new
 * noise.  While a device address is set on the CALLING THREAD (dev_seed3 != NULL), every sampler / dropout launch made from it
 * (qbnn_sample_weights_i8(_multi), qbnn_dropout_q_mc, qbnn_sample_weights_f32(_strided,_ohwi)) reads
 *   dev_seed3[0..2] = { (uint32)seed, (uint32)(seed >> 32), sample_begin }
 * from device memory at run time instead of using its `seed` / `sample_begin` arguments; NULL restores the arguments.  The host
 * updates the three words (an ordinary stream-ordered copy) before each replay.  This is the library's only mutable state besides
 * the last-error string: thread-local, set and cleared explicitly around a capture. */
int qbnn_set_device_noise_source(const uint32_t* dev_seed3);

/* The same sampler for up to 24 layers in ONE launch (one launch per MC batch instead of one per layer).
 * Results are identical to calling qbnn_sample_weights_i8 per entry (Philox path only). */
typedef struct qbnn_sampler_layer {
  const int8_t* mu_packed; const int8_t* sigma_packed;
  int8_t* w_out; int64_t w_sample_stride;
  int32_t cout, k, krow, layout;
  uint32_t layer_id;
  qbnn_sample_params params;
} qbnn_sampler_layer;

int qbnn_sample_weights_i8_multi(const qbnn_sampler_layer* host_layers, int32_t n_layers, uint64_t seed,
                                 uint32_t sample_begin, int32_t n_samples, void* stream);

/* One int8 stochastic conv layer for S MC samples, including everything the reference model
 * applies between this conv and the next one:
 *   quantized.functional.conv2d / quantized::conv2d_relu      conv_q.py:120-125, :206-209
 *   clamp_activation                                          src/utils.py:25-30 (models_bbb.py:173-174)
 *   and, when has_res: Add (quantized::add) -> clamp -> ReLU -> clamp   models_bbb.py:179-182
 * Supported geometries: the conv_resnet_bbb layers. */
typedef struct qbnn_conv_desc {
  int32_t B, H, W, Cin, Cout, ksize, stride, pad;
  float s_x; int32_t z_x;          /* input activation qparams                                  */
  float s_w; int32_t z_w;          /* sampled weight qparams = add_weight.scale / .zero_point   */
  float s_y; int32_t z_y;          /* layer output qparams (module .scale / .zero_point)        */
  int32_t relu;                    /* ConvReLU2d: lower clamp = z_y                             */
  int32_t a_hi;                    /* UINT_BOUNDS[activation_precision][1]                      */
  int32_t has_bias;
  int32_t has_res;                 /* fuse residual add + ReLU of BasicBlock                    */
  float s_r; int32_t z_r;          /* residual operand qparams                                  */
  float s_o; int32_t z_o;          /* Add output qparams (add.add.scale / zero_point)           */
  int32_t x_is_centered_im2col;    /* layer 0 only: x is the int8 im2col tensor of qbnn_im2col3x3_c3 */
} qbnn_conv_desc;

int qbnn_conv2d_i8_mc(const uint8_t* x, int64_t x_sample_stride, const int8_t* w_packed, int64_t w_sample_stride,
                      const float* bias, const uint8_t* res, int64_t res_sample_stride, uint8_t* y,
                      int64_t y_sample_stride, int32_t n_samples, const qbnn_conv_desc* host_desc, void* stream);

/* The same conv with the ops an MC-Dropout graph puts behind it (mcdropout/models_mc.py:116-160) in its epilogue:
 *   BernoulliDropout.forward (mcdropout/dropout.py:15-40) on the conv output -- qbnn_dropout_q_mc's arithmetic and mask stream
 *       (index b * Cout + c, tensor id drop_layer_id, MC sample sample_begin + s; mask_in fp32 [S][B][Cout] = parity mode);
 *   add != 0: then quantized::add(., other) -> clamp -> ReLU -> clamp (qbnn_add_relu_q_mc's arithmetic with relu = 1); (s_a, z_m)
 *       are the qparams of the first operand, the dropped conv output after mul_scalar: s_a = s_m / (1 - p);
 *       other [S|1][B][Ho][Wo][Cout] quint8 with (s_b, z_b).
 * d->has_res and d->x_is_centered_im2col must be 0; z_m in [0,127].  Output qparams are the caller's bookkeeping: (s_o, z_o) with
 * add, else (s_m / (1 - p), z_m).  Same bits as qbnn_conv2d_i8_mc -> qbnn_dropout_q_mc (-> qbnn_add_relu_q_mc). */
typedef struct qbnn_post_desc {
  float keep_prob; float s_m; int32_t z_m; uint32_t drop_layer_id;
  int32_t add; float s_a; float s_b; int32_t z_b; float s_o; int32_t z_o;
} qbnn_post_desc;

int qbnn_conv2d_i8_post_mc(const uint8_t* x, int64_t x_sample_stride, const int8_t* w_packed, int64_t w_sample_stride,
                           const float* bias, uint8_t* y, int64_t y_sample_stride, int32_t n_samples,
                           const qbnn_conv_desc* host_desc, const qbnn_post_desc* host_post, const float* mask_in,
                           const uint8_t* other, int64_t other_sample_stride, uint64_t seed, uint32_t sample_begin, void* stream);

/* A chain of identity BasicBlocks (models_bbb.py:170-183, no shortcut conv) fused in one persistent kernel -- 1 or 2 blocks at
 * 24 / 48 channels (the blocks' weights stay in LDS), 1 block per call at 96 / 192 channels (its weights stream through LDS):
 *   per block: stem.0 ConvReLU2d -> clamp -> stem.3 Conv2d -> clamp -> Add(block input) -> clamp -> ReLU -> clamp.
 * Activations stay in LDS between the convs; HBM traffic is one read of x and one write of y per image.
 * Same results as the corresponding sequence of qbnn_conv2d_i8_mc calls. */
typedef struct qbnn_block_desc {
  const int8_t* w_a; int64_t w_a_sample_stride; const float* bias_a;   /* stem.0 sampled weights (MFMA32 layout), bias */
  float s_wa; int32_t z_wa;                                            /* stem.0 add_weight.scale / zero_point        */
  float s_a; int32_t z_a;                                              /* stem.0 output qparams                       */
  const int8_t* w_b; int64_t w_b_sample_stride; const float* bias_b;   /* stem.3                                      */
  float s_wb; int32_t z_wb;
  float s_b; int32_t z_b;
  float s_o; int32_t z_o;                                              /* add.add.scale / zero_point                  */
  int32_t w_layout;                /* packed layout of the block's weights: QBNN_LAYOUT_MFMA32 (0, the default of a zeroed descriptor: every
                                    * geometry), or the layout set of a 16-wave kernel (round 5):
                                    *   QBNN_LAYOUT_MFMA32_TAIL -- the two 24-channel blocks behind the fused stem (w_a, w_b in that layout);
                                    *   QBNN_LAYOUT_MFMA32_N24  -- the 16x16x48 identity block (w_a, w_b), and the 24 -> 48 down block:
                                    *                              w_b and w_s as MFMA32_N24, w_a (3x3 on 24 channels) as MFMA32_N24_TAIL */
  int32_t flags;                 /* QBNN_BLOCK_* bits; 0 for the plain block */
} qbnn_block_desc;

int qbnn_block_chain_i8_mc(const uint8_t* x, int64_t x_sample_stride, float s_x, int32_t z_x, int32_t B, int32_t H,
                           int32_t C, int32_t a_hi, const qbnn_block_desc* host_blocks, int32_t n_blocks, uint8_t* y,
                           int64_t y_sample_stride, int32_t n_samples, void* stream);

/* A down-sampling BasicBlock (stride 2, Cout = 2 Cin; models_bbb.py:146-183) fused in one persistent kernel:
 *   shortcut.0 (1x1/s2 Conv2d) -> clamp ;  stem.0 (3x3/s2 ConvReLU2d) -> clamp -> stem.3 (3x3 Conv2d) -> clamp ;
 *   Add(stem, shortcut) -> clamp -> ReLU -> clamp.     `blk` describes stem.0 / stem.3 / add as in qbnn_block_desc. */
typedef struct qbnn_down_desc {
  qbnn_block_desc blk;
  const int8_t* w_s; int64_t w_s_sample_stride; const float* bias_s;   /* shortcut.0 sampled weights, bias */
  float s_ws; int32_t z_ws;                                            /* shortcut.0 add_weight qparams    */
  float s_s; int32_t z_s;                                              /* shortcut.0 output qparams        */
} qbnn_down_desc;

/* layers.0 (ConvReLU2d 3 -> 24 on the centred 27-tap patches of qbnn_im2col3x3_c3, shared by all samples) fused in front
 * of the 32x32x24 identity chain of qbnn_block_chain_i8_mc: the first conv's output (the network's largest activation)
 * never reaches HBM.  Same arithmetic as qbnn_conv2d_i8_mc(x_is_centered_im2col) followed by the chain.
 * Replaces models_bbb.py:226-232 (layers.0 ... layers.3) of the converted model. */
int qbnn_stem_chain_i8_mc(const int8_t* im2col, int32_t B, const int8_t* w0_packed, int64_t w0_sample_stride, const float* bias0,
                          float s_x, float s_w0, int32_t z_w0, float s_y0, int32_t z_y0, int32_t a_hi,
                          const qbnn_block_desc* host_blocks, int32_t n_blocks, uint8_t* y, int64_t y_sample_stride,
                          int32_t n_samples, void* stream);

int qbnn_block_down_i8_mc(const uint8_t* x, int64_t x_sample_stride, float s_x, int32_t z_x, int32_t B, int32_t H,
                          int32_t Cin, int32_t a_hi, const qbnn_down_desc* host_desc, uint8_t* y, int64_t y_sample_stride,
                          int32_t n_samples, void* stream);

/* ---- the fused blocks with a quantised channel dropout behind every conv (conv_resnet_mc) ---------------------------------
 * reference mcdropout/models_mc.py:116-160 (BasicBlock: stem = ConvReLU2d, Dropout, Conv2d, Dropout; shortcut = Conv2d, Dropout;
 * Add; ReLU) and :162-211 (layers.0 ConvReLU2d, layers.3 Dropout in front of the blocks); dropout.py:15-40 per dropout:
 *   mask ~ Bernoulli(keep_prob) per (sample, image, channel) -> quantize_per_tensor(mask, s_m, z_m) -> quantized::mul(x, mask_q)
 *   with output qparams (s_m, z_m) -> clamp_activation -> mul_scalar(1 / (1 - p)): scale becomes s_out, integers unchanged.
 * The dropouts run in the convs' epilogues from per-work-item mask tables in LDS; the weights are the converted model's fixed
 * qint8 tensors in the MFMA32 layout (sample stride 0).  Mask stream: slot i = b * C + c of MC sample s draws
 * philox4x32_10(ctr = {i >> 2, layer_id, sample_begin + s, 1}, key = seed)[i & 3], kept when (u >> 8) * 2^-24 < keep_prob -- the
 * stream of qbnn_dropout_q_mc / qbnn_conv2d_i8_post_mc; `mask_in` (fp32 [n_samples][B][C], Bernoulli 0 / 1) replaces the draw.
 * Same bits as the per-layer calls (qbnn_conv2d_i8_post_mc per conv).  z_m in [0, 127]. */
typedef struct qbnn_drop_desc {
  float keep_prob;                 /* 1 - p                                                      */
  float s_m; int32_t z_m;          /* mul_mask.scale / zero_point                                */
  float s_out;                     /* scale after mul_scalar: s_m * multiplier                   */
  uint32_t layer_id;               /* Philox tensor id: index of this dropout among the model's  */
  const float* mask_in;            /* optional injected mask, else NULL                          */
} qbnn_drop_desc;

/* identity blocks; drops[2 k] sits behind block k's stem.0 (reference stem.3), drops[2 k + 1] behind its second conv (stem.6) */
int qbnn_block_chain_drop_i8_mc(const uint8_t* x, int64_t x_sample_stride, float s_x, int32_t z_x, int32_t B, int32_t H,
                                int32_t C, int32_t a_hi, const qbnn_block_desc* host_blocks, const qbnn_drop_desc* drops,
                                int32_t n_blocks, uint8_t* y, int64_t y_sample_stride, int32_t n_samples, uint64_t seed,
                                uint32_t sample_begin, void* stream);
/* layers.0 + layers.3 (drop0) + the two 32x32x24 identity blocks */
int qbnn_stem_chain_drop_i8_mc(const int8_t* im2col, int32_t B, const int8_t* w0_packed, int64_t w0_sample_stride, const float* bias0,
                               float s_x, float s_w0, int32_t z_w0, float s_y0, int32_t z_y0, int32_t a_hi,
                               const qbnn_drop_desc* drop0, const qbnn_block_desc* host_blocks, const qbnn_drop_desc* drops,
                               int32_t n_blocks, uint8_t* y, int64_t y_sample_stride, int32_t n_samples, uint64_t seed,
                               uint32_t sample_begin, void* stream);
/* down-sampling block; drops[0] behind stem.0 (reference stem.3), drops[1] behind the second conv (stem.6), drops[2] behind the
 * shortcut conv (shortcut.2) */
int qbnn_block_down_drop_i8_mc(const uint8_t* x, int64_t x_sample_stride, float s_x, int32_t z_x, int32_t B, int32_t H,
                               int32_t Cin, int32_t a_hi, const qbnn_down_desc* host_desc, const qbnn_drop_desc* drops, uint8_t* y,
                               int64_t y_sample_stride, int32_t n_samples, uint64_t seed, uint32_t sample_begin, void* stream);

/* ---- several independent calls in ONE launch (ensemble members) -------------------------------------------------------
 * reference sgld/models_sgld.py:214-288: `Network` holds args.samples deterministic members, each a converted network of its own
 * (own weights, biases AND quantisation parameters), evaluated one after the other.  Members cannot ride the MC-sample
 * dimension of the calls above (qparams are per call), so these entry points take an ARRAY of calls and run them side by
 * side in one grid (any n_calls; the library splits them into launches of up to 8 -- 4 with the fused stem, 32 for the head).  Every call
 * means exactly what the single-call entry point means; all calls of one array share the geometry arguments. */
typedef struct qbnn_chain_call {
  const uint8_t* x; int64_t x_sample_stride; float s_x; int32_t z_x;      /* as qbnn_block_chain_i8_mc (unused behind the stem) */
  const qbnn_block_desc* blocks;                                          /* n_blocks descriptors                                */
  uint8_t* y; int64_t y_sample_stride; int32_t n_samples;
  /* with_stem != 0: the arguments of qbnn_stem_chain_i8_mc */
  const int8_t* im2col; const int8_t* w0_packed; int64_t w0_sample_stride; const float* bias0;
  float s_in; float s_w0; int32_t z_w0; float s_y0; int32_t z_y0;
} qbnn_chain_call;
int qbnn_block_chain_i8_multi(const qbnn_chain_call* calls, int32_t n_calls, int32_t with_stem, int32_t B, int32_t H, int32_t C,
                              int32_t a_hi, int32_t n_blocks, void* stream);

typedef struct qbnn_down_call {
  const uint8_t* x; int64_t x_sample_stride; float s_x; int32_t z_x;
  const qbnn_down_desc* desc;
  uint8_t* y; int64_t y_sample_stride; int32_t n_samples;
} qbnn_down_call;
int qbnn_block_down_i8_multi(const qbnn_down_call* calls, int32_t n_calls, int32_t B, int32_t H, int32_t Cin, int32_t a_hi, void* stream);

/* Prepared form of the two calls above, for call arrays that are replayed (an ensemble's members at a fixed batch size): _prepare
 * writes the argument blocks of ALL n_calls calls into caller-owned device memory once (qbnn_*_multi_args_bytes bytes, 16-byte aligned;
 * copied on `stream` -- the device block typically comes from a stream-ordered allocator -- and, UNLIKE every other entry point of this
 * header, the call returns only once the copy has landed (it synchronises `stream`: the staging copy of the blocks is a host vector
 * that dies with the call): do not call it under stream capture), _launch then runs them in ONE grid however many they are (the by-value
 * forms above are limited to 8 / 4 calls per launch by the 4 KiB of kernel arguments): with 16 members every workgroup walks 2 - 16
 * work items of its member instead of 1 - 4, and a stage is one launch instead of two or four.  max_samples = the largest n_samples
 * of the calls.  Same results as the by-value forms.
 * _prepare remembers, per dev_args pointer, what it baked into the blocks (n_calls, B, a_hi, the one w_layout of the call array, n_blocks /
 * with_stem); _launch answers QBNN_E_INVALID for a dev_args it did not prepare and for arguments that differ from those (they pick the
 * kernel: a mismatching w_layout would read the weights in the wrong fragment order).  n_calls may be smaller at launch (a prefix of the calls). */
size_t qbnn_chain_multi_args_bytes(int32_t n_calls, int32_t n_blocks);
size_t qbnn_down_multi_args_bytes(int32_t n_calls);
int qbnn_block_chain_i8_multi_prepare(const qbnn_chain_call* calls, int32_t n_calls, int32_t with_stem, int32_t B, int32_t a_hi,
                                      int32_t n_blocks, void* dev_args, void* stream);
int qbnn_block_chain_i8_multi_launch(const void* dev_args, int32_t n_calls, int32_t with_stem, int32_t B, int32_t H, int32_t C,
                                     int32_t a_hi, int32_t w_layout, int32_t n_blocks, int32_t max_samples, void* stream);
                                     /* a_hi, w_layout: as given to _prepare (w_layout: the calls' blocks[0].w_layout -- one layout per launch) */
int qbnn_block_down_i8_multi_prepare(const qbnn_down_call* calls, int32_t n_calls, int32_t B, int32_t a_hi, void* dev_args, void* stream);
int qbnn_block_down_i8_multi_launch(const void* dev_args, int32_t n_calls, int32_t B, int32_t H, int32_t Cin, int32_t w_layout,
                                    int32_t max_samples, void* stream);      /* w_layout: the calls' desc->blk.w_layout (one per launch) */

/* QuantStub + clamp_activation + layer-0 patch gather (qbnn_quantize_input_nchw followed by qbnn_im2col3x3_c3) for n input
 * quantisations at once: x fp32 NCHW [B][3][H][W] -> out[m][B][H*W][32] centred int8 patches, m < n (host arrays scales / zero_points). */
int qbnn_quantize_im2col3x3_c3_multi(const float* x, int32_t B, int32_t H, int32_t W, const float* scales, const int32_t* zero_points,
                                     int32_t n, int32_t a_hi, int8_t* out, int64_t out_stride, void* stream);

/* QuantStub + clamp_activation (models_bbb.py:227-229): fp32 NCHW -> uint8 NHWC. */
int qbnn_quantize_input_nchw(const float* x, int32_t B, int32_t C, int32_t H, int32_t W, float scale,
                             int32_t zero_point, int32_t a_hi, uint8_t* out, void* stream);

/* Layer-0 helper: 3x3/pad-1 patches of a [B][H][W][3] uint8 image, centred by z_x, as int8
 * [B][H*W][32] (27 taps in (kh,kw,c) order + 5 zero bytes).  Shared by all MC samples. */
int qbnn_im2col3x3_c3(const uint8_t* x, int32_t B, int32_t H, int32_t W, int32_t z_x, int8_t* out, void* stream);

/* Network head for S samples: AvgPool2d(k) -> clamp -> Flatten -> int8 Linear -> clamp ->
 * DeQuantStub -> softmax     (models_bbb.py:209-211, :240-243; linear_q.py:80-94).
 *   x [S][B][k][k][C] uint8, w_rowmajor [S][N][C] int8 (QBNN_LAYOUT_ROWMAJOR), probs [S][B][N] fp32.
 *   C <= 256, N <= 16 (one wave per image-sample, four lanes per class); sample strides of 0 share an operand. */
typedef struct qbnn_head_desc {
  int32_t B, k, C, N;
  float s_x; int32_t z_x;
  float s_w; int32_t z_w;
  float s_y; int32_t z_y;
  int32_t a_hi;
  int32_t has_bias;
} qbnn_head_desc;

int qbnn_head_i8_mc(const uint8_t* x, int64_t x_sample_stride, const int8_t* w_rowmajor, int64_t w_sample_stride,
                    const float* bias, float* probs, int32_t n_samples, const qbnn_head_desc* host_desc, void* stream);

typedef struct qbnn_head_call {
  const uint8_t* x; int64_t x_sample_stride; const int8_t* w; int64_t w_sample_stride; const float* bias; float* probs;
  int32_t n_samples; const qbnn_head_desc* desc;
} qbnn_head_call;
/* n_calls heads side by side (see qbnn_block_chain_i8_multi); the calls of one array evaluate the same number of samples. */
int qbnn_head_i8_multi(const qbnn_head_call* calls, int32_t n_calls, void* stream);

/* MC reduction (experiments/utils.py:342-355): sum over the S per-sample outputs of p and p*p, in sample order
 * (deterministic), kept in fp64 (the variance below cancels in fp32):  moments[0][n] (+)= sum_s p, moments[1][n] (+)= sum_s p^2.
 * finalize_total > 0 (single rank, last chunk): the same launch also writes mean = sum / total and, if var_out != NULL,
 * the unbiased variance (sum2 - sum^2 / total) / (total - 1) of experiments/utils.py:352 as fp32. */
int qbnn_reduce_moments(const float* probs, int32_t n_samples, int64_t n, int32_t accumulate, double* moments,
                        int32_t finalize_total, float* mean_out, float* var_out, void* stream);

/* After the cross-GPU sum of the fp64 moments: mean / unbiased variance as above. */
int qbnn_finalize_moments(const double* moments, int64_t n, int32_t total_samples, float* mean_out, float* var_out, void* stream);

/* ---- MC-Dropout path (BASELINE config 2: LeNet; reference src/models/stochastic/mcdropout/) ----------------------- */

/* Deterministic int8 conv / linear of any geometry (standard torch.nn.quantized Conv2d / Linear(ReLU) as produced by
 * quant_utils.convert for the MC-Dropout nets, models_mc.py:83-93), for S samples, + clamp_activation.
 * w_ohwi: int8 [Cout][KH][KW][Cin] row-major; w_sample_stride 0 = shared by all samples.  A linear layer is the 1x1 case
 * (H = W = 1, Cin = in_features). */
int qbnn_conv2d_i8_generic_mc(const uint8_t* x, int64_t x_sample_stride, const int8_t* w_ohwi, int64_t w_sample_stride,
                              const float* bias, uint8_t* y, int64_t y_sample_stride, int32_t n_samples,
                              const qbnn_conv_desc* host_desc, void* stream);

/* The same contract on the one-thread-per-output scalar kernel (the first implementation): kept as the on-device
 * cross-check of the MFMA form -- tests require bit-identical outputs -- and selected for every call by
 * QBNN_GENERIC_NAIVE=1. */
int qbnn_conv2d_i8_generic_scalar_mc(const uint8_t* x, int64_t x_sample_stride, const int8_t* w_ohwi, int64_t w_sample_stride,
                              const float* bias, uint8_t* y, int64_t y_sample_stride, int32_t n_samples,
                              const qbnn_conv_desc* host_desc, void* stream);

/* Quantised BernoulliDropout.forward (mcdropout/dropout.py:15-40) + clamp_activation, x [S][B][HW][C] channels-last:
 *   mask ~ Bernoulli(keep_prob), one draw per (sample, b, c) -- per channel for 4-D inputs, per element when HW == 1 --
 *   from the Philox uniform stream  philox4x32_10(ctr = {i >> 2, layer_id, sample_begin + s, 1}, key = seed)[i & 3] >> 8,
 *   i = b * C + c, keep iff u * 2^-24 < keep_prob;   mask_in != NULL: fp32 [S][B][C] masks are used instead (parity mode);
 *   mask_q = quantize_per_tensor(mask, s_m, z_m, quint8); y = quantized::mul(x, mask_q) with output (s_m, z_m).
 * The following mul_scalar(., 1/(1-p)) leaves the integers untouched: the CALLER multiplies the scale. */
int qbnn_dropout_q_mc(const uint8_t* x, int64_t x_sample_stride, int32_t B, int32_t HW, int32_t C, float keep_prob, float s_x,
                      int32_t z_x, float s_m, int32_t z_m, int32_t a_hi, uint64_t seed, uint32_t layer_id, uint32_t sample_begin,
                      const float* mask_in, uint8_t* y, int64_t y_sample_stride, int32_t n_samples, void* stream);

/* The small networks' layers on the matrix pipe (LeNet / MLP graphs: mcdropout/models_mc.py:75-111, bbb/models_bbb.py), weights in
 * the QBNN_LAYOUT_MFMA32 fragment layout (fixed: sample stride 0; sampled: the sampler's output).  Same arithmetic contract -- and the
 * same bits -- as qbnn_conv2d_i8_generic_mc -> qbnn_maxpool2_q_mc -> qbnn_dropout_q_mc.  Activations <= 7 bit, zero points in [0,127]. */
typedef struct qbnn_dropout_desc {
  float keep_prob; float s_m; int32_t z_m;   /* 1 - p; mul_mask.scale / .zero_point                       */
  uint32_t layer_id;                         /* Philox tensor id of the dropout (mask index b * C + c)     */
} qbnn_dropout_desc;

/* conv (k x k, stride 1, pad (k-1)/2, weights packed with krow = k * Cin) -> clamp [-> MaxPool2d(2,2)] [-> BernoulliDropout, one draw per
 * (sample, image, channel)] -> Flatten in NHWC order: y [S][B][ldy], ldy % 16 == 0, bytes beyond the map written 0.  drop == NULL: no
 * dropout; mask_in fp32 [S][B][Cout]: parity mode.  drop_in != NULL: a BernoulliDropout (one draw per (sample, image, INPUT channel);
 * mask_in_in fp32 [S][B][Cin]) is applied to x -- qparams (s_in, z_in), typically shared by the samples -- while it enters LDS; the
 * desc's (s_x, z_x) are then that dropout's output qparams (s_m / (1 - p), z_m).  Built geometry: 14 x 14, 20 -> 50, 5 x 5 (LeNet layers.3). */
int qbnn_conv_pool_drop_i8_mc(const uint8_t* x, int64_t x_sample_stride, const int8_t* w_packed, int64_t w_sample_stride,
                              const float* bias, uint8_t* y, int64_t y_sample_stride, int32_t ldy, int32_t n_samples,
                              const qbnn_conv_desc* host_desc, int32_t pool, const qbnn_dropout_desc* drop, const float* mask_in,
                              const qbnn_dropout_desc* drop_in, const float* mask_in_in, float s_in, int32_t z_in,
                              uint64_t seed, uint32_t sample_begin, void* stream);

/* LeNet's first conv with sampled weights (bbb/models_bbb.py:120-133 layers.0 -> layers.1): k x k conv on a ONE-channel image -> clamp ->
 * MaxPool2d(2,2) in one launch.  qbnn_im2col5x5_c1 builds the centred 25-tap patches [B][H * W][32] (int8, taps in (kh, kw) order, 7 zero
 * bytes) once per batch; the conv is then one MFMA per 32 pixels against the sample's fragment tile (weights packed / sampled with
 * krow = k = 25: 1 KiB per sample) and only the pooled map y [S][B][H/2][W/2][Cout] is written.  Same bits as
 * qbnn_conv2d_i8_generic_mc -> qbnn_maxpool2_q_mc.  Built geometry: 28 x 28, 1 -> 20, 5 x 5, pad 2; desc->relu must be 0. */
int qbnn_im2col5x5_c1(const uint8_t* x, int32_t B, int32_t H, int32_t W, int32_t z_x, int8_t* out, void* stream);
int qbnn_conv_c1_pool_i8_mc(const int8_t* patches, int64_t patches_sample_stride, const int8_t* w_packed, int64_t w_sample_stride,
                            const float* bias, uint8_t* y, int64_t y_sample_stride, int32_t n_samples, const qbnn_conv_desc* host_desc,
                            void* stream);

/* Linear / LinearReLU (desc: B rows, Cin = K, Cout = N, H = W = ksize = stride = 1; weights packed with krow = k = K) -> clamp
 * [-> BernoulliDropout on the 2-D activation: one draw per element, index b * N + n]: x [S|1][B][ldx] (ldx % 16 == 0, bytes K..ldx-1
 * ignored), y [S][B][ldy] (ldy % 4 == 0, bytes N..ldy-1 written 0; or ldy == N: dense rows).  mask_in fp32 [S][B][N]: parity mode. */
int qbnn_linear_i8_mc(const uint8_t* x, int64_t x_sample_stride, int32_t ldx, const int8_t* w_packed, int64_t w_sample_stride,
                      const float* bias, uint8_t* y, int64_t y_sample_stride, int32_t ldy, int32_t n_samples,
                      const qbnn_conv_desc* host_desc, const qbnn_dropout_desc* drop, const float* mask_in, uint64_t seed,
                      uint32_t sample_begin, void* stream);

/* nn.MaxPool2d(2,2) on quint8 (keeps scale / zero point) + clamp_activation; x [S][B][H][W][C]. */
int qbnn_maxpool2_q_mc(const uint8_t* x, int64_t x_sample_stride, int32_t B, int32_t H, int32_t W, int32_t C, int32_t a_hi,
                       uint8_t* y, int64_t y_sample_stride, int32_t n_samples, void* stream);

/* Flatten (src/utils.py:40-47) of a channels-last map into the reference's NCHW feature order: x [S][B][HW][C] ->
 * y [S][B][C*HW].  Used in front of a stochastic Linear (its noise stream follows the reference's column order). */
int qbnn_flatten_nchw_mc(const uint8_t* x, int64_t x_sample_stride, int32_t B, int32_t HW, int32_t C, uint8_t* y,
                         int64_t y_sample_stride, int32_t n_samples, void* stream);

/* The same on pitched rows: x [S][B][ldx] (NHWC order in the first HW * C bytes, as qbnn_conv_pool_drop_i8_mc writes them) ->
 * y [S][B][ldy] in NCHW order, bytes HW * C .. ldy - 1 written 0 (what qbnn_linear_i8_mc reads). */
int qbnn_flatten_nchw_rows_mc(const uint8_t* x, int64_t x_sample_stride, int32_t ldx, int32_t B, int32_t HW, int32_t C, uint8_t* y,
                              int64_t y_sample_stride, int32_t ldy, int32_t n_samples, void* stream);

/* DeQuantStub + F.softmax(dim=-1): x [S][B][N] uint8 -> probs [S][B][N] fp32. */
int qbnn_dequant_softmax_mc(const uint8_t* x, int64_t x_sample_stride, int32_t B, int32_t N, float scale, int32_t zero_point,
                            float* probs, int32_t n_samples, void* stream);

/* ---- fp32 Bayes-by-backprop path (BASELINE config 0; reference src/models/stochastic/bbb/linear.py:42-50) ---------- */

/* W[s][i] = mu[i] + eps(s,i) * sigma[i]  (sigma = softplus(rho), computed once by the host), eps from the Philox normal
 * stream {ctr = {i >> 2, layer_id, sample_begin + s, 0}}[i & 3] or eps_in [S][n] (parity mode). */
int qbnn_sample_weights_f32(const float* mu, const float* sigma, int64_t n, uint64_t seed, uint32_t layer_id,
                            uint32_t sample_begin, int32_t n_samples, const float* eps_in, float* w_out, void* stream);

/* y[s][b][n] = act( sum_k x[s][b][k] * w[s][n][k] + bias[n] ),  act: 0 none, 1 ReLU, 2 exp (the log_var head,
 * models_bbb.py:78).  Sample strides in elements; 0 = shared. */
int qbnn_linear_f32_mc(const float* x, int64_t x_sample_stride, const float* w, int64_t w_sample_stride, const float* bias,
                       float* y, int64_t y_sample_stride, int32_t B, int32_t K, int32_t N, int32_t act, int32_t n_samples,
                       void* stream);

/* quantized::add of two quint8 tensors [S|1][n] (+ clamp_activation, ReLU = max(q, z_o), clamp_activation): the `Add` + `end`
 * of a BasicBlock (src/utils.py:49-55, mcdropout/models_mc.py:156-160) when it cannot be fused into the preceding conv. */
int qbnn_add_relu_q_mc(const uint8_t* a, int64_t a_sample_stride, float s_a, int32_t z_a, const uint8_t* b, int64_t b_sample_stride,
                       float s_b, int32_t z_b, uint8_t* y, int64_t y_sample_stride, int64_t n, float s_o, int32_t z_o, int32_t a_hi,
                       int32_t relu, int32_t n_samples, void* stream);

/* ---- fp32 convolutional graphs (row a1: reference bbb/conv.py:33-39 eval branch, models_bbb.py:100-245) -------------- */

/* Z_s = conv2d(X_s, W_s) (+ bias) (ReLU).  x [S|1][B][H][W][Cin] fp32 NHWC, w [S|1][Cout][Cin][k][k] in the REFERENCE's
 * weight order (so qbnn_sample_weights_f32's noise index is the reference's element index), y [S][B][Ho][Wo][Cout].
 * Replaces F.conv2d at bbb/conv.py:38 / conv_qat.py:47,158.  Sample strides in elements; 0 = shared.
 * `relu` is a flag word: bit 0 = fused ReLU, bit 1 = accumulate in fp64 (one rounding of the exact sum; the QAT path
 * uses it because a fake-quantiser follows every conv), bit 2 = w is stored [Cout][k][k][Cin] (qbnn_sample_weights_f32_ohwi). */
int qbnn_conv2d_f32_mc(const float* x, int64_t x_sample_stride, const float* w, int64_t w_sample_stride, const float* bias, float* y,
                       int64_t y_sample_stride, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize,
                       int32_t stride, int32_t pad, int32_t relu, int32_t n_samples, void* stream);

/* qbnn_conv2d_f32_mc with the float graphs' tail fused: v = conv; v /= div[n] (conv_qat.py:159, Z / scale_factor); v += bias[n];
 * v *= alpha[n]; v += beta[n] (nn.BatchNorm2d in eval, alpha = weight / sqrt(var + eps), beta = bias - mean alpha);
 * v += res (Add, res [S|1][B][Ho][Wo][Cout]); ReLU (flags bit 0).
 * Every step is rounded to fp32 as the stand-alone kernels round it; NULL pointers skip a step.
 * Replaces conv -> bn -> (add) -> relu of models_bbb.py:154-183 for the un-prepared float model, and conv -> / c + b -> bn -> relu
 * of conv_qat.py:150-165 for the prepared one.  With flags bit 1 (fp64 accumulation) and Cin % 4 == 0, OHWI weights, the sum
 * runs on v_mfma_f64_16x16x4_f64. */
int qbnn_conv2d_f32_fused_mc(const float* x, int64_t x_sample_stride, const float* w, int64_t w_sample_stride, const float* div,
                             const float* bias, const float* alpha, const float* beta, const float* res, int64_t res_sample_stride, float* y,
                             int64_t y_sample_stride, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize,
                             int32_t stride, int32_t pad, int32_t relu, int32_t n_samples, float* minmax_partials, void* stream);

/* Observer fusion for the QAT path: with `minmax_partials` ([S][qbnn_conv2d_f32_blocks(...)][2] floats) non-NULL every workgroup
 * of the conv writes the (min, max) of the outputs it produced, and qbnn_observe_partials_f32_mc runs the
 * MovingAverageMinMaxObserver recurrence of qbnn_observe_f32_mc on them -- the conv output is not read a second time. */
int32_t qbnn_conv2d_f32_blocks(int32_t B, int32_t H, int32_t W, int32_t Cout, int32_t ksize, int32_t stride, int32_t pad);
int qbnn_observe_partials_f32_mc(const float* partials, int32_t n_blocks, int32_t n_samples, float* state, float avg_const,
                                 int32_t qmin, int32_t qmax, float* scale, int32_t* zero_point, void* stream);

/* QAT convs on the int8 matrix pipe (round 5).  In the prepared model both operands of conv_qat.py:150-158's conv are fake-quantised tensors --
 * integers on a per-sample grid: X = (q_x - z_x) s_x[s] with |q_x - z_x| <= 127 (ReLU and max-pooling keep the grid), W = (q_w - z_w) s_w[s] with q_w an
 * int8 -- so conv(X, W) = s_x s_w sum (q_x - z_x)(q_w - z_w): an exact integer sum (v_mfma_i32_32x32x32_i8; the weight zero point through the window
 * sum), scaled once in fp64 and rounded to fp32, then the same fused tail as qbnn_conv2d_f32_fused_mc (Z / c, + bias, bn, ReLU; min / max partials of
 * qbnn_conv2d_q8_blocks(...) workgroups per sample for the observer).  Against the fp64 sum of the fp32-rounded operands it differs by the operands'
 * own rounding (<= 1.2e-7 relative).
 *   qbnn_grid_to_i8_mc: out[s][i] = clamp(rne(x[s][i] / scale[s]) + (zero_point ? zero_point[s] : 0), -128, 127) -- the centred activation integer
 *   (zero_point NULL) or the raw weight integer q_w;  x_sample_stride 0 shares x.
 *   qbnn_conv2d_q8_f32_mc: x int8 [S][B][H][W][Cin] centred, w int8 [S][Cout][k][k][Cin] raw, s_x / s_w / z_w per sample; y fp32 [S][B][Ho][Wo][Cout]. */
/* qbnn_fake_quant_f32_mc with the ReLU that follows it in the graph (BasicBlock: Add -> FakeQuantize -> ReLU) and, optionally, the grid integers
 * q - z (after the ReLU) as int8 [S][n]: the activation operand of qbnn_conv2d_q8_f32_mc without a qbnn_grid_to_i8_mc pass (qmax - qmin <= 127, as q - z
 * spans +-(qmax - qmin); QBNN_E_INVALID for a wider grid).  y may be NULL when q8_out is given (round 6): where every consumer takes the grid
 * integers -- a conv on the int8 pipe, qbnn_add_q8_f32_mc -- the fp32 tensor is never written. */
int qbnn_fake_quant_ex_f32_mc(const float* x, int64_t x_sample_stride, float* y, int64_t y_sample_stride, int64_t n, const float* scale,
                              const int32_t* zero_point, int32_t qmin, int32_t qmax, int32_t relu, int8_t* q8_out, int32_t n_samples, void* stream);
int qbnn_grid_to_i8_mc(const float* x, int64_t x_sample_stride, int64_t n, const float* scale, const int32_t* zero_point, int8_t* out,
                       int32_t n_samples, void* stream);
int32_t qbnn_conv2d_q8_blocks(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride, int32_t pad);
int qbnn_conv2d_q8_f32_mc(const int8_t* x, int64_t x_sample_stride, const int8_t* w, int64_t w_sample_stride, const float* s_x, const float* s_w,
                          const int32_t* z_w, const float* div, const float* bias, const float* alpha, const float* beta, float* y,
                          int64_t y_sample_stride, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride,
                          int32_t pad, int32_t relu, int32_t n_samples, float* minmax_partials, void* stream);

/* FloatFunctional.add (src/utils.py:49-55 `Add`; models_bbb.py:178 out + shortcut) of two fake-quantised tensors given as grid integers (round 6):
 * y[s][i] = fl32((float)a[s][i] * s_a[s]) + fl32((float)b[s][i] * s_b[s]) -- each addend is the fp32 value its FakeQuantize would have written, so
 * the sum equals the fp32 Add bit for bit --, plus the (min, max) of each of the qbnn_add_q8_blocks(n) workgroups' sums per sample for the Add's observer
 * (minmax_partials [S][blocks][2], may be NULL; y may be NULL when the partials are asked for).  a / b sample stride 0 shares the operand. */
int32_t qbnn_add_q8_blocks(int64_t n);
int qbnn_add_q8_f32_mc(const int8_t* a, int64_t a_sample_stride, const float* s_a, const int8_t* b, int64_t b_sample_stride, const float* s_b, float* y,
                       int64_t y_sample_stride, int64_t n, int32_t n_samples, float* minmax_partials, void* stream);
/* ... and the Add's own FakeQuantize (+ ReLU) from the same two operands: with y = NULL above (partials only) and this call after the observer scan, the fp32
 * sum is never written nor read -- x = fl32((float)a s_a) + fl32((float)b s_b) is recomputed per element, then qbnn_fake_quant_ex_f32_mc's arithmetic and
 * outputs (fp32 y and / or the grid integers q8_out, either may be NULL but not both). */
int qbnn_fake_quant_add_q8_mc(const int8_t* a, int64_t a_sample_stride, const float* s_a, const int8_t* b, int64_t b_sample_stride, const float* s_b, float* y,
                              int64_t y_sample_stride, int64_t n, const float* scale, const int32_t* zero_point, int32_t qmin, int32_t qmax, int32_t relu,
                              int8_t* q8_out, int32_t n_samples, void* stream);

/* The float Bayes-by-backprop draw of ALL layers and MC samples in one launch (round 6; bbb/conv.py:33-39, bbb/linear.py:42-50): per layer
 * W[s] = mu + eps_s * sigma, bit-identical to qbnn_sample_weights_f32_ohwi (KS > 0: mu / sigma in the reference's [Cout][Cin][kh][kw] order, W written
 * [Cout][kh][kw][Cin]) / qbnn_sample_weights_f32 (KS = 0).  `dev_layers`: descriptors IN DEVICE MEMORY (device pointers; w = [n_samples][n] output;
 * [blk0, blk0 + nblk) the layer's workgroups, consecutive layers, total_blocks in all). */
typedef struct qbnn_f32_wlayer {
  const float* mu; const float* sigma; float* w;
  int32_t n, Cout, Cin, KS;
  uint32_t layer_id;
  int32_t blk0, nblk;
} qbnn_f32_wlayer;
int qbnn_sample_weights_f32_batch(const qbnn_f32_wlayer* dev_layers, int32_t n_layers, int32_t total_blocks, uint64_t seed, uint32_t sample_begin,
                                  int32_t n_samples, void* stream);

/* The QAT weight pipelines of all stochastic layers at once (round 6; conv_qat.py:26-49, linear_qat.py:18-41 in eval): per layer and MC sample
 *   w = FQ_w(mu c), s = FQ_s(softplus(rho) c), t = FQ_m(eps * s), W = FQ_a(w + t)      (FQ_x: live MovingAverageMinMax observer + fake_quantize)
 * in four launches instead of ~15 per layer; bit-identical to the per-layer calls (qbnn_observe_f32_mc, qbnn_fake_quant_f32_mc,
 * qbnn_sample_weights_f32_ohwi / _strided, qbnn_affine_f32_mc, qbnn_grid_to_i8_mc).  `dev_layers`: n_layers descriptors IN DEVICE MEMORY, every pointer
 * a device pointer: mu = mu c in the output element order ([Cout][kh][kw][Cin] for a conv: KS > 0; the reference's order for a linear: KS = 0), sg =
 * softplus(rho) c in the reference's element order (the noise stream's index), st_* the four observers' (min, max, seen) states, cmm = (min mu, max mu,
 * min sg, max sg), [blk0, blk0 + nblk) the layer's workgroups in the launch (nblk <= 64, consecutive layers, total_blocks in all), pm / pa workspaces of
 * n_samples * nblk * 2 floats, outputs W fp32 [S][n], its raw integers q8 [S][n] and FQ_a's per-sample (scale, zero point).  1 <= n_samples <= 64. */
typedef struct qbnn_qat_wlayer {
  const float* mu; const float* sg;
  float* st_w; float* st_s; float* st_m; float* st_a;
  float cmm[4];
  int32_t n, Cout, Cin, KS;
  uint32_t layer_id;
  int32_t qmin, qmax;
  int32_t blk0, nblk;
  float* pm; float* pa;
  float* W; int8_t* q8; float* scale; int32_t* zp;
} qbnn_qat_wlayer;
int qbnn_qat_weights_mc(const qbnn_qat_wlayer* dev_layers, int32_t n_layers, int32_t total_blocks, float avg_const, uint64_t seed,
                        uint32_t sample_begin, int32_t n_samples, void* stream);

/* Pointwise on [S][n] with the channel as fastest axis:  v = (mode 0) x * p0[c] + p1[c]  |  (mode 1) x / p0[c] + p1[c]
 * (p0 / p1 NULL skip that step), v += res (if given), ReLU (if asked).  nn.BatchNorm2d in eval (x * alpha + beta as ATen
 * computes it), the `Z / scale_factor + bias` of conv_qat.py:159-161, Add (src/utils.py:49-55), nn.ReLU. */
int qbnn_affine_f32_mc(const float* x, int64_t x_sample_stride, const float* res, int64_t res_sample_stride, const float* p0,
                       const float* p1, float* y, int64_t y_sample_stride, int64_t n, int32_t C, int32_t mode, int32_t relu,
                       int32_t n_samples, void* stream);

/* ---- float MC-Dropout (row a6+/a7: reference mcdropout/dropout.py:15-40 with FloatFunctional, models_mc.py graphs with q=False) --- */

/* The Bernoulli(keep_prob) mask of a float BernoulliDropout as fp32 0 / 1, mask_out [S][n_slots]: slot = b * C + c for a 4-D
 * activation (dropout.py:24-29: whole channels), the element index for a 2-D one (:19-23).  Same Philox uniform stream as
 * qbnn_dropout_q_mc ({ctr = {i >> 2, layer_id, sample_begin + s, 1}}[i & 3] < keep_prob); honours qbnn_set_device_noise_source. */
int qbnn_dropout_mask_f32_mc(int64_t n_slots, float keep_prob, uint64_t seed, uint32_t layer_id, uint32_t sample_begin, int32_t n_samples,
                             float* mask_out, void* stream);

/* y = (x * mask[s][b][c]) * multiplier  (+ res) (ReLU)  on x [S|1][B][HW][C]: mul_mask.mul then mul_scalar.mul_scalar
 * (dropout.py:38-39), each rounded to fp32; res / ReLU = the Add + `end` of a BasicBlock behind its last dropout (models_mc.py:156-159).
 * mask [S][B][C] is caller-provided (qbnn_dropout_mask_f32_mc, or injected in parity tests). */
int qbnn_dropout_f32_mc(const float* x, int64_t x_sample_stride, const float* mask, int32_t B, int32_t HW, int32_t C, float multiplier,
                        const float* res, int64_t res_sample_stride, int32_t relu, float* y, int64_t y_sample_stride, int32_t n_samples,
                        void* stream);

/* qbnn_conv2d_f32_fused_mc with a float BernoulliDropout in the tail: v = conv; v += bias[n]; v = v * alpha[n] + beta[n] (BatchNorm
 * eval); v = (v * drop_mask[s][b][n]) * drop_mult; v += res; ReLU (flags bit 0; a ReLU that the graph places in front of the dropout
 * commutes with it exactly: mask and multiplier are >= 0).  conv -> bn -> relu -> dropout and conv -> bn -> dropout -> Add -> relu of
 * models_mc.py:124-160 in one launch each.  drop_mask NULL = no dropout. */
int qbnn_conv2d_f32_drop_mc(const float* x, int64_t x_sample_stride, const float* w, int64_t w_sample_stride, const float* bias,
                            const float* alpha, const float* beta, const float* drop_mask, float drop_mult, const float* res,
                            int64_t res_sample_stride, float* y, int64_t y_sample_stride, int32_t B, int32_t H, int32_t W, int32_t Cin,
                            int32_t Cout, int32_t ksize, int32_t stride, int32_t pad, int32_t relu, int32_t n_samples, void* stream);

/* nn.MaxPool2d(k, k) (mode 0) / nn.AvgPool2d(k) (mode 1) on NHWC fp32. */
int qbnn_pool2d_f32_mc(const float* x, int64_t x_sample_stride, float* y, int64_t y_sample_stride, int32_t B, int32_t H, int32_t W,
                       int32_t C, int32_t k, int32_t mode, int32_t n_samples, void* stream);

/* Flatten (src/utils.py:40-47) of an NHWC activation in NCHW order: y[s][b][c * HW + p] = x[s][b][p][c]. */
int qbnn_flatten_nchw_f32_mc(const float* x, int64_t x_sample_stride, int32_t B, int32_t HW, int32_t C, float* y,
                             int64_t y_sample_stride, int32_t n_samples, void* stream);

/* F.softmax(dim=-1) on fp32 logits [S][B][N] -> probs [S][B][N]. */
int qbnn_softmax_f32_mc(const float* x, int64_t x_sample_stride, int32_t B, int32_t N, float* probs, int32_t n_samples, void* stream);

/* ---- QAT fake-quant evaluation with live observers (row a2: conv_qat.py:26-49,139-167, linear_qat.py:18-41) ---------- */

#define QBNN_OBSERVER_BLOCKS 512        /* workspace: n_samples * QBNN_OBSERVER_BLOCKS * 2 floats */

/* The whole float Bayes-by-backprop MLP of reference models_bbb.LinearNetwork (:32-78, eval branch of bbb/linear.py:42-50) for all S
 * samples in two launches: layers[0..2] = Linear + ReLU, layers[3] = `mu` head, layers[4] = `log_var` head (one output each);
 *   mu_out[s][b], var_out[s][b] = exp(log_var).  Weights are drawn as qbnn_sample_weights_f32 draws them (same Philox stream per
 * layer_id; honours qbnn_set_device_noise_source) into w_workspace[n_samples][qbnn_mlp_bbb_f32_workspace_floats(layers)].
 * Widths <= 128.  Equals qbnn_sample_weights_f32 + qbnn_linear_f32_mc per layer up to the fp32 summation order of the dot products. */
typedef struct qbnn_mlp_layer {
  const float* mu; const float* sigma; const float* bias;      /* [out][in], [out][in] (softplus(rho)), [out] or NULL */
  int32_t out_features, in_features;
  uint32_t layer_id;
} qbnn_mlp_layer;
int qbnn_mlp_bbb_f32_mc(const float* x, int32_t B, const qbnn_mlp_layer* layers, uint64_t seed, uint32_t sample_begin, int32_t n_samples,
                        float* w_workspace, float* mu_out, float* var_out, void* stream);
int64_t qbnn_mlp_bbb_f32_workspace_floats(const qbnn_mlp_layer* layers);

/* W[s][i] = mu[s][i] + eps(s,i) * sigma[s][i] with per-sample operands (strides in elements, 0 = shared); mu NULL gives
 * the noise term alone (conv_qat.py:45 / linear_qat.py:33: mul_noise.mul(noise, std)).  Noise stream as qbnn_sample_weights_f32. */
int qbnn_sample_weights_f32_strided(const float* mu, int64_t mu_sample_stride, const float* sigma, int64_t sigma_sample_stride,
                                    int64_t n, uint64_t seed, uint32_t layer_id, uint32_t sample_begin, int32_t n_samples,
                                    const float* eps_in, float* w_out, void* stream);

/* Conv-weight form of the sampler above: same noise stream (indexed by the reference's [Cout][Cin][k][k] element order),
 * output written [Cout][k][k][Cin] so that the K axis of the implicit GEMM is contiguous for qbnn_conv2d_f32_mc (flags bit 2). */
int qbnn_sample_weights_f32_ohwi(const float* mu, int64_t mu_sample_stride, const float* sigma, int64_t sigma_sample_stride, int32_t Cout,
                                 int32_t Cin, int32_t ksize, uint64_t seed, uint32_t layer_id, uint32_t sample_begin, int32_t n_samples,
                                 const float* eps_in, float* w_out, void* stream);

/* MovingAverageMinMaxObserver (averaging constant `avg_const`, active in eval) over the S samples IN ORDER, then
 * calculate_qparams (per-tensor affine, quant range [qmin, qmax]):  state = {min, max, seen?} is read and written
 * back; scale[s], zero_point[s] are the qparams fake-quant uses for sample s (observer update precedes the
 * quantisation inside FakeQuantize.forward). */
int qbnn_observe_f32_mc(const float* x, int64_t x_sample_stride, int64_t n, int32_t n_samples, float* state, float avg_const,
                        int32_t qmin, int32_t qmax, float* workspace, float* scale, int32_t* zero_point, void* stream);

/* fake_quantize_per_tensor_affine: y = (clamp(rne(x * (1/s)) + z, qmin, qmax) - z) * s with the qparams of sample s at
 * index s * qparam_stride (stride 0 = one pair for all samples). */
int qbnn_fake_quant_f32_mc(const float* x, int64_t x_sample_stride, float* y, int64_t y_sample_stride, int64_t n, const float* scale,
                           const int32_t* zero_point, int32_t qparam_stride, int32_t qmin, int32_t qmax, int32_t n_samples,
                           void* stream);

/* Classification metrics of a [B][C] predictive mean against int64 targets (reference src/metrics.py: Error :8-33,
 * ClassificationNegativeLogLikelihood :36-62, BrierScore :65-91, PredictiveEntropy :94-116, 10-bin L1 calibration error
 * :381-383).  Writes ceil(B/256) rows of 34 partial sums (layout: csrc/qbnn_kernels.hip); the caller adds the rows. */
int qbnn_classification_metrics(const float* probs, const int64_t* target, int32_t B, int32_t C, float* partials, void* stream);

/* Regression metrics on the reduced MC output (reference src/metrics.py:119-230, RegressionMetric.update :468-500):
 * per block of 256 rows, partials[blk][0..2] = sum of the Gaussian NLL 0.5 log(2 pi var + 1e-8) + (t - mean)^2 / (2 var + 1e-8),
 * of the squared error and of the absolute error.  var == NULL: variance 1 (a mean-only model, :154). */
int qbnn_regression_metrics(const float* mean, const float* var, const float* target, int32_t B, float* partials, void* stream);

const char* qbnn_last_error(void);
int qbnn_version(void);

#ifdef __cplusplus
}
#endif
#endif /* QBNN_H_ */
