// libqbnn_hip.so -- the small kernels around the conv path: stand-alone quantized::add, input quantisation and the layer-0
// patch gather, head (AvgPool -> Linear -> softmax), MC reduction, the any-geometry int8 conv and the MC-Dropout ops,
// the fp32 sampler / linear of the MLP config, metrics, Flatten.
#include "qbnn_host.h"

// Stand-alone quantized::add (+ clamp_activation, ReLU, clamp_activation) for graphs where something sits between the
// last conv of a block and its Add (MC-Dropout ResNet: mcdropout/models_mc.py:136-160).  Four elements per thread.
__global__ __launch_bounds__(256) void add_relu_q_kernel(const uint8_t* __restrict__ a, int64_t a_ss, const uint8_t* __restrict__ b,
                                                          int64_t b_ss, uint8_t* __restrict__ y, int64_t y_ss, int64_t n4, float s_a,
                                                          float nzs_a, float s_b, float nzs_b, float inv_s_o, int z_o, int a_hi, int relu) {
  const int s = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const uint32_t av = reinterpret_cast<const uint32_t*>(a + (int64_t)s * a_ss)[i];
    const uint32_t bv = reinterpret_cast<const uint32_t*>(b + (int64_t)s * b_ss)[i];
    uint32_t o = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o |= add_relu_one((av >> (8 * j)) & 0xffu, (bv >> (8 * j)) & 0xffu, s_a, nzs_a, s_b, nzs_b, inv_s_o, z_o, a_hi, relu) << (8 * j);
    }
    reinterpret_cast<uint32_t*>(y + (int64_t)s * y_ss)[i] = o;
  }
}

QBNN_EXPORT int qbnn_add_relu_q_mc(const uint8_t* a, int64_t a_ss, float s_a, int32_t z_a, const uint8_t* b, int64_t b_ss, float s_b,
                                   int32_t z_b, uint8_t* y, int64_t y_ss, int64_t n, float s_o, int32_t z_o, int32_t a_hi, int32_t relu,
                                   int32_t n_samples, void* stream) {
  if (!a || !b || !y || n <= 0 || (n & 3) || n_samples <= 0 || (a_ss & 3) || (b_ss & 3) || (y_ss & 3))
    return fail(QBNN_E_INVALID, "qbnn_add_relu_q_mc: bad argument (element counts and strides must be multiples of 4)%s");
  const int64_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(add_relu_q_kernel, dim3(blocks, n_samples), dim3(256), 0, (hipStream_t)stream, a, a_ss, b, b_ss, y, y_ss, n4, s_a,
                     (float)(-z_a) * s_a, s_b, (float)(-z_b) * s_b, 1.0f / s_o, z_o, a_hi, relu);
  return check_launch("qbnn_add_relu_q_mc");
}

// =====================================================================================
// Input quantisation, layer-0 im2col, head, MC reduction
// =====================================================================================
__global__ __launch_bounds__(256) void quantize_input_kernel(const float* __restrict__ x, int B, int Cc, int H, int W,
                                                             float inv, int z, int a_hi, uint8_t* __restrict__ out) {
  const int64_t n = (int64_t)B * Cc * H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    // i indexes the NHWC output
    const int c = (int)(i % Cc);
    int64_t t = i / Cc;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int b = (int)(t / H);
    const float v = x[(((int64_t)b * Cc + c) * H + h) * W + w];
    int q = min(max(z + rne_sat(v * inv), 0), 255);
    out[i] = (uint8_t)min(q, a_hi);
  }
}

QBNN_EXPORT int qbnn_quantize_input_nchw(const float* x, int32_t B, int32_t Cc, int32_t H, int32_t W, float scale,
                                         int32_t zp, int32_t a_hi, uint8_t* out, void* stream) {
  if (!x || !out || B <= 0 || Cc <= 0 || H <= 0 || W <= 0) return fail(QBNN_E_INVALID, "qbnn_quantize_input_nchw: bad argument%s");
  const int64_t n = (int64_t)B * Cc * H * W;
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(quantize_input_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, B, Cc, H, W,
                     1.0f / scale, zp, a_hi, out);
  return check_launch("qbnn_quantize_input_nchw");
}

__global__ __launch_bounds__(256) void im2col3x3_c3_kernel(const uint8_t* __restrict__ x, int B, int H, int W, int z_x,
                                                           int8_t* __restrict__ out) {
  const int64_t npix = (int64_t)B * H * W;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
    const int ow = (int)(p % W);
    const int oh = (int)((p / W) % H);
    const int64_t b = p / ((int64_t)W * H);
    uint32_t wds[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int8_t* by = reinterpret_cast<int8_t*>(wds);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = oh + kh - 1, iw = ow + kw - 1;
        const bool in = ih >= 0 && ih < H && iw >= 0 && iw < W;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          by[(kh * 3 + kw) * 3 + c] = in ? (int8_t)((int)x[((b * H + ih) * W + iw) * 3 + c] - z_x) : (int8_t)0;
      }
    v4i* o = reinterpret_cast<v4i*>(out + p * 32);
    o[0] = v4i{(int)wds[0], (int)wds[1], (int)wds[2], (int)wds[3]};
    o[1] = v4i{(int)wds[4], (int)wds[5], (int)wds[6], (int)wds[7]};
  }
}

QBNN_EXPORT int qbnn_im2col3x3_c3(const uint8_t* x, int32_t B, int32_t H, int32_t W, int32_t z_x, int8_t* out, void* stream) {
  if (!x || !out || B <= 0) return fail(QBNN_E_INVALID, "qbnn_im2col3x3_c3: bad argument%s");
  const int64_t n = (int64_t)B * H * W;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(im2col3x3_c3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, B, H, W, z_x, out);
  return check_launch("qbnn_im2col3x3_c3");
}

// QuantStub + clamp_activation + the layer-0 patch gather for SEVERAL input quantisations at once (ensemble members each own
// a `quant.scale / zero_point`): fp32 NCHW [B][3][H][W] -> centred int8 patches out[m][B][H*W][32], member m = blockIdx.y.
struct QuantIm2colArgs { float inv[16]; int z[16]; };
// One thread = one output pixel for ALL `n` members of the launch: the 27 fp32 taps are loaded once and quantised n times (a thread
// per (member, pixel) re-read them per member: 74 us for 16 members at B = 256, now bound by the 8 MB of patches written per member).
__global__ __launch_bounds__(256) void quantize_im2col3x3_c3_kernel(const float* __restrict__ x, int B, int H, int W, const QuantIm2colArgs q,
                                                                     int n, int a_hi, int8_t* __restrict__ out, int64_t out_stride) {
  const int64_t npix = (int64_t)B * H * W, plane = (int64_t)H * W;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
    const int ow = (int)(p % W);
    const int oh = (int)((p / W) % H);
    const int64_t b = p / plane;
    float f[27];
    uint32_t inside = 0u;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = oh + kh - 1, iw = ow + kw - 1;
        const bool in = ih >= 0 && ih < H && iw >= 0 && iw < W;
        inside |= (in ? 1u : 0u) << (kh * 3 + kw);
#pragma unroll
        for (int c = 0; c < 3; ++c) f[(kh * 3 + kw) * 3 + c] = in ? x[((b * 3 + c) * H + ih) * W + iw] : 0.f;
      }
#pragma unroll 1
    for (int m = 0; m < n; ++m) {
      const float inv = q.inv[m];
      const int z = q.z[m];
      uint32_t wds[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < 27; ++t) {
        int v = 0;
        if ((inside >> (t / 3)) & 1u) v = min(min(max(z + rne_sat(f[t] * inv), 0), 255), a_hi) - z;     // quantize_input_kernel, then im2col3x3_c3_kernel's centring
        wds[t >> 2] |= ((uint32_t)v & 0xffu) << (8 * (t & 3));
      }
      v4i* o = reinterpret_cast<v4i*>(out + (int64_t)m * out_stride + p * 32);
      o[0] = v4i{(int)wds[0], (int)wds[1], (int)wds[2], (int)wds[3]};
      o[1] = v4i{(int)wds[4], (int)wds[5], (int)wds[6], (int)wds[7]};
    }
  }
}

QBNN_EXPORT int qbnn_quantize_im2col3x3_c3_multi(const float* x, int32_t B, int32_t H, int32_t W, const float* scales, const int32_t* zero_points,
                                                 int32_t n, int32_t a_hi, int8_t* out, int64_t out_stride, void* stream) {
  if (!x || !scales || !zero_points || !out || B <= 0 || n <= 0) return fail(QBNN_E_INVALID, "qbnn_quantize_im2col3x3_c3_multi: bad argument%s");
  const int64_t npix = (int64_t)B * H * W;
  const int blocks = (int)((npix + 255) / 256 < 1024 ? (npix + 255) / 256 : 1024);
  for (int c0 = 0; c0 < n; c0 += 16) {
    const int k = n - c0 < 16 ? n - c0 : 16;
    QuantIm2colArgs q;
    memset(&q, 0, sizeof(q));
    for (int i = 0; i < k; ++i) {
      if (zero_points[c0 + i] < 0 || zero_points[c0 + i] > 127) return fail(QBNN_E_INVALID, "qbnn_quantize_im2col3x3_c3_multi: zero points must be in [0,127]%s");
      if ((a_hi < 255 ? a_hi : 255) - zero_points[c0 + i] > 127)      // the centred patch bytes are int8
        return fail(QBNN_E_INVALID, "qbnn_quantize_im2col3x3_c3_multi: a_hi - zero_point must be <= 127 (the patches are centred int8)%s");
      q.inv[i] = 1.0f / scales[c0 + i]; q.z[i] = zero_points[c0 + i];
    }
    hipLaunchKernelGGL(quantize_im2col3x3_c3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, B, H, W, q, k, a_hi,
                       out + (int64_t)c0 * out_stride, out_stride);
    if (int rc = check_launch("qbnn_quantize_im2col3x3_c3_multi")) return rc;
  }
  return QBNN_OK;
}

// head: one wave per (sample, image).  C <= 256 channels, N <= 16 classes (4 lanes per class).
struct HeadArgs {
  const uint8_t* x; int64_t x_ss;
  const int8_t* w; int64_t w_ss;
  const float* bias;
  float* probs;
  int B, kk, C, N;
  int z_x, z_w, z_y, a_hi;
  float inv_kk, rcp, mult, s_y;
};

#define QBNN_HEAD_IMGS 1            // images per wave: the sample's Linear weights are loaded once for all of them
template <int NM = 1>
__global__ __launch_bounds__(256) void head_i8_kernel(const ArgsArr<HeadArgs, NM> all) {
  const HeadArgs& a = all.m[NM == 1 ? 0 : blockIdx.z];
  __shared__ int pooled[4][256];
  __shared__ __attribute__((aligned(16))) uint8_t stage[4][4096];      // an image's kk x C bytes, fetched with 16-byte loads (see below)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = blockIdx.y;
  const int b0 = (blockIdx.x * 4 + wave) * QBNN_HEAD_IMGS;
  if (b0 >= a.B) return;
  const int8_t* ws = a.w + (int64_t)s * a.w_ss;
  const int n = lane >> 2, j = lane & 3;
  // packed form: C a multiple of 16 and dword-aligned weights -- pooled activations as int8 dwords, v_dot4 against the raw weight
  // dwords (kept in registers for the wave's images), the weights' zero point through the channel sum:  sum p (w - z_w) = p.w - z_w sum p
  // (the pooled value q - z_x must fit an int8 for the signed dot product: z_x <= 128 and a_hi - z_x <= 127; any other qparams -- legal for a
  //  C-ABI caller, e.g. 8-bit activations -- take the exact scalar path below)
  const bool packed = (a.C & 15) == 0 && a.C <= 256 && ((reinterpret_cast<uintptr_t>(ws) | (uintptr_t)a.w_ss) & 3) == 0 &&
                      a.z_x >= 0 && a.z_x <= 128 && min(a.a_hi, 255) - a.z_x <= 127;
  for (int bi = 0; bi < QBNN_HEAD_IMGS; ++bi) {
    const int b = b0 + bi;
    if (b >= a.B) break;
    const uint8_t* xs = a.x + (int64_t)s * a.x_ss + (int64_t)b * a.kk * a.C;
    // The pooling loop below reads one dword per lane and pixel -- C bytes per load instruction, kk dependent-latency loads in a row.
    // Where the image is a whole number of 16-byte chunks (ResNet: 16 x 192 = 3 KiB) it is brought into LDS with kk C / 1024 wide loads
    // per lane first and the loop reads it from there.
    const int img_bytes = a.kk * a.C;
    const bool staged = (img_bytes & 15) == 0 && img_bytes <= 4096 && ((reinterpret_cast<uintptr_t>(xs)) & 15) == 0;
    if (staged) {
      for (int i = lane; i < img_bytes / 16; i += 64)
        *reinterpret_cast<v4i*>(&stage[wave][16 * i]) = *reinterpret_cast<const v4i*>(xs + 16 * i);
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): same-wave LDS write -> read
      __builtin_amdgcn_wave_barrier();
    }
    // AvgPool2d(k) on quint8, channels-last: q = clamp(rne((sum - kk z) / kk) + z, 0, 255); then clamp_activation.
    // Four channels per lane (one dword per pixel) when C is a multiple of 4.
    if ((a.C & 3) == 0) {
      for (int c4 = lane; c4 < a.C / 4; c4 += 64) {
        int sum[4] = {0, 0, 0, 0};
        for (int p = 0; p < a.kk; ++p) {
          const uint32_t v = staged ? *reinterpret_cast<const uint32_t*>(&stage[wave][p * a.C + 4 * c4]) : *reinterpret_cast<const uint32_t*>(xs + p * a.C + 4 * c4);
          sum[0] += v & 0xffu; sum[1] += (v >> 8) & 0xffu; sum[2] += (v >> 16) & 0xffu; sum[3] += v >> 24;
        }
        uint32_t pk = 0;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          int q = min(max(rne_sat((float)(sum[jj] - a.kk * a.z_x) * a.inv_kk) + a.z_x, 0), 255);
          q = min(q, a.a_hi) - a.z_x;
          if (packed) pk |= ((uint32_t)q & 0xffu) << (8 * jj);
          else pooled[wave][4 * c4 + jj] = q;
        }
        if (packed) pooled[wave][c4] = (int)pk;
      }
    } else {
      for (int c = lane; c < a.C; c += 64) {
        int sum = 0;
        for (int p = 0; p < a.kk; ++p) sum += xs[p * a.C + c];
        int q = min(max(rne_sat((float)(sum - a.kk * a.z_x) * a.inv_kk) + a.z_x, 0), 255);
        pooled[wave][c] = min(q, a.a_hi) - a.z_x;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): same-wave LDS write -> read
    __builtin_amdgcn_wave_barrier();
    // Linear: lane = (output n, quarter j of the channels); integer partial sums, then a 4-lane butterfly (exact, any order)
    int acc = 0;
    if (n < a.N) {
      if (packed) {
        const int dpq = a.C / 16;                                  // dwords per quarter
        const uint32_t* wp = reinterpret_cast<const uint32_t*>(ws + n * a.C) + j * dpq;
        int ps = 0;
        for (int d = 0; d < dpq; ++d) {
          const int pv = pooled[wave][j * dpq + d];
          acc = __builtin_amdgcn_sdot4(pv, (int)wp[d], acc, false);
          ps = __builtin_amdgcn_sdot4(pv, 0x01010101, ps, false);
        }
        acc -= a.z_w * ps;
      } else {
        const int c_per = (a.C + 3) / 4, c0 = j * c_per, c1 = min(c0 + c_per, a.C);
        for (int c = c0; c < c1; ++c) acc += pooled[wave][c] * ((int)ws[n * a.C + c] - a.z_w);
      }
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    float logit = -INFINITY;
    if (n < a.N && j == 0) {
      float xf = (float)acc;
      if (a.bias) xf = __builtin_fmaf(a.bias[n], a.rcp, xf);
      int q = min(max(a.z_y + rne_sat(xf * a.mult), 0), 255);
      q = min(q, a.a_hi);
      logit = (float)(q - a.z_y) * a.s_y;     // DeQuantStub
    }
    float mx = logit;
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float e = (n < a.N && j == 0) ? expf(logit - mx) : 0.f;
    float sum = e;
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (n < a.N && j == 0) a.probs[((int64_t)s * a.B + b) * a.N + n] = e / sum;
    __builtin_amdgcn_wave_barrier();      // the next image overwrites pooled[wave]
  }
}

static int build_head_args(HeadArgs& a, const uint8_t* x, int64_t x_ss, const int8_t* w, int64_t w_ss, const float* bias, float* probs,
                           const qbnn_head_desc* d) {
  if (d->C > 256 || d->N > 16 || d->C <= 0 || d->N <= 0) return fail(QBNN_E_INVALID, "qbnn_head_i8: C <= 256 and N <= 16 required%s");
  a.x = x; a.x_ss = x_ss; a.w = w; a.w_ss = w_ss; a.bias = d->has_bias ? bias : nullptr; a.probs = probs;
  a.B = d->B; a.kk = d->k * d->k; a.C = d->C; a.N = d->N;
  a.z_x = d->z_x; a.z_w = d->z_w; a.z_y = d->z_y; a.a_hi = d->a_hi;
  a.inv_kk = 1.0f / (float)(d->k * d->k);
  const float atw = d->s_x * d->s_w;
  a.rcp = 1.0f / atw; a.mult = atw / d->s_y; a.s_y = d->s_y;
  return QBNN_OK;
}

QBNN_EXPORT int qbnn_head_i8_mc(const uint8_t* x, int64_t x_ss, const int8_t* w, int64_t w_ss, const float* bias,
                                float* probs, int32_t n_samples, const qbnn_head_desc* d, void* stream) {
  if (!x || !w || !probs || !d || n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_head_i8_mc: bad argument%s");
  ArgsArr<HeadArgs, 1> one;
  if (int rc = build_head_args(one.m[0], x, x_ss, w, w_ss, bias, probs, d)) return rc;
  hipLaunchKernelGGL(head_i8_kernel<1>, dim3(ceil_div(d->B, 4 * QBNN_HEAD_IMGS), n_samples), dim3(256), 0, (hipStream_t)stream, one);
  return check_launch("qbnn_head_i8_mc");
}

#define QBNN_HEAD_CALLS 32           // a head's argument block is 96 bytes: 32 of them fit the 4 KiB of kernel arguments (one launch for a 16-member ensemble)
static_assert(sizeof(ArgsArr<HeadArgs, QBNN_HEAD_CALLS>) <= 3840, "kernel arguments are limited to 4 KiB");
QBNN_EXPORT int qbnn_head_i8_multi(const qbnn_head_call* calls, int32_t n_calls, void* stream) {
  if (!calls || n_calls <= 0) return fail(QBNN_E_INVALID, "qbnn_head_i8_multi: bad argument%s");
  for (int c0 = 0; c0 < n_calls;) {
    const int n = n_calls - c0 < QBNN_HEAD_CALLS ? n_calls - c0 : QBNN_HEAD_CALLS;
    ArgsArr<HeadArgs, QBNN_HEAD_CALLS> all;
    memset(&all, 0, sizeof(all));
    int maxB = 0, maxS = 0;
    for (int i = 0; i < n; ++i) {
      const qbnn_head_call& k = calls[c0 + i];
      if (!k.x || !k.w || !k.probs || !k.desc || k.n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_head_i8_multi: bad call entry%s");
      if (int rc = build_head_args(all.m[i], k.x, k.x_sample_stride, k.w, k.w_sample_stride, k.bias, k.probs, k.desc)) return rc;
      if (k.n_samples != calls[c0].n_samples) return fail(QBNN_E_INVALID, "qbnn_head_i8_multi: the calls of one launch evaluate the same number of samples%s");
      maxB = k.desc->B > maxB ? k.desc->B : maxB; maxS = k.n_samples;
    }
    hipLaunchKernelGGL(head_i8_kernel<QBNN_HEAD_CALLS>, dim3(ceil_div(maxB, 4 * QBNN_HEAD_IMGS), maxS, n), dim3(256), 0, (hipStream_t)stream, all);
    if (int rc = check_launch("qbnn_head_i8_multi")) return rc;
    c0 += n;
  }
  return QBNN_OK;
}

// MC reduction.  The sums are kept in fp64: var = (sum p^2 - (sum p)^2 / S) / (S - 1) cancels catastrophically in fp32 when the
// spread of a class probability is small against its mean (the regression head's predictive variance feeds the reference's NLL).
// FINAL: this launch also finalises (single rank, last chunk): mean / unbiased variance as fp32, no further launches.
template <bool FINAL>
__global__ __launch_bounds__(256) void reduce_moments_kernel(const float* __restrict__ probs, int S, int64_t n, int accumulate,
                                                             double* __restrict__ mom, int total, float* __restrict__ mean_out,
                                                             float* __restrict__ var_out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s1 = accumulate ? mom[i] : 0.0, s2 = accumulate ? mom[n + i] : 0.0;
  int s = 0;
  for (; s + 8 <= S; s += 8) {            // 8 independent loads in flight, summed in sample order
    float p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = probs[(int64_t)(s + j) * n + i];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1 += (double)p[j]; s2 += (double)p[j] * (double)p[j]; }
  }
  for (; s < S; ++s) {
    const double p = (double)probs[(int64_t)s * n + i];
    s1 += p;
    s2 += p * p;
  }
  mom[i] = s1; mom[n + i] = s2;
  if (FINAL) {
    const double m = s1 / (double)total;
    mean_out[i] = (float)m;
    if (var_out) var_out[i] = total > 1 ? (float)(fmax(s2 - s1 * m, 0.0) / (double)(total - 1)) : 0.f;
  }
}

__global__ __launch_bounds__(256) void finalize_moments_kernel(const double* __restrict__ mom, int64_t n, int total,
                                                               float* __restrict__ mean_out, float* __restrict__ var_out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double s1 = mom[i], s2 = mom[n + i], m = s1 / (double)total;
  mean_out[i] = (float)m;
  if (var_out) var_out[i] = total > 1 ? (float)(fmax(s2 - s1 * m, 0.0) / (double)(total - 1)) : 0.f;
}

QBNN_EXPORT int qbnn_reduce_moments(const float* probs, int32_t S, int64_t n, int32_t accumulate, double* mom,
                                    int32_t finalize_total, float* mean_out, float* var_out, void* stream) {
  if (!probs || !mom || S <= 0 || n <= 0) return fail(QBNN_E_INVALID, "qbnn_reduce_moments: bad argument%s");
  if (finalize_total > 0 && !mean_out) return fail(QBNN_E_INVALID, "qbnn_reduce_moments: finalize_total > 0 needs mean_out%s");
  const dim3 grid((unsigned)((n + 255) / 256));
  if (finalize_total > 0)
    hipLaunchKernelGGL(reduce_moments_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, probs, S, n, accumulate, mom, finalize_total, mean_out, var_out);
  else
    hipLaunchKernelGGL(reduce_moments_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, probs, S, n, accumulate, mom, 0, nullptr, nullptr);
  return check_launch("qbnn_reduce_moments");
}

QBNN_EXPORT int qbnn_finalize_moments(const double* mom, int64_t n, int32_t total_samples, float* mean_out, float* var_out, void* stream) {
  if (!mom || !mean_out || n <= 0 || total_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_finalize_moments: bad argument%s");
  hipLaunchKernelGGL(finalize_moments_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mom, n, total_samples, mean_out, var_out);
  return check_launch("qbnn_finalize_moments");
}

// =====================================================================================
// MC-Dropout path (BASELINE config 2: LeNet, mcdropout/models_mc.py:75-111).  These nets are tiny and
// launch/latency-bound at their sizes: plain one-thread-per-output kernels, any geometry.
// =====================================================================================
struct GenConvArgs {
  const uint8_t* x; int64_t x_ss; const int8_t* w; int64_t w_ss; const float* bias; uint8_t* y; int64_t y_ss;
  int B, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
  int z_x, z_w, z_y, lo, hi;
  float rcp, mult;
};

__global__ __launch_bounds__(256) void conv_generic_i8_kernel(const GenConvArgs a) {
  const int64_t total = (int64_t)a.B * a.Ho * a.Wo * a.Cout;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int s = blockIdx.y;
  const int co = (int)(idx % a.Cout);
  int64_t t = idx / a.Cout;
  const int ow = (int)(t % a.Wo); t /= a.Wo;
  const int oh = (int)(t % a.Ho);
  const int b = (int)(t / a.Ho);
  const uint8_t* xs = a.x + (int64_t)s * a.x_ss + (int64_t)b * a.H * a.W * a.Cin;
  const int8_t* ws = a.w + (int64_t)s * a.w_ss + (int64_t)co * a.KH * a.KW * a.Cin;
  int acc = 0;
  for (int kh = 0; kh < a.KH; ++kh) {
    const int ih = oh * a.stride - a.pad + kh;
    if (ih < 0 || ih >= a.H) continue;
    for (int kw = 0; kw < a.KW; ++kw) {
      const int iw = ow * a.stride - a.pad + kw;
      if (iw < 0 || iw >= a.W) continue;
      const uint8_t* xp = xs + ((int64_t)ih * a.W + iw) * a.Cin;
      const int8_t* wp = ws + (kh * a.KW + kw) * a.Cin;
      for (int c = 0; c < a.Cin; ++c) acc += ((int)xp[c] - a.z_x) * ((int)wp[c] - a.z_w);
    }
  }
  float xf = (float)acc;
  if (a.bias) xf = __builtin_fmaf(a.bias[co], a.rcp, xf);
  int q = a.z_y + rne_sat(xf * a.mult);
  q = min(max(q, a.lo), a.hi);
  a.y[(int64_t)s * a.y_ss + idx] = (uint8_t)q;
}

// The same contract on the matrix pipe, any geometry.  Workgroup = 64 output pixels x 64 output channels (4 waves of
// 32 x 32), K = KH KW Cin walked in 32-byte chunks that are gathered byte by byte (im2col on the fly; Cin = 1, 20, 50, 2450 ...
// give no alignment to build on) into LDS rows of 48 bytes (conflict-free ds_read_b128 fragments).
// Neither x - z_x nor w - z_w fits a signed byte in general, so the MFMA runs on the raw bytes x' = x - 128 (= x ^ 0x80)
// and w, with out-of-map taps fed x = z_x (their exact contribution is then 0), and the zero points enter afterwards:
//   sum_k (x_k - z_x)(w_k - z_w) = acc - z_w R + a Wsum[co] - K a z_w,   a = 128 - z_x, R = sum_k x'_k, Wsum = sum_k w_k
// -- all int32-exact; R and Wsum are v_dot4 sums over the fragments the MFMA consumes.  Bit-identical to
// conv_generic_i8_kernel (tests compare the two), 50-200x faster on the LeNet / MLP layers.
// GB = 4 / 2: Cin % GB == 0 and GB-byte aligned operands -- a unit never straddles a tap, so the gather moves GB channels per
// load with one index step per unit instead of per byte (the gather's vector-ALU work is what bounds this kernel).
template <int GB>
__global__ __launch_bounds__(256) void conv_generic_mfma_i8_kernel(const GenConvArgs a) {
  constexpr int LD = 48;
  __shared__ __attribute__((aligned(16))) uint8_t As[64 * LD];      // weights [n][k]
  __shared__ __attribute__((aligned(16))) uint8_t Bs[64 * LD];      // pixels  [p][k], bytes x ^ 0x80
  __shared__ int wsum_lds[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int s = blockIdx.z;
  const int p0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int npix = a.B * a.Ho * a.Wo;
  const int K = a.KH * a.KW * a.Cin;
  const uint8_t* xs = a.x + (int64_t)s * a.x_ss;
  const int8_t* ws = a.w + (int64_t)s * a.w_ss;
  const int row = tid >> 2, kb = (tid & 3) * 8;
  const int p = p0 + row, n = n0 + row;
  int pb = -1, ih0 = 0, iw0 = 0;
  if (p < npix) { pb = p / (a.Ho * a.Wo); const int rem = p - pb * a.Ho * a.Wo; ih0 = (rem / a.Wo) * a.stride - a.pad; iw0 = (rem % a.Wo) * a.stride - a.pad; }
  const int64_t xbase = (int64_t)(pb < 0 ? 0 : pb) * a.H * a.W * a.Cin;
  const int8_t* wrow = ws + (int64_t)(n < a.Cout ? n : 0) * K;
  const uint32_t xpad = (uint32_t)(a.z_x ^ 0x80) & 0xffu;
  // the thread's 8 bytes of the chunk in units of GB = 1, 2 or 4 bytes (Cin % GB == 0: a unit never straddles a tap)
  auto gather = [&](int k0, uint32_t (&xv)[2], uint32_t (&wv)[2]) {
    int kk = k0 + kb;
    int tap = kk / a.Cin, c = kk - tap * a.Cin;
    int kh = tap / a.KW, kw = tap - kh * a.KW;
    xv[0] = xv[1] = wv[0] = wv[1] = 0u;
    constexpr uint32_t UMASK = GB == 4 ? 0xffffffffu : (GB == 2 ? 0xffffu : 0xffu);
#pragma unroll
    for (int j = 0; j < 8 / GB; ++j, kk += GB) {
      uint32_t xb = 0u, wb = 0u;
      if (kk < K) {
        if (pb >= 0) {
          const int ih = ih0 + kh, iw = iw0 + kw;
          const bool in = (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
          const uint8_t* src = xs + xbase + ((int64_t)ih * a.W + iw) * a.Cin + c;
          uint32_t v;
          if constexpr (GB == 4) v = *reinterpret_cast<const uint32_t*>(src);
          else if constexpr (GB == 2) v = *reinterpret_cast<const uint16_t*>(src);
          else v = *src;
          xb = in ? (v ^ (0x80808080u & UMASK)) : (xpad * 0x01010101u) & UMASK;
        }
        if (n < a.Cout) {
          if constexpr (GB == 4) wb = *reinterpret_cast<const uint32_t*>(wrow + kk);
          else if constexpr (GB == 2) wb = *reinterpret_cast<const uint16_t*>(wrow + kk);
          else wb = (uint32_t)(uint8_t)wrow[kk];
        }
      }
      constexpr int PER = 4 / GB;                       // units per dword
      xv[j / PER] |= xb << (8 * GB * (j % PER));
      wv[j / PER] |= wb << (8 * GB * (j % PER));
      c += GB;
      if (c == a.Cin) { c = 0; if (++kw == a.KW) { kw = 0; ++kh; } }
    }
  };
  v16i acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0;
  int rsum = 0, wsum = 0;
  uint32_t xv[2], wv[2];
  gather(0, xv, wv);
  for (int k0 = 0; k0 < K; k0 += 32) {
    *reinterpret_cast<v2i*>(&Bs[row * LD + kb]) = v2i{(int)xv[0], (int)xv[1]};
    *reinterpret_cast<v2i*>(&As[row * LD + kb]) = v2i{(int)wv[0], (int)wv[1]};
    __syncthreads();
    if (k0 + 32 < K) gather(k0 + 32, xv, wv);           // next chunk in flight under the MFMA
    const v4i av = *reinterpret_cast<const v4i*>(&As[(wn * 32 + (lane & 31)) * LD + 16 * (lane >> 5)]);
    const v4i bv = *reinterpret_cast<const v4i*>(&Bs[(wm * 32 + (lane & 31)) * LD + 16 * (lane >> 5)]);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rsum = __builtin_amdgcn_sdot4(bv[i], 0x01010101, rsum, false);
      wsum = __builtin_amdgcn_sdot4(av[i], 0x01010101, wsum, false);
    }
    __syncthreads();
  }
  const int R = rsum + __shfl_xor(rsum, 32);              // this lane's pixel (lane & 31), all k
  const int Wn = wsum + __shfl_xor(wsum, 32);             // weight row wn * 32 + (lane & 31), all k
  if (wm == 0 && lane < 32) wsum_lds[wn * 32 + lane] = Wn;
  __syncthreads();
  const int po = p0 + wm * 32 + (lane & 31);
  if (po >= npix) return;
  const int aoff = 128 - a.z_x;
  const int base = -a.z_w * R - K * aoff * a.z_w;
  uint8_t* yp = a.y + (int64_t)s * a.y_ss + (int64_t)po * a.Cout;
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nl = wn * 32 + 8 * g + 4 * (lane >> 5) + i, no = n0 + nl;
      if (no < a.Cout) {
        float xf = (float)(acc[4 * g + i] + aoff * wsum_lds[nl] + base);
        if (a.bias) xf = __builtin_fmaf(a.bias[no], a.rcp, xf);
        int q = a.z_y + rne_sat(xf * a.mult);
        q = min(max(q, a.lo), a.hi);
        yp[no] = (uint8_t)q;
      }
    }
}

static bool generic_naive() { static const bool v = [] { const char* e = getenv("QBNN_GENERIC_NAIVE"); return e && e[0] == '1'; }(); return v; }

static int conv2d_i8_generic(const uint8_t* x, int64_t x_ss, const int8_t* w_ohwi, int64_t w_ss, const float* bias, uint8_t* y,
                             int64_t y_ss, int32_t n_samples, const qbnn_conv_desc* d, void* stream, bool scalar_form);

QBNN_EXPORT int qbnn_conv2d_i8_generic_mc(const uint8_t* x, int64_t x_ss, const int8_t* w_ohwi, int64_t w_ss, const float* bias,
                                          uint8_t* y, int64_t y_ss, int32_t n_samples, const qbnn_conv_desc* d, void* stream) {
  return conv2d_i8_generic(x, x_ss, w_ohwi, w_ss, bias, y, y_ss, n_samples, d, stream, generic_naive());
}

QBNN_EXPORT int qbnn_conv2d_i8_generic_scalar_mc(const uint8_t* x, int64_t x_ss, const int8_t* w_ohwi, int64_t w_ss, const float* bias,
                                                 uint8_t* y, int64_t y_ss, int32_t n_samples, const qbnn_conv_desc* d, void* stream) {
  return conv2d_i8_generic(x, x_ss, w_ohwi, w_ss, bias, y, y_ss, n_samples, d, stream, true);
}

static int conv2d_i8_generic(const uint8_t* x, int64_t x_ss, const int8_t* w_ohwi, int64_t w_ss, const float* bias, uint8_t* y,
                             int64_t y_ss, int32_t n_samples, const qbnn_conv_desc* d, void* stream, bool scalar_form) {
  if (!x || !w_ohwi || !y || !d || n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_generic_mc: bad argument%s");
  if (d->a_hi > 255 || d->a_hi < 1) return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_generic_mc: bad a_hi%s");
  GenConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.x_ss = x_ss; a.w = w_ohwi; a.w_ss = w_ss; a.bias = d->has_bias ? bias : nullptr; a.y = y; a.y_ss = y_ss;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.KH = a.KW = d->ksize; a.stride = d->stride; a.pad = d->pad;
  a.Ho = (d->H + 2 * d->pad - d->ksize) / d->stride + 1; a.Wo = (d->W + 2 * d->pad - d->ksize) / d->stride + 1;
  a.z_x = d->z_x; a.z_w = d->z_w; a.z_y = d->z_y; a.lo = d->relu ? d->z_y : 0; a.hi = d->a_hi < 255 ? d->a_hi : 255;
  const float atw = d->s_x * d->s_w;
  a.rcp = 1.0f / atw; a.mult = atw / d->s_y;
  if (a.Ho <= 0 || a.Wo <= 0 || a.Cin <= 0 || a.Cout <= 0) return fail(QBNN_E_INVALID, "qbnn_conv2d_i8_generic_mc: empty geometry%s");
  const int64_t npix = (int64_t)a.B * a.Ho * a.Wo;
  const int64_t total = npix * a.Cout;
  if (scalar_form || (int64_t)a.KH * a.KW * a.Cin > (1 << 16))      // (int32 head-room of the correction terms)
    hipLaunchKernelGGL(conv_generic_i8_kernel, dim3((unsigned)((total + 255) / 256), n_samples), dim3(256), 0, (hipStream_t)stream, a);
  else {
    const dim3 grid((unsigned)((npix + 63) / 64), (unsigned)((a.Cout + 63) / 64), n_samples);
    auto unit_ok = [&](int gb) {
      return (a.Cin % gb) == 0 && (x_ss % gb) == 0 && (w_ss % gb) == 0 && (reinterpret_cast<uintptr_t>(x) % gb) == 0 &&
             (reinterpret_cast<uintptr_t>(w_ohwi) % gb) == 0;
    };
    if (unit_ok(4)) hipLaunchKernelGGL(conv_generic_mfma_i8_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (unit_ok(2)) hipLaunchKernelGGL(conv_generic_mfma_i8_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(conv_generic_mfma_i8_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a);
  }
  return check_launch("qbnn_conv2d_i8_generic_mc");
}

// Quantised BernoulliDropout (mcdropout/dropout.py:15-40), x [S][B][HW][C]: one Bernoulli(keep) draw per (sample, b, c)
// from the Philox uniform stream {ctr = {i >> 2, layer, sample, 1}}[i & 3], i = b * C + c  (or mask_in in parity mode).
// One workgroup = one (sample, image): thread t owns channel slot t % CS (CS = C, or C / 4 dwords when C % 4 == 0) and
// draws that slot's mask ONCE, then walks the pixels t / CS, t / CS + 256 / CS, ... -- consecutive threads touch consecutive
// bytes.  (The first form drew a Philox block per element: HW-fold redundant, 2 ms per LeNet pass.)
template <bool VEC4>
__global__ __launch_bounds__(256) void dropout_q_kernel(const uint8_t* __restrict__ x, int64_t x_ss, int B, int HW, int C,
                                                         float keep, int z_x, float inv_sm, int z_m, float mult, int hi,
                                                         uint32_t seed_lo, uint32_t seed_hi, uint32_t layer_id, uint32_t sample_begin,
                                                         const float* __restrict__ mask_in, uint8_t* __restrict__ y, int64_t y_ss,
                                                         const uint32_t* __restrict__ nd) {
  if (nd) { seed_lo = nd[0]; seed_hi = nd[1]; sample_begin = nd[2]; }      // captured-graph mode: the seed lives in device memory
  const int b = blockIdx.x, s = blockIdx.y;
  const int CS = VEC4 ? C / 4 : C;
  const int per_pass = 256 / CS > 0 ? 256 / CS : 1;          // pixels covered by the workgroup per trip (CS <= 256), else slots loop
  const uint8_t* xs = x + (int64_t)s * x_ss + (int64_t)b * HW * C;
  uint8_t* ys = y + (int64_t)s * y_ss + (int64_t)b * HW * C;
  auto mask_q = [&](int c) {                                   // quantised mask value minus its zero point, channel c of image b
    return drop_mask_q(b * C + c, s, (int64_t)B * C, keep, inv_sm, z_m, seed_lo, seed_hi, layer_id, sample_begin, mask_in);
  };
  auto one = [&](int xb, int mq) { return drop_one(xb, mq, z_x, z_m, mult, hi); };
  for (int slot = threadIdx.x % (CS < 256 ? CS : 256); slot < CS; slot += 256) {       // one trip unless C > 256 (VEC4: C > 1024)
    const int first = CS < 256 ? threadIdx.x / CS : 0;
    if (CS < 256 && first >= per_pass) break;                                       // threads beyond a whole number of pixels idle
    if constexpr (VEC4) {
      int mq[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) mq[j] = mask_q(4 * slot + j);
      for (int hw = first; hw < HW; hw += per_pass) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>(xs + (int64_t)hw * C + 4 * slot);
        const uint32_t o = one((int)(v & 0xffu), mq[0]) | (one((int)((v >> 8) & 0xffu), mq[1]) << 8) |
                           (one((int)((v >> 16) & 0xffu), mq[2]) << 16) | (one((int)(v >> 24), mq[3]) << 24);
        *reinterpret_cast<uint32_t*>(ys + (int64_t)hw * C + 4 * slot) = o;
      }
    } else {
      const int mq = mask_q(slot);
      for (int hw = first; hw < HW; hw += per_pass) ys[(int64_t)hw * C + slot] = (uint8_t)one((int)xs[(int64_t)hw * C + slot], mq);
    }
  }
}

QBNN_EXPORT int qbnn_dropout_q_mc(const uint8_t* x, int64_t x_ss, int32_t B, int32_t HW, int32_t C, float keep_prob, float s_x,
                                  int32_t z_x, float s_m, int32_t z_m, int32_t a_hi, uint64_t seed, uint32_t layer_id,
                                  uint32_t sample_begin, const float* mask_in, uint8_t* y, int64_t y_ss, int32_t n_samples,
                                  void* stream) {
  if (!x || !y || B <= 0 || HW <= 0 || C <= 0 || n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_dropout_q_mc: bad argument%s");
  const float mult = (float)((double)s_x * (double)s_m / (double)s_m);     // ATen qmul: self_scale * other_scale / out_scale
  const bool vec4 = (C % 4) == 0 && (x_ss % 4) == 0 && (y_ss % 4) == 0 && (reinterpret_cast<uintptr_t>(x) % 4) == 0 &&
                    (reinterpret_cast<uintptr_t>(y) % 4) == 0;
  if (vec4)
    hipLaunchKernelGGL(dropout_q_kernel<true>, dim3((unsigned)B, n_samples), dim3(256), 0, (hipStream_t)stream,
                       x, x_ss, B, HW, C, keep_prob, z_x, 1.0f / s_m, z_m, mult, a_hi < 255 ? a_hi : 255, (uint32_t)seed,
                       (uint32_t)(seed >> 32), layer_id, sample_begin, mask_in, y, y_ss, g_noise_dev);
  else
    hipLaunchKernelGGL(dropout_q_kernel<false>, dim3((unsigned)B, n_samples), dim3(256), 0, (hipStream_t)stream,
                       x, x_ss, B, HW, C, keep_prob, z_x, 1.0f / s_m, z_m, mult, a_hi < 255 ? a_hi : 255, (uint32_t)seed,
                       (uint32_t)(seed >> 32), layer_id, sample_begin, mask_in, y, y_ss, g_noise_dev);
  return check_launch("qbnn_dropout_q_mc");
}

__global__ __launch_bounds__(256) void maxpool2_q_kernel(const uint8_t* __restrict__ x, int64_t x_ss, int B, int H, int W, int C,
                                                          int hi, uint8_t* __restrict__ y, int64_t y_ss) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)B * Ho * Wo * C;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int s = blockIdx.y;
  const int c = (int)(idx % C);
  int64_t t = idx / C;
  const int ow = (int)(t % Wo); t /= Wo;
  const int oh = (int)(t % Ho);
  const int64_t b = t / Ho;
  const uint8_t* xs = x + (int64_t)s * x_ss;
  int m = 0;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j) m = max(m, (int)xs[((b * H + oh * 2 + i) * W + ow * 2 + j) * C + c]);
  y[(int64_t)s * y_ss + idx] = (uint8_t)min(m, hi);
}

QBNN_EXPORT int qbnn_maxpool2_q_mc(const uint8_t* x, int64_t x_ss, int32_t B, int32_t H, int32_t W, int32_t C, int32_t a_hi,
                                   uint8_t* y, int64_t y_ss, int32_t n_samples, void* stream) {
  if (!x || !y || B <= 0 || H < 2 || W < 2 || C <= 0 || n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_maxpool2_q_mc: bad argument%s");
  const int64_t total = (int64_t)B * (H / 2) * (W / 2) * C;
  hipLaunchKernelGGL(maxpool2_q_kernel, dim3((unsigned)((total + 255) / 256), n_samples), dim3(256), 0, (hipStream_t)stream,
                     x, x_ss, B, H, W, C, a_hi < 255 ? a_hi : 255, y, y_ss);
  return check_launch("qbnn_maxpool2_q_mc");
}

// DeQuantStub + softmax over the last dim (models_mc.py:104-111): x [S][B][N] uint8 -> probs [S][B][N] fp32
__global__ __launch_bounds__(256) void dequant_softmax_kernel(const uint8_t* __restrict__ x, int64_t x_ss, int B, int N, float sc,
                                                               int z, float* __restrict__ probs) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const int s = blockIdx.y;
  const uint8_t* xs = x + (int64_t)s * x_ss + (int64_t)b * N;
  float mx = -INFINITY;
  for (int i = 0; i < N; ++i) mx = fmaxf(mx, (float)((int)xs[i] - z) * sc);
  float sum = 0.f;
  for (int i = 0; i < N; ++i) sum += expf((float)((int)xs[i] - z) * sc - mx);
  for (int i = 0; i < N; ++i) probs[((int64_t)s * B + b) * N + i] = expf((float)((int)xs[i] - z) * sc - mx) / sum;
}

QBNN_EXPORT int qbnn_dequant_softmax_mc(const uint8_t* x, int64_t x_ss, int32_t B, int32_t N, float scale, int32_t zero_point,
                                        float* probs, int32_t n_samples, void* stream) {
  if (!x || !probs || B <= 0 || N <= 0 || n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_dequant_softmax_mc: bad argument%s");
  hipLaunchKernelGGL(dequant_softmax_kernel, dim3((B + 255) / 256, n_samples), dim3(256), 0, (hipStream_t)stream, x, x_ss, B, N,
                     scale, zero_point, probs);
  return check_launch("qbnn_dequant_softmax_mc");
}

// =====================================================================================
// fp32 Bayes-by-backprop path (BASELINE config 0: 3x100 MLP; reference bbb/linear.py:42-50).  Tiny, latency-bound:
// plain VALU kernels.  W = mu + eps * sigma is two fp32 roundings (FloatFunctional mul then add), as in the reference.
// =====================================================================================
__global__ __launch_bounds__(256) void sample_weights_f32_kernel(const float* __restrict__ mu, const float* __restrict__ sigma,
                                                                  int64_t n, uint32_t seed_lo, uint32_t seed_hi, uint32_t layer_id,
                                                                  uint32_t sample_begin, const float* __restrict__ eps_in,
                                                                  float* __restrict__ w, const uint32_t* __restrict__ nd) {
  if (nd) { seed_lo = nd[0]; seed_hi = nd[1]; sample_begin = nd[2]; }      // captured-graph mode: the seed lives in device memory
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;      // group of 4 consecutive weights
  if (g * 4 >= n) return;
  const int s = blockIdx.y;
  float e[4];
  if (eps_in) {
    for (int j = 0; j < 4; ++j) e[j] = (g * 4 + j < n) ? eps_in[(int64_t)s * n + g * 4 + j] : 0.f;
  } else {
    qbnn::normal4(qbnn::philox4x32_10((uint32_t)g, layer_id, sample_begin + s, 0u, seed_lo, seed_hi), e);
  }
  if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(mu) | reinterpret_cast<uintptr_t>(sigma) | reinterpret_cast<uintptr_t>(w)) & 15) == 0) {
    // whole group in range and 16-byte aligned: one vector load per operand, one vector store (same arithmetic per element)
    const float4 m4 = reinterpret_cast<const float4*>(mu)[g], s4 = reinterpret_cast<const float4*>(sigma)[g];
    float4 o;
    { const float t = e[0] * s4.x; o.x = m4.x + t; }
    { const float t = e[1] * s4.y; o.y = m4.y + t; }
    { const float t = e[2] * s4.z; o.z = m4.z + t; }
    { const float t = e[3] * s4.w; o.w = m4.w + t; }
    reinterpret_cast<float4*>(w + (int64_t)s * n)[g] = o;
    return;
  }
  for (int j = 0; j < 4; ++j) {
    const int64_t i = g * 4 + j;
    if (i < n) { const float t = e[j] * sigma[i]; w[(int64_t)s * n + i] = mu[i] + t; }
  }
}

QBNN_EXPORT int qbnn_sample_weights_f32(const float* mu, const float* sigma, int64_t n, uint64_t seed, uint32_t layer_id,
                                        uint32_t sample_begin, int32_t n_samples, const float* eps_in, float* w_out, void* stream) {
  if (!mu || !sigma || !w_out || n <= 0 || n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_sample_weights_f32: bad argument%s");
  const int64_t groups = (n + 3) / 4;
  hipLaunchKernelGGL(sample_weights_f32_kernel, dim3((unsigned)((groups + 255) / 256), n_samples), dim3(256), 0, (hipStream_t)stream,
                     mu, sigma, n, (uint32_t)seed, (uint32_t)(seed >> 32), layer_id, sample_begin, eps_in, w_out, g_noise_dev);
  return check_launch("qbnn_sample_weights_f32");
}

__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ x, int64_t x_ss, const float* __restrict__ w,
                                                          int64_t w_ss, const float* __restrict__ bias, float* __restrict__ y,
                                                          int64_t y_ss, int B, int K, int N, int act) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * N) return;
  const int s = blockIdx.y;
  const int n = (int)(idx % N), b = (int)(idx / N);
  const float* xp = x + (int64_t)s * x_ss + (int64_t)b * K;
  const float* wp = w + (int64_t)s * w_ss + (int64_t)n * K;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = __builtin_fmaf(xp[k], wp[k], acc);
  if (bias) acc = acc + bias[n];
  if (act == 1) acc = fmaxf(acc, 0.f);
  else if (act == 2) acc = expf(acc);
  y[(int64_t)s * y_ss + idx] = acc;
}

QBNN_EXPORT int qbnn_linear_f32_mc(const float* x, int64_t x_ss, const float* w, int64_t w_ss, const float* bias, float* y,
                                   int64_t y_ss, int32_t B, int32_t K, int32_t N, int32_t act, int32_t n_samples, void* stream) {
  if (!x || !w || !y || B <= 0 || K <= 0 || N <= 0 || n_samples <= 0 || act < 0 || act > 2)
    return fail(QBNN_E_INVALID, "qbnn_linear_f32_mc: bad argument%s");
  // a Linear is a 1x1 conv over a 1x1 map with [out][in] = OHWI weights: the MFMA implicit-GEMM kernels of qbnn_f32.hip
  // (float4 path when in_features % 4 == 0); only the exp head (N = 1) stays on the one-thread-per-output kernel
  if (act != 2)
    return qbnn_conv2d_f32_fused_mc(x, x_ss, w, w_ss, nullptr, bias, nullptr, nullptr, nullptr, 0, y, y_ss, B, 1, 1, K, N, 1, 1, 0,
                                    (act == 1 ? 1 : 0) | 4, n_samples, nullptr, stream);
  const int64_t total = (int64_t)B * N;
  hipLaunchKernelGGL(linear_f32_kernel, dim3((unsigned)((total + 255) / 256), n_samples), dim3(256), 0, (hipStream_t)stream,
                     x, x_ss, w, w_ss, bias, y, y_ss, B, K, N, act);
  return check_launch("qbnn_linear_f32_mc");
}

// =====================================================================================
// Classification metrics on the reduced output (reference src/metrics.py:8-116, :355-430), on device so the [B,C]
// predictive mean need not return to the host per batch.  One thread per image; per-block partial sums
//   [0] errors  [1] sum -log(p_target + 1e-8)  [2] sum_c (p - onehot)^2  [3] sum_c -p log(p + 1e-8)
//   [4+b] count, [14+b] confidence sum, [24+b] accuracy sum of calibration bin b (10 uniform bins on max-prob;
//   bin = index of the first boundary >= confidence, minus 1: torch.bucketize(conf, linspace(0,1,11), right=True) - 1).
// The host sums the partial rows (deterministic).
// =====================================================================================
#define QBNN_METRIC_SLOTS 34
__global__ __launch_bounds__(256) void classification_metrics_kernel(const float* __restrict__ probs, const int64_t* __restrict__ target,
                                                                      int B, int C, float* __restrict__ partials) {
  __shared__ float red[QBNN_METRIC_SLOTS][4];
  const int b = blockIdx.x * 256 + threadIdx.x;
  float v[QBNN_METRIC_SLOTS];
#pragma unroll
  for (int i = 0; i < QBNN_METRIC_SLOTS; ++i) v[i] = 0.f;
  if (b < B) {
    const float* p = probs + (int64_t)b * C;
    const int t = (int)target[b];
    int am = 0; float conf = p[0], brier = 0.f, ent = 0.f;
    for (int c = 0; c < C; ++c) {
      const float pc = p[c];
      if (pc > conf) { conf = pc; am = c; }
      const float oh = c == t ? 1.f : 0.f;
      brier += (pc - oh) * (pc - oh);
      ent += -pc * logf(pc + 1e-8f);
    }
    const float acc = am == t ? 1.f : 0.f;
    v[0] = 1.f - acc;
    v[1] = -logf(p[t] + 1e-8f);
    v[2] = brier;
    v[3] = ent;
    int bin = 0;                                   // boundaries k/10: right=True -> first k with k/10 > conf ... minus 1
    for (int k = 1; k <= 10; ++k) bin = (conf >= (float)k * 0.1f) ? k : bin;
    bin = min(bin, 9);
    v[4 + bin] = 1.f; v[14 + bin] = conf; v[24 + bin] = acc;
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
  for (int i = 0; i < QBNN_METRIC_SLOTS; ++i) {
    float x = v[i];
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    if (lane == 0) red[i][wave] = x;
  }
  __syncthreads();
  if (threadIdx.x < QBNN_METRIC_SLOTS)
    partials[(int64_t)blockIdx.x * QBNN_METRIC_SLOTS + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

QBNN_EXPORT int qbnn_classification_metrics(const float* probs, const int64_t* target, int32_t B, int32_t C, float* partials, void* stream) {
  if (!probs || !target || !partials || B <= 0 || C <= 0) return fail(QBNN_E_INVALID, "qbnn_classification_metrics: bad argument%s");
  hipLaunchKernelGGL(classification_metrics_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, probs, target, B, C, partials);
  return check_launch("qbnn_classification_metrics");
}

// Regression metrics on the reduced MC output (reference src/metrics.py:119-230 fed by RegressionMetric.update :468-500):
//   partial sums per 256-row block of   nll = 0.5 log(2 pi var + 1e-8) + (t - mean)^2 / (2 var + 1e-8)   (:143),
//   squared error (:186), absolute error (:224).  Accumulated in fp64 by the caller.
__global__ __launch_bounds__(256) void regression_metrics_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                                                                  const float* __restrict__ target, int B, float* __restrict__ partials) {
  __shared__ float red[3][4];
  const int b = blockIdx.x * 256 + threadIdx.x;
  float v[3] = {0.f, 0.f, 0.f};
  if (b < B) {
    const float m = mean[b], vr = var ? var[b] : 1.0f, t = target[b];
    const float d = t - m;
    v[0] = 0.5f * logf(2.0f * 3.14159265358979323846f * vr + 1e-8f) + d * d / (2.0f * vr + 1e-8f);
    v[1] = d * d;
    v[2] = fabsf(d);
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float x = v[i];
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    if (lane == 0) red[i][wave] = x;
  }
  __syncthreads();
  if (threadIdx.x < 3) partials[(int64_t)blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

QBNN_EXPORT int qbnn_regression_metrics(const float* mean, const float* var, const float* target, int32_t B, float* partials, void* stream) {
  if (!mean || !target || !partials || B <= 0) return fail(QBNN_E_INVALID, "qbnn_regression_metrics: bad argument%s");
  hipLaunchKernelGGL(regression_metrics_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean, var, target, B, partials);
  return check_launch("qbnn_regression_metrics");
}

// Flatten (reference src/utils.py:40-47) of a channels-last activation into the reference's NCHW feature order:
// x [S][B][HW][C] -> y [S][B][C*HW], y[c * HW + p] = x[p * C + c].  Needed where a stochastic Linear follows a conv map:
// its noise stream is indexed by the reference's (c, h, w) column order.
__global__ __launch_bounds__(256) void flatten_nchw_kernel(const uint8_t* __restrict__ x, int64_t x_ss, int B, int HW, int C,
                                                            uint8_t* __restrict__ y, int64_t y_ss) {
  const int64_t total = (int64_t)B * HW * C;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int s = blockIdx.y;
  const int64_t b = idx / ((int64_t)HW * C);
  const int r = (int)(idx - b * HW * C);
  const int c = r / HW, p = r - c * HW;
  y[(int64_t)s * y_ss + idx] = x[(int64_t)s * x_ss + (b * HW + p) * C + c];
}

// the same on pitched rows (the small networks' fused kernels write one row per image, pitch a multiple of 16): y[b][c * HW + p] = x[b][p * C + c],
// bytes C * HW .. ldy - 1 of every output row are written 0
__global__ __launch_bounds__(256) void flatten_nchw_rows_kernel(const uint8_t* __restrict__ x, int64_t x_ss, int ldx, int B, int HW, int C,
                                                                 uint8_t* __restrict__ y, int64_t y_ss, int ldy) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * ldy) return;
  const int s = blockIdx.y;
  const int b = (int)(idx / ldy), r = (int)(idx - (int64_t)b * ldy);
  uint8_t v = 0;
  if (r < C * HW) {
    const int c = r / HW, p = r - c * HW;
    v = x[(int64_t)s * x_ss + (int64_t)b * ldx + p * C + c];
  }
  y[(int64_t)s * y_ss + idx] = v;
}

QBNN_EXPORT int qbnn_flatten_nchw_rows_mc(const uint8_t* x, int64_t x_ss, int32_t ldx, int32_t B, int32_t HW, int32_t C, uint8_t* y, int64_t y_ss,
                                          int32_t ldy, int32_t n_samples, void* stream) {
  if (!x || !y || B <= 0 || HW <= 0 || C <= 0 || n_samples <= 0 || ldx < HW * C || ldy < HW * C)
    return fail(QBNN_E_INVALID, "qbnn_flatten_nchw_rows_mc: bad argument%s");
  const int64_t total = (int64_t)B * ldy;
  hipLaunchKernelGGL(flatten_nchw_rows_kernel, dim3((unsigned)((total + 255) / 256), n_samples), dim3(256), 0, (hipStream_t)stream,
                     x, x_ss, ldx, B, HW, C, y, y_ss, ldy);
  return check_launch("qbnn_flatten_nchw_rows_mc");
}

QBNN_EXPORT int qbnn_flatten_nchw_mc(const uint8_t* x, int64_t x_ss, int32_t B, int32_t HW, int32_t C, uint8_t* y, int64_t y_ss,
                                     int32_t n_samples, void* stream) {
  if (!x || !y || B <= 0 || HW <= 0 || C <= 0 || n_samples <= 0) return fail(QBNN_E_INVALID, "qbnn_flatten_nchw_mc: bad argument%s");
  const int64_t total = (int64_t)B * HW * C;
  hipLaunchKernelGGL(flatten_nchw_kernel, dim3((unsigned)((total + 255) / 256), n_samples), dim3(256), 0, (hipStream_t)stream,
                     x, x_ss, B, HW, C, y, y_ss);
  return check_launch("qbnn_flatten_nchw_mc");
}
