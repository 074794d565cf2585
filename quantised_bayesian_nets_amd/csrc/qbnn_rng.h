// Device-side Philox4x32-10 + deterministic fp32 Box-Muller for gfx950.
//
// The stream definition (counter / key layout, u32 -> normal mapping) is the
// build's own: the reference draws eps from torch's global generator
// (reference src/models/stochastic/bbb/quantized/conv_q.py:113, linear_q.py:86,
// bbb/conv.py:34-35, bbb/linear.py:44-45), which no GPU kernel can replay.
// Every fp32 operation below is an IEEE add / mul / fma / sqrt, written out
// explicitly (this translation unit is compiled with -ffp-contract=off), so the
// bits equal those of the x86 restatement used by the parity tests.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qbnn {

struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32x32->64 product per multiplier (v_mad_u64_u32): mul_lo + mul_hi would be two quarter-rate instructions
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    // three-input xor in one instruction (v_bitop3_b32, truth table 0x96); the compiler emits two v_xor_b32 otherwise
    const uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96);
    const uint32_t n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return {c0, c1, c2, c3};
}

// log(x), x in (0,1]; Cephes logf polynomial, fma-only.
__device__ __forceinline__ float det_logf(float x) {
  uint32_t u = __float_as_uint(x);
  int e = (int)((u >> 23) & 0xffu) - 126;
  float m = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
  if (m < 0.70710678118654752440f) { e -= 1; m = m + m; }
  const float t = m - 1.0f;
  const float z = t * t;
  float y = 7.0376836292E-2f;
  y = __builtin_fmaf(y, t, -1.1514610310E-1f);
  y = __builtin_fmaf(y, t, 1.1676998740E-1f);
  y = __builtin_fmaf(y, t, -1.2420140846E-1f);
  y = __builtin_fmaf(y, t, 1.4249322787E-1f);
  y = __builtin_fmaf(y, t, -1.6668057665E-1f);
  y = __builtin_fmaf(y, t, 2.0000714765E-1f);
  y = __builtin_fmaf(y, t, -2.4999993993E-1f);
  y = __builtin_fmaf(y, t, 3.3333331174E-1f);
  y = y * t * z;
  const float fe = (float)e;
  y = __builtin_fmaf(-2.12194440e-4f, fe, y);
  y = __builtin_fmaf(-0.5f, z, y);
  float r = t + y;
  r = __builtin_fmaf(0.693359375f, fe, r);
  return r;
}

// sin, cos of 2*pi*u, u = k * 2^-24.
__device__ __forceinline__ void det_sincos2pi(float u, float& s, float& c) {
  const float t = u * 4.0f;
  const int q = (int)t;
  const float f = t - (float)q;
  const bool flip = f > 0.5f;
  const float g = flip ? 1.0f - f : f;
  const float a = g * 1.57079632679489661923f;
  const float z = a * a;
  float ps = -1.9515295891E-4f;
  ps = __builtin_fmaf(ps, z, 8.3321608736E-3f);
  ps = __builtin_fmaf(ps, z, -1.6666654611E-1f);
  const float sn = __builtin_fmaf(ps * z, a, a);
  float pc = 2.443315711809948E-5f;
  pc = __builtin_fmaf(pc, z, -1.388731625493765E-3f);
  pc = __builtin_fmaf(pc, z, 4.166664568298827E-2f);
  const float cs = __builtin_fmaf(pc, z * z, __builtin_fmaf(-0.5f, z, 1.0f));
  const float s0 = flip ? cs : sn;
  const float c0 = flip ? sn : cs;
  switch (q & 3) {
    case 0: s = s0;  c = c0;  break;
    case 1: s = c0;  c = -s0; break;
    case 2: s = -s0; c = -c0; break;
    default: s = -c0; c = s0; break;
  }
}

// Four N(0,1) values from one Philox block.
__device__ __forceinline__ void normal4(const u32x4& r, float out[4]) {
  const float k = 5.9604644775390625e-8f;   // 2^-24
  {
    const float u1 = (float)((r.x >> 8) + 1u) * k;
    const float u2 = (float)(r.y >> 8) * k;
    const float rad = __builtin_sqrtf(-2.0f * det_logf(u1));
    float s, c;
    det_sincos2pi(u2, s, c);
    out[0] = rad * c; out[1] = rad * s;
  }
  {
    const float u1 = (float)((r.z >> 8) + 1u) * k;
    const float u2 = (float)(r.w >> 8) * k;
    const float rad = __builtin_sqrtf(-2.0f * det_logf(u1));
    float s, c;
    det_sincos2pi(u2, s, c);
    out[2] = rad * c; out[3] = rad * s;
  }
}

}  // namespace qbnn
