// Microbenchmark: what one SIMD of gfx950 issues per shader cycle, measured IN the kernel (s_memtime = shader cycles,
// s_memrealtime = 100 MHz wall clock), so the result does not depend on the clock the chip holds under load.
//   * per-instruction issue cost of the vector instructions of the exact requantisation epilogue at 1 / 2 / 4 / 8 waves per SIMD;
//   * the epilogue sequence itself (sub, cvt, fma, mul, min, cvt_pk per output);
//   * v_mfma_i32_32x32x32_i8 alone, and beside the epilogue: in the same wave (9 MFMAs + 72 vector instructions per tile, the
//     layer-1 shape) and in the partner waves of the SIMD (waves 0-3 multiply, waves 4-7 requantise).
// Every CU runs the same workgroup (grid = 256 x blocks per CU), so the clock is the one a full-chip kernel sees.
//   hipcc --offload-arch=gfx950 -O3 tools/issue_bench.hip -o tools/_build/issue_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4acc __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define OP16(ASM) _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(v[i]) : "v"(a), "v"(b));

enum Mode { FMA, ADD, MUL, SUBU, MIN, MED3, CVT_I2F, CVT_PK, PERM, PKFMA, PKMUL, EPI6, EPI5, MFMA, MIX_WAVE, MIX_ROLE, MIX_WAVE_INTER, MFMA16, MIX_WAVE16, MIX_WAVE_INDEP, NMODES };
static const char* mode_name[NMODES] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_sub_u32", "v_min_f32", "v_med3_f32", "v_cvt_f32_i32",
  "v_cvt_pk_u8_f32", "v_perm_b32", "v_pk_fma_f32", "v_pk_mul_f32", "epilogue x16 (6 ops: sub cvt fma mul min cvt_pk)",
  "epilogue x16 (5 ops: sub cvt fma min cvt_pk)", "v_mfma_i32_32x32x32_i8", "same wave: 9 MFMA then 72 VALU", "partner waves: 0-3 MFMA / 4-7 epilogue",
  "same wave: 9 x (MFMA + 8 VALU)", "v_mfma_i32_16x16x64_i8", "same wave: 18 MFMA 16x16x64 then 72 VALU", "same wave: 9 MFMA (3 accumulators) then 72 VALU"};

// one requantisation of 4 accumulator values into one dword: 6 ops per value (or 5 without the separate multiply)
template <bool MUL6>
__device__ __forceinline__ void requant4(int& a0, int& a1, int& a2, int& a3, int zwr, float bias, float rcp, float mult, float hi) {
  int* p[4] = {&a0, &a1, &a2, &a3};
  int r = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int x = *p[j];
    float f;
    asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(zwr));
    asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f) : "v"(x));
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f) : "v"(bias), "v"(rcp));
    if (MUL6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f) : "v"(mult));
    asm volatile("v_min_f32 %0, %0, %1" : "+v"(f) : "v"(hi));
    if (j == 0) asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(r) : "v"(f));
    if (j == 1) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(r) : "v"(f));
    if (j == 2) asm volatile("v_cvt_pk_u8_f32 %0, %1, 2, %0" : "+v"(r) : "v"(f));
    if (j == 3) asm volatile("v_cvt_pk_u8_f32 %0, %1, 3, %0" : "+v"(r) : "v"(f));
  }
  a0 = r;      // keep a dependence so nothing is dropped
}

template <int MODE, int WPS>     // WPS = waves per SIMD inside one workgroup (1, 2, 4)
__global__ __launch_bounds__(256 * WPS) void k(unsigned long long* stamps, float* out, const v4i* __restrict__ src, float a, float b, int iters) {
  float v[16];
  int q[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { v[i] = (float)(threadIdx.x + i) * 0.37f; q[i] = (int)threadIdx.x * 3 + i; }
  v4i wa = src[threadIdx.x & 1023], xb = src[(threadIdx.x * 7 + 5) & 1023];
  v16i acc = {}, accb = {}, accc = {};
  v4acc acc4 = {};
  const int wave = threadIdx.x >> 6;
  const int zwr = (int)threadIdx.x & 31;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == FMA) { OP16("v_fma_f32 %0, %1, %2, %0") }
    if (MODE == ADD) { OP16("v_add_f32 %0, %0, %2") }
    if (MODE == MUL) { OP16("v_mul_f32 %0, %0, %2") }
    if (MODE == SUBU) { OP16("v_sub_u32 %0, %0, %1") }
    if (MODE == MIN) { OP16("v_min_f32 %0, %0, %2") }
    if (MODE == MED3) { OP16("v_med3_f32 %0, %0, %1, %2") }
    if (MODE == CVT_I2F) { OP16("v_cvt_f32_i32 %0, %0") }
    if (MODE == CVT_PK) { OP16("v_cvt_pk_u8_f32 %0, %1, 1, %0") }
    if (MODE == PERM) { OP16("v_perm_b32 %0, %0, %1, %2") }
    if (MODE == PKFMA || MODE == PKMUL) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        f2 x = {v[i], v[i + 1]};
        const f2 aa = {a, a}, bb = {b, b};
        if (MODE == PKFMA) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(aa), "v"(bb));
        else asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(bb));
        v[i] = x.x; v[i + 1] = x.y;
      }
    }
    if (MODE == EPI6 || MODE == EPI5) {
#pragma unroll
      for (int g = 0; g < 4; ++g) requant4<MODE == EPI6>(q[4 * g], q[4 * g + 1], q[4 * g + 2], q[4 * g + 3], zwr, a, b, a, 127.0f);
    }
    if (MODE == MFMA) {
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wa, xb, acc, 0, 0, 0);
    }
    if (MODE == MIX_WAVE) {          // one layer-1 tile: 9 MFMAs into acc, the epilogue of the previous tile (12 outputs, 6 ops) from q
#pragma unroll
      for (int j = 0; j < 9; ++j) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wa, xb, acc, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 3; ++g) requant4<true>(q[4 * g], q[4 * g + 1], q[4 * g + 2], q[4 * g + 3], zwr, a, b, a, 127.0f);
    }
    if (MODE == MFMA16) {
      acc4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, xb, acc4, 0, 0, 0);
    }
    if (MODE == MIX_WAVE16) {        // the same MACs as 9 x 32x32x32 on the 16x16x64 shape, then the epilogue
#pragma unroll
      for (int j = 0; j < 18; ++j) acc4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, xb, acc4, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 3; ++g) requant4<true>(q[4 * g], q[4 * g + 1], q[4 * g + 2], q[4 * g + 3], zwr, a, b, a, 127.0f);
    }
    if (MODE == MIX_WAVE_INDEP) {    // 9 MFMAs into three independent accumulators
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wa, xb, acc, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_i32_32x32x32_i8(wa, xb, accb, 0, 0, 0);
        accc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wa, xb, accc, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 3; ++g) requant4<true>(q[4 * g], q[4 * g + 1], q[4 * g + 2], q[4 * g + 3], zwr, a, b, a, 127.0f);
    }
    if (MODE == MIX_WAVE_INTER) {    // the same work, source order MFMA, 8 VALU, MFMA, 8 VALU ...
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wa, xb, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (j % 3 == 0) requant4<true>(q[4 * (j / 3)], q[4 * (j / 3) + 1], q[4 * (j / 3) + 2], q[4 * (j / 3) + 3], zwr, a, b, a, 127.0f);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (MODE == MIX_ROLE) {          // waves 0..(4*WPS/2 - 1) multiply, the others requantise: 9 MFMAs against 72 vector instructions
      if (wave < 2 * WPS) {
#pragma unroll
        for (int j = 0; j < 9; ++j) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wa, xb, acc, 0, 0, 0);
      } else {
#pragma unroll
        for (int g = 0; g < 3; ++g) requant4<true>(q[4 * g], q[4 * g + 1], q[4 * g + 2], q[4 * g + 3], zwr, a, b, a, 127.0f);
      }
    }
  }
  asm volatile("s_nop 0" ::: "memory");
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i] + (float)q[i] + (float)acc[i] + (float)accb[i] + (float)accc[i] + (float)acc4[i & 3];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + wave;
    stamps[2 * w] = t1 - t0;
    stamps[2 * w + 1] = r1 - r0;
  }
}

static unsigned long long* d_st; static float* d_out; static v4i* d_src;

template <int MODE, int WPS>
static void run1(int blocks_per_cu, int iters, int vinst_per_iter, int mfma_per_iter) {
  const int nblk = 256 * blocks_per_cu, nwav_blk = 4 * WPS, nw = nblk * nwav_blk;
  std::vector<unsigned long long> st(2 * (size_t)nw);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k<MODE, WPS>), dim3(nblk), dim3(256 * WPS), 0, 0, d_st, d_out, d_src, 1.0001f, 0.9999f, iters);
    (void)hipDeviceSynchronize();
  }
  (void)hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
  // role split: report the two halves of each workgroup separately
  auto med = [&](int lo_wave, int hi_wave, double& cyc, double& clk) {
    std::vector<double> c, f;
    for (int bl = 0; bl < nblk; ++bl)
      for (int w = lo_wave; w < hi_wave; ++w) {
        const size_t i = (size_t)bl * nwav_blk + w;
        c.push_back((double)st[2 * i]);
        f.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 0.1);      // GHz
      }
    std::sort(c.begin(), c.end()); std::sort(f.begin(), f.end());
    cyc = c[c.size() / 2]; clk = f[f.size() / 2];
  };
  const int wps = WPS * blocks_per_cu;
  double cyc, clk;
  if (MODE == MIX_ROLE) {
    double c2, k2;
    med(0, nwav_blk / 2, cyc, clk); med(nwav_blk / 2, nwav_blk, c2, k2);
    printf("%-52s %d waves/SIMD: MFMA waves %.1f cyc per 9-MFMA tile (%.1f per MFMA), epilogue waves %.1f cyc per 72 instr (%.2f per instr); clock %.2f GHz\n",
           mode_name[MODE], wps, cyc / iters, cyc / iters / 9, c2 / iters, c2 / iters / 72, clk);
    return;
  }
  med(0, nwav_blk, cyc, clk);
  const double per_iter_simd = cyc / iters;       // cycles the SIMD spent per iteration of EVERY wave on it
  if (vinst_per_iter && !mfma_per_iter)
    printf("%-52s %d waves/SIMD: %.2f SIMD cycles per wave-instruction (one wave sees %.2f); clock %.2f GHz\n", mode_name[MODE], wps,
           per_iter_simd / (vinst_per_iter * wps), per_iter_simd / vinst_per_iter, clk);
  else if (mfma_per_iter && !vinst_per_iter)
    printf("%-52s %d waves/SIMD: %.2f SIMD cycles per MFMA; clock %.2f GHz -> %.2f POP/s\n", mode_name[MODE], wps, per_iter_simd / (mfma_per_iter * wps),
           clk, 1024.0 * 65536.0 / (per_iter_simd / (mfma_per_iter * wps)) * clk * 1e9 / 1e15);
  else
    printf("%-52s %d waves/SIMD: %.1f SIMD cycles per tile (9 MFMA = 288 + 72 VALU); clock %.2f GHz\n", mode_name[MODE], wps, per_iter_simd / wps, clk);
}

template <int MODE>
static void run(int vinst, int mfma, int iters = 4000) {
  run1<MODE, 1>(1, iters, vinst, mfma);
  run1<MODE, 2>(1, iters, vinst, mfma);
  run1<MODE, 4>(1, iters, vinst, mfma);
  run1<MODE, 4>(2, iters, vinst, mfma);
}

int main() {
  (void)hipMalloc(&d_st, 2 * 8 * 512 * 16 * sizeof(unsigned long long));
  (void)hipMalloc(&d_out, 512 * 1024 * 4);
  (void)hipMalloc(&d_src, 1024 * 16);
  std::vector<int> h(4096);
  for (auto& x : h) x = rand() ^ (rand() << 16);
  (void)hipMemcpy(d_src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  if (getenv("ISSUE_BENCH_QUICK")) {
    run<MFMA>(0, 1, 20000); run<MFMA16>(0, 1, 20000); run<MIX_WAVE>(72, 9, 1000); run<MIX_WAVE16>(72, 18, 1000); run<MIX_WAVE_INDEP>(72, 9, 1000);
    return 0;
  }
  run<FMA>(16, 0); run<ADD>(16, 0); run<MUL>(16, 0); run<SUBU>(16, 0); run<MIN>(16, 0); run<MED3>(16, 0);
  run<CVT_I2F>(16, 0); run<CVT_PK>(16, 0); run<PERM>(16, 0); run<PKFMA>(8, 0); run<PKMUL>(8, 0);
  run<EPI6>(96, 0, 1000); run<EPI5>(80, 0, 1000);
  run<MFMA>(0, 1, 20000);
  run<MIX_WAVE>(72, 9, 1000);
  run<MIX_WAVE_INTER>(72, 9, 1000);
  run<MFMA16>(0, 1, 20000);
  run<MIX_WAVE16>(72, 18, 1000);
  run<MIX_WAVE_INDEP>(72, 9, 1000);
  run1<MIX_ROLE, 2>(1, 1000, 72, 9);
  run1<MIX_ROLE, 4>(1, 1000, 72, 9);
  return 0;
}
