// Microbenchmark: issue cost (cycles per wave64 instruction per SIMD) of the vector instructions the exact requantisation epilogue is
// made of, on gfx950 with every SIMD busy (2 waves per SIMD, 16 independent registers per lane, no memory traffic).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_bench.hip -o tools/_build/valu_rate_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

#define OP16(ASM)                                                      \
  _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(v[i]) : "v"(a), "v"(b));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, float a, float b, int iters) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = (float)(threadIdx.x + i) * 0.37f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { OP16("v_fma_f32 %0, %1, %2, %0") }
    if (MODE == 1) { OP16("v_mul_f32 %0, %0, %2") }
    if (MODE == 2) { OP16("v_cvt_f32_i32 %0, %0") }
    if (MODE == 3) { OP16("v_cvt_pk_u8_f32 %0, %1, 1, %0") }
    if (MODE == 4) { OP16("v_min_f32 %0, %0, %2") }
    if (MODE == 5) { OP16("v_med3_f32 %0, %0, %1, %2") }
    if (MODE == 6) { OP16("v_rndne_f32 %0, %0") }
    if (MODE == 7) { OP16("v_sub_u32 %0, %0, %1") }
    if (MODE == 8) { OP16("v_add_f32 %0, %0, %2") }
    if (MODE == 9) { OP16("v_cvt_f32_ubyte1 %0, %0") }
    if (MODE == 10) { OP16("v_bfe_i32 %0, %0, 8, 8") }
    if (MODE == 12) { OP16("v_add_u32 %0, %0, %1") }
    if (MODE == 13) { OP16("v_add3_u32 %0, %0, %1, %2") }
    if (MODE == 14) { OP16("v_and_b32 %0, %0, %1") }
    if (MODE == 15) { OP16("v_lshlrev_b32 %0, 3, %0") }
    if (MODE == 16) { OP16("v_perm_b32 %0, %0, %1, %2") }
    if (MODE == 17) { OP16("v_min_i32 %0, %0, %1") }
    if (MODE == 18) { OP16("v_med3_i32 %0, %0, %1, %2") }
    if (MODE == 19) { OP16("v_lshl_or_b32 %0, %0, 8, %1") }
    if (MODE == 20) { OP16("v_bfi_b32 %0, %1, %0, %2") }
    if (MODE == 21) { OP16("v_cvt_f32_i32_sdwa %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1") }
    if (MODE == 22) { OP16("v_max_f32 %0, %0, %2") }
    if (MODE == 23) { OP16("v_sub_f32 %0, %0, %2") }
    if (MODE == 24) { OP16("v_dot4_i32_i8 %0, %1, %2, %0") }
    if (MODE == 25) { OP16("v_or_b32 %0, %0, %1") }
    if (MODE == 26) { OP16("v_mul_lo_u32 %0, %0, %1") }
    if (MODE == 27) { OP16("v_cvt_i32_f32 %0, %0") }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, float* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, 1.0001f, 0.9999f, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  // per SIMD: 2 waves x 16 instructions x iters wave-instructions in `best` ms
  const double instr_per_simd = 2.0 * 16 * iters;
  printf("%-22s %.3f ms  -> %.2f cycles per wave-instruction at 2.4 GHz\n", name, best, best * 1e-3 * 2.4e9 / instr_per_simd);
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * 4);
  run<0>("v_fma_f32", out);
  run<1>("v_mul_f32", out);
  run<8>("v_add_f32", out);
  run<7>("v_sub_u32", out);
  run<4>("v_min_f32", out);
  run<5>("v_med3_f32", out);
  run<6>("v_rndne_f32", out);
  run<2>("v_cvt_f32_i32", out);
  run<9>("v_cvt_f32_ubyte1", out);
  run<3>("v_cvt_pk_u8_f32", out);
  run<10>("v_bfe_i32", out);
  run<12>("v_add_u32", out);
  run<13>("v_add3_u32", out);
  run<14>("v_and_b32", out);
  run<25>("v_or_b32", out);
  run<15>("v_lshlrev_b32", out);
  run<19>("v_lshl_or_b32", out);
  run<20>("v_bfi_b32", out);
  run<16>("v_perm_b32", out);
  run<17>("v_min_i32", out);
  run<18>("v_med3_i32", out);
  run<22>("v_max_f32", out);
  run<23>("v_sub_f32", out);
  run<21>("v_cvt_f32_i32_sdwa", out);
  run<27>("v_cvt_i32_f32", out);
  run<24>("v_dot4_i32_i8", out);
  run<26>("v_mul_lo_u32", out);
  return 0;
}
