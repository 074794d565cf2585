// Sustained v_mfma_i32_32x32x32_i8 rate of the whole chip (256 workgroups x 8 waves, 6 independent accumulators per wave,
// nothing but MFMAs in the loop) with all-zero and with random int8 operands.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_sustained.hip
// Result on the MI355X used in round 1: profiles/r01_mfma_sustained.txt (4.7 POP/s on zeros, 3.26 POP/s on random data).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
// 6 independent accumulators, operands from memory (random or zero), 2 waves per SIMD, all CUs
__global__ __launch_bounds__(512) void k(const v4i* __restrict__ src, int* out, int iters) {
  v4i a[3], b[2];
  for (int i = 0; i < 3; ++i) a[i] = src[(threadIdx.x + 512 * i) & 4095];
  for (int i = 0; i < 2; ++i) b[i] = src[(threadIdx.x * 7 + 512 * (i + 3)) & 4095];
  v16i c[6] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) c[i * 2 + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], c[i * 2 + j], 0, 0, 0);
    // rotate operands a little so they are not loop-invariant constants for the data path
    v4i t = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = t;
  }
  int s = 0;
  for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
  v4i* src; int* out; hipMalloc(&src, 4096 * 16); hipMalloc(&out, 256 * 512 * 4);
  int* h = (int*)malloc(4096 * 16);
  for (int mode = 0; mode < 2; ++mode) {
    for (int i = 0; i < 4096 * 4; ++i) h[i] = mode ? rand() ^ (rand() << 16) : 0;
    hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
    for (int iters : {2000, 20000, 100000}) {
      k<<<256, 512>>>(src, out, 100); hipDeviceSynchronize();
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0); k<<<256, 512>>>(src, out, iters); hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double mf = 6.0 * iters * 2;   // MFMAs per SIMD
      printf("%s data, %6d iters: %.3f ms -> %.2f ns per MFMA per SIMD = %.1f cycles @2.4GHz, %.2f POP/s\n", mode ? "random" : "zero  ", iters, ms,
             ms * 1e6 / mf, ms * 1e6 / mf * 2.4, 256 * 4 * mf * 65536.0 / (ms * 1e-3) / 1e15);
    }
  }
  return 0;
}
