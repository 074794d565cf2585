// Microbenchmark: per-element throughput of v_fma_f32 / v_mul_f32 against v_pk_fma_f32 / v_pk_mul_f32 on gfx950, every SIMD busy
// (2 waves per SIMD like the fused kernels), 16 independent chains per lane.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, float a, float b, int iters) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = (float)(threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {            // scalar: fma then mul per element
#pragma unroll
      for (int i = 0; i < 16; ++i) {        // inline asm: the SLP vectoriser would pack plain C into v_pk_* by itself here
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      }
    } else {                    // packed: two elements per instruction
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        f2 x = {v[i], v[i + 1]};
        const f2 aa = {a, a}, bb = {b, b};
        x = __builtin_elementwise_fma(aa, bb, x);
        x = x * bb;
        v[i] = x.x; v[i + 1] = x.y;
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, 1.0001f, 0.9999f, iters);
      else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, 1.0001f, 0.9999f, iters);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double elem_ops = 256.0 * 512 * 16 * 2 * iters;      // (fma + mul) per element
      printf("%s: %.3f ms, %.2f T element-ops/s (peak scalar 39.3 T lane-ops/s)\n", mode ? "packed v_pk_fma_f32 + v_pk_mul_f32" : "scalar v_fma_f32 + v_mul_f32", ms, elem_ops / ms / 1e9);
    }
  }
  return 0;
}
