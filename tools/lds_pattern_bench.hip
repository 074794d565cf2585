// Microbenchmark (round 5): what an LDS instruction costs for a GIVEN per-lane address pattern on gfx950 -- the patterns of the fused kernels' tile
// reads / writes and candidate re-layouts are measured instead of modelled (tools/lds_conflicts.py holds the model for ds_read_b128 only).
// 16 waves per workgroup (the LDS pipe saturated), one workgroup per CU, every wave the same pattern, s_memtime around the loop of wave 0:
// LDS-pipe cycles per wave-instruction = cycles / (16 x instructions per wave).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_pattern_bench.hip -o tools/_build/lds_pattern_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>
#include <algorithm>

typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
enum { R64, R2_64, R128, W32, W64, NMODE };
static const char* mode_name[NMODE] = {"ds_read_b64", "ds_read2_b64 (+0, +off1*8)", "ds_read_b128", "ds_write_b32", "ds_write_b64"};

template <int MODE>
__global__ __launch_bounds__(1024) void k(const uint32_t* __restrict__ offs, int off1, int iters, unsigned long long* cyc, int* sink) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) reinterpret_cast<uint32_t*>(lds)[i] = i;
  __syncthreads();
  const uint32_t a = offs[threadIdx.x & 63];
  int acc = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t ad = a + ((it + u) & 1) * 0;      // (same address every time: the pattern is what is measured)
      if (MODE == R64) { v2i v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(ad)); asm volatile("" :: "v"(v)); }
      if (MODE == R2_64) { v4i v; if (off1 == 1) asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(v) : "v"(ad)); else asm volatile("ds_read2_b64 %0, %1 offset1:102" : "=v"(v) : "v"(ad)); asm volatile("" :: "v"(v)); }
      if (MODE == R128) { v4i v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(ad)); asm volatile("" :: "v"(v)); }
      if (MODE == W32) { asm volatile("ds_write_b32 %0, %1" :: "v"(ad), "v"(acc)); }
      if (MODE == W64) { v2i v = {acc, acc}; asm volatile("ds_write_b64 %0, %1" :: "v"(ad), "v"(v)); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (threadIdx.x < 64) sink[blockIdx.x * 64 + threadIdx.x] = acc + lds[threadIdx.x];
}

static uint32_t* d_offs; static unsigned long long* d_cyc; static int* d_sink;
template <int MODE>
static double run(const std::vector<uint32_t>& offs, int off1 = 1) {
  const int iters = 500, nb = 64;
  (void)hipMemcpy(d_offs, offs.data(), 256, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(1024), 160 * 1024, 0, d_offs, off1, iters, d_cyc, d_sink);
    (void)hipDeviceSynchronize();
  }
  std::vector<unsigned long long> c(nb);
  (void)hipMemcpy(c.data(), d_cyc, nb * 8, hipMemcpyDeviceToHost);
  std::sort(c.begin(), c.end());
  return (double)c[nb / 2] / (iters * 8.0 * 16.0);
}
static std::vector<uint32_t> pat(uint32_t (*f)(int lane)) { std::vector<uint32_t> v(64); for (int l = 0; l < 64; ++l) v[l] = f(l); return v; }

int main() {
  (void)hipMalloc(&d_offs, 256); (void)hipMalloc(&d_cyc, 64 * 8); (void)hipMalloc(&d_sink, 64 * 64 * 4);
  struct P { const char* name; std::vector<uint32_t> o; };
  std::vector<P> ps;
  ps.push_back({"contiguous 8 B per lane", pat([](int l) { return (uint32_t)l * 8; })});
  ps.push_back({"contiguous 16 B per lane", pat([](int l) { return (uint32_t)l * 16; })});
  ps.push_back({"all lanes one address", pat([](int l) { return 0u; })});
  ps.push_back({"stride 256 B (one bank)", pat([](int l) { return (uint32_t)l * 256; })});
  ps.push_back({"stride 128 B", pat([](int l) { return (uint32_t)l * 128; })});
  ps.push_back({"stride 24 B (layer 1: pixel r, k-half h at +16)", pat([](int l) { return (uint32_t)((l & 31) * 24 + 16 * (l >> 5)); })});
  ps.push_back({"stride 48 B, 16 columns x 2 images (image stride 27744), k-half +16  [down 24->48 today]", pat([](int l) { int r = l & 31; return (uint32_t)((r >> 4) * 27744 + (r & 15) * 48 + 16 * (l >> 5)); })});
  ps.push_back({"... image stride 27744 + 8", pat([](int l) { int r = l & 31; return (uint32_t)((r >> 4) * 27752 + (r & 15) * 48 + 16 * (l >> 5)); })});
  ps.push_back({"... image stride 27744 + 16", pat([](int l) { int r = l & 31; return (uint32_t)((r >> 4) * 27760 + (r & 15) * 48 + 16 * (l >> 5)); })});
  ps.push_back({"... image stride 27744 + 32", pat([](int l) { int r = l & 31; return (uint32_t)((r >> 4) * 27776 + (r & 15) * 48 + 16 * (l >> 5)); })});
  ps.push_back({"... image stride 27744 + 64", pat([](int l) { int r = l & 31; return (uint32_t)((r >> 4) * 27808 + (r & 15) * 48 + 16 * (l >> 5)); })});
  ps.push_back({"... image stride 27744 + 128", pat([](int l) { int r = l & 31; return (uint32_t)((r >> 4) * 27872 + (r & 15) * 48 + 16 * (l >> 5)); })});
  ps.push_back({"stride 48 B, lanes 8 columns x 2 images interleaved per 16 (image stride + 8)", pat([](int l) { int r = l & 31; int img = (r >> 3) & 1, col = (r & 7) | ((r >> 4) << 3); return (uint32_t)(img * 27752 + col * 48 + 16 * (l >> 5)); })});
  ps.push_back({"stride 48 B, 32 columns of one image, k-half +16", pat([](int l) { return (uint32_t)((l & 31) * 48 + 16 * (l >> 5)); })});
  ps.push_back({"stride 48 B, one image, k-half at +8192", pat([](int l) { return (uint32_t)((l & 31) * 48 + 8192 * (l >> 5)); })});
  ps.push_back({"stride 40 B", pat([](int l) { return (uint32_t)((l & 31) * 40 + 16 * (l >> 5)); })});
  ps.push_back({"stride 56 B", pat([](int l) { return (uint32_t)((l & 31) * 56 + 16 * (l >> 5)); })});
  ps.push_back({"stride 72 B", pat([](int l) { return (uint32_t)((l & 31) * 72 + 16 * (l >> 5)); })});
  ps.push_back({"stride 96 B (48-channel tile, stride 2)", pat([](int l) { return (uint32_t)((l & 31) * 96 + 16 * (l >> 5)); })});
  ps.push_back({"stride 104 B", pat([](int l) { return (uint32_t)((l & 31) * 104 + 16 * (l >> 5)); })});
  ps.push_back({"stride 112 B (dense T tile of the 96-channel blocks)", pat([](int l) { return (uint32_t)((l & 31) * 112 + 16 * (l >> 5)); })});
  printf("%-100s %12s %12s %12s %12s %12s\n", "cycles per wave-instruction", "read_b64", "read2_b64", "read_b128", "write_b32", "write_b64");
  for (auto& p : ps)
    printf("%-100s %12.1f %12.1f %12.1f %12.1f %12.1f\n", p.name, run<R64>(p.o), run<R2_64>(p.o), run<R128>(p.o), run<W32>(p.o), run<W64>(p.o));
  // which lanes share a conflict domain: lane i and lane j at the SAME bank (addresses 256 B apart), all other lanes on distinct banks
  printf("\npairwise (ds_read_b64): lane j's address put 256 B above lane 0's, the rest contiguous 8 B -- cycles per instruction by j:\n");
  for (int j = 1; j < 64; ++j) {
    std::vector<uint32_t> o(64);
    for (int l = 0; l < 64; ++l) o[l] = 4096 + l * 8;
    o[0] = 0; o[j] = 256;
    printf("%5.1f%s", run<R64>(o), (j % 16 == 15) ? "\n" : " ");
  }
  printf("\npairwise (ds_read_b128): lane j 256 B above lane 0, the rest contiguous 16 B:\n");
  for (int j = 1; j < 64; ++j) {
    std::vector<uint32_t> o(64);
    for (int l = 0; l < 64; ++l) o[l] = 8192 + l * 16;
    o[0] = 0; o[j] = 256;
    printf("%5.1f%s", run<R128>(o), (j % 16 == 15) ? "\n" : " ");
  }
  printf("\npairwise (ds_write_b32): lane j 256 B above lane 0, the rest contiguous 4 B:\n");
  for (int j = 1; j < 64; ++j) {
    std::vector<uint32_t> o(64);
    for (int l = 0; l < 64; ++l) o[l] = 8192 + l * 4;
    o[0] = 0; o[j] = 256;
    printf("%5.1f%s", run<W32>(o), (j % 16 == 15) ? "\n" : " ");
  }
  printf("\n");
  return 0;
}
