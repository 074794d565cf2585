// Microbenchmark (round 5): issue cost of further gfx950 vector instructions at 1 / 2 / 4 / 8 waves per SIMD, measured like
// tools/issue_bench.hip (s_memtime inside the kernel, every CU busy, 16 independent registers per wave so that latency is hidden),
// and of candidate forms of the requantisation epilogue built from the cheaper ones.
//   hipcc --offload-arch=gfx950 -O3 tools/issue_bench2.hip -o tools/_build/issue_bench2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <string.h>
#include <math.h>
#include <stdint.h>

#define OPS(X) \
  X(0, "v_add_f32 %0, %0, %2") X(1, "v_min_f32 %0, %0, %2") X(2, "v_max_f32 %0, %0, %2") X(3, "v_min_u32 %0, %0, %3") X(4, "v_min_i32 %0, %0, %3") \
  X(5, "v_min_u16 %0, %0, %3") X(6, "v_pk_min_u16 %0, %0, %3") X(7, "v_pk_max_i16 %0, %0, %3") X(8, "v_lshl_or_b32 %0, %0, 8, %3") \
  X(9, "v_and_or_b32 %0, %0, %3, %4") X(10, "v_and_b32 %0, %0, %3") X(11, "v_cvt_f32_ubyte0 %0, %0") X(12, "v_cvt_f32_ubyte2 %0, %0") \
  X(13, "v_rndne_f32 %0, %0") X(14, "v_cvt_pk_i16_i32 %0, %0, %3") X(15, "v_sat_pk_u8_i16 %0, %0") X(16, "v_min3_f32 %0, %0, %1, %2") \
  X(17, "v_pk_min_f16 %0, %0, %3") X(18, "v_pack_b32_f16 %0, %0, %3") X(19, "v_bfi_b32 %0, %3, %0, %4") X(20, "v_mov_b32 %0, %3") \
  X(21, "v_add3_u32 %0, %0, %3, %4") X(22, "v_mad_u32_u24 %0, %0, %3, %4") X(23, "v_lshlrev_b32 %0, 3, %0") X(24, "v_add_u32 %0, %0, %3") \
  X(25, "v_cvt_u32_f32 %0, %0") X(26, "v_cvt_i32_f32 %0, %0") X(27, "v_mul_f32 %0, %0, %2 clamp") X(28, "v_sub_f32 %0, %0, %2") \
  X(29, "v_cvt_pk_u8_f32 %0, %1, 2, %0") X(30, "v_med3_i32 %0, %0, %3, %4") X(31, "v_cvt_f32_i32 %0, %0") X(32, "v_fma_f32 %0, %1, %2, %0") \
  X(33, "v_mul_f32 %0, %0, %2") X(34, "v_max_i16 %0, %0, %3") X(35, "v_pk_add_u16 %0, %0, %3") X(36, "v_perm_b32 %0, %0, %3, %4") \
  X(37, "v_cvt_f16_f32 %0, %0") X(38, "v_xor_b32 %0, %0, %3") X(39, "v_fmac_f32 %0, %1, %2") X(40, "v_minimum3_f32 %0, %0, %1, %2") \
  X(41, "v_cvt_pkrtz_f16_f32 %0, %0, %1") X(42, "v_cvt_pknorm_u16_f32 %0, %0, %1") X(43, "v_min_f32 %0, %0, %5") X(44, "v_add_f32 %0, %0, %5") \
  X(45, "v_min_i16_sdwa %0, %0, %3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0") \
  X(46, "v_max_i16_sdwa %0, %0, %3 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:DWORD") \
  X(47, "v_or_b32_sdwa %0, %3, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2") \
  X(48, "v_min_i16_sdwa %0, %0, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0") \
  X(49, "v_cndmask_b32 %0, %0, %3, vcc") X(50, "v_readlane_b32 s20, %0, 3") X(51, "v_dot4c_i32_i8 %0, %3, %4") X(52, "v_mul_lo_u32 %0, %0, %3") \
  X(53, "v_mul_hi_u32 %0, %0, %3") X(54, "v_mul_u32_u24 %0, %0, %3") X(55, "v_or_b32 %0, %0, %3") X(56, "v_sub_u32 %0, %0, %3") X(57, "v_lshrrev_b32 %0, 3, %0") \
  X(58, "v_max_u16 %0, %0, %3") X(59, "v_min_f16 %0, %0, %3") X(60, "v_add_f16 %0, %0, %3") X(61, "v_lshlrev_b16 %0, 3, %0") X(62, "v_ldexp_f32 %0, %0, %3") \
  X(63, "v_mul_f32_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD") X(64, "v_bfe_u32 %0, %0, 8, 8") X(65, "v_mul_i32_i24 %0, %0, %3")
constexpr int NOPS = 66, EPI_M5 = 100, EPI_PKMIN = 101, EPI_M5_S = 102, EPI_SDWA = 103;

template <int MODE, int WPS>
__global__ __launch_bounds__(256 * WPS) void k(unsigned long long* stamps, float* out, float a, float b, int c, int d, int iters) {
  int v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = (int)threadIdx.x * 3 + i;
  float s_ = b;       // a wave-uniform operand (SGPR)
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#define X(ID, ASM) if (MODE == ID) { _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(v[i]) : "v"(a), "v"(b), "v"(c), "v"(d), "s"(s_)); }
    OPS(X)
#undef X
    if (MODE == EPI_M5 || MODE == EPI_M5_S) {       // sub_f32, fma, mul, min, cvt_pk_u8: 5 per output, 16 outputs -> 4 dwords
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int r = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int x = v[4 * g + j];
          asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(a));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
          if (MODE == EPI_M5_S) { asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x) : "s"(s_)); asm volatile("v_min_f32 %0, %1, %0" : "+v"(x) : "s"(s_)); }
          else { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b)); asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(a)); }
          if (j == 0) asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(r) : "v"(x));
          if (j == 1) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(r) : "v"(x));
          if (j == 2) asm volatile("v_cvt_pk_u8_f32 %0, %1, 2, %0" : "+v"(r) : "v"(x));
          if (j == 3) asm volatile("v_cvt_pk_u8_f32 %0, %1, 3, %0" : "+v"(r) : "v"(x));
        }
        v[4 * g] = r;
      }
    }
    if (MODE == EPI_SDWA) {     // sub_f32, fma, mul, v_min_i16_sdwa on the float's high half (upper clamp, see check_sdwa_clamp), cvt_pk_u8
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int r = 0;
        int x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          x[j] = v[4 * g + j];
          asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[j]) : "v"(a));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[j]) : "v"(a), "v"(b));
          asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(b));
        }
        asm volatile("v_min_i16_sdwa %0, %0, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0\n"
                     "v_min_i16_sdwa %1, %1, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0\n"
                     "v_min_i16_sdwa %2, %2, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0\n"
                     "v_min_i16_sdwa %3, %3, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0\n"
                     "v_cvt_pk_u8_f32 %4, %0, 0, %4\nv_cvt_pk_u8_f32 %4, %1, 1, %4\nv_cvt_pk_u8_f32 %4, %2, 2, %4\nv_cvt_pk_u8_f32 %4, %3, 3, %4"
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(r) : "v"(c));
        v[4 * g] = r;
      }
    }
    if (MODE == EPI_PKMIN) {    // sub_f32, fma, mul, cvt_pk_u8 into bytes 0 / 2 of two registers, 2 x v_pk_min_u16, v_lshl_or_b32: 3 + 1 + 3/4 per output
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int ra = 0, rb = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int x = v[4 * g + j];
          asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(a));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
          asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b));
          if (j == 0) asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(ra) : "v"(x));
          if (j == 1) asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(rb) : "v"(x));
          if (j == 2) asm volatile("v_cvt_pk_u8_f32 %0, %1, 2, %0" : "+v"(ra) : "v"(x));
          if (j == 3) asm volatile("v_cvt_pk_u8_f32 %0, %1, 2, %0" : "+v"(rb) : "v"(x));
        }
        asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(ra) : "v"(c));
        asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(rb) : "v"(c));
        asm volatile("v_lshl_or_b32 %0, %1, 8, %0" : "+v"(ra) : "v"(rb));
        v[4 * g] = ra;
      }
    }
  }
  asm volatile("s_nop 0" ::: "memory");
  int s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + wave;
    stamps[2 * w] = t1 - t0;
    stamps[2 * w + 1] = r1 - r0;
  }
}

static unsigned long long* d_st; static float* d_out;

// min(v, hi) in front of a round-to-nearest-even conversion, as ONE 16-bit integer minimum on the float's HIGH half: for an integer hi with
// hi + 0.5 representable in 8 significant bits (every hi <= 127), c16 = high half of (float)(hi + 0.5) - 1 ulp16 ... see the kernels.
__global__ void sdwa_clamp_kernel(const float* in, const int* c16, uint32_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = in[i];
  uint32_t r = 0;
  const int c = c16[0];
  asm volatile("v_min_i16_sdwa %0, %0, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0\n"
               "s_nop 0\n"
               "v_cvt_pk_u8_f32 %1, %0, 0, %1" : "+v"(x), "+v"(r) : "v"(c));
  out[i] = r;
}
static void check_sdwa_clamp() {
  const int n = 1 << 22;
  std::vector<float> h(n);
  float *d_in; int* d_c; uint32_t* d_o;
  (void)hipMalloc(&d_in, n * 4); (void)hipMalloc(&d_c, 4); (void)hipMalloc(&d_o, n * 4);
  std::vector<uint32_t> o(n);
  long bad = 0, cases = 0;
  for (int hi = 0; hi <= 127; ++hi) {
    const float lim = (float)hi + 0.5f;
    uint32_t lb; memcpy(&lb, &lim, 4);
    const int c16 = (int)(lb >> 16) - 1;                 // the largest high half whose every float is below hi + 0.5
    if ((lb & 0xffffu) != 0) { printf("hi %d: hi + 0.5 does not fit the high half\n", hi); continue; }
    for (int i = 0; i < n; ++i) {
      const int k = i & 7;
      float v;
      if (k == 0) v = (float)hi + (float)((i >> 3) % 2001 - 1000) * 1e-3f;              // around the limit
      else if (k == 1) { uint32_t b = lb + (uint32_t)((i >> 3) % 4001) - 2000u; memcpy(&v, &b, 4); }      // +-2000 ulps around hi + 0.5
      else if (k == 2) v = (float)((double)rand() / RAND_MAX * 300.0 - 40.0);
      else if (k == 3) v = (float)((double)rand() / RAND_MAX * 4e6 - 2e6);
      else if (k == 4) v = (float)((i >> 3) % 600) * 0.5f - 20.f;                         // ties
      else if (k == 5) { uint32_t b = (uint32_t)rand() ^ ((uint32_t)rand() << 16); b &= 0xCFFFFFFFu; memcpy(&v, &b, 4); if (v != v) v = 1.f; }     // any finite pattern up to 2^33
      else if (k == 6) v = -(float)((double)rand() / RAND_MAX * 3.0);
      else v = (float)hi + 0.5f - (float)((i >> 3) % 50) * 1e-5f;
      h[i] = v;
    }
    (void)hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_c, &c16, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sdwa_clamp_kernel, dim3(n / 256), dim3(256), 0, 0, d_in, d_c, d_o, n);
    (void)hipMemcpy(o.data(), d_o, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) {
      float v = h[i] < (float)hi ? h[i] : (float)hi;        // v_min_f32, then v_cvt_pk_u8_f32 (round to nearest even, saturating at 0 and 255)
      float r = nearbyintf(v);
      r = r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
      ++cases;
      if ((uint32_t)r != (o[i] & 0xffu)) { if (bad < 5) printf("hi %d v %.9g (%08x): want %u got %u\n", hi, h[i], *(uint32_t*)&h[i], (uint32_t)r, o[i] & 0xff); ++bad; }
    }
  }
  printf("sdwa upper clamp against v_min_f32 + v_cvt_pk_u8_f32: %ld cases, %ld differ\n", cases, bad);
}

template <int MODE, int WPS>
static double run1(int blocks_per_cu, int iters, int vinst) {
  const int nblk = 256 * blocks_per_cu, nwav_blk = 4 * WPS, nw = nblk * nwav_blk;
  std::vector<unsigned long long> st(2 * (size_t)nw);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<MODE, WPS>), dim3(nblk), dim3(256 * WPS), 0, 0, d_st, d_out, 1.0001f, 0.9999f, 0x007f007f, 0x00ff00ff, iters);
    (void)hipDeviceSynchronize();
  }
  (void)hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cs;
  for (int i = 0; i < nw; ++i) cs.push_back((double)st[2 * i]);
  std::sort(cs.begin(), cs.end());
  return cs[cs.size() / 2] / iters / (vinst * WPS * blocks_per_cu);
}
template <int MODE>
static void run(const char* name, int vinst = 16, int iters = 2000) {
  printf("%-44s %5.2f %5.2f %5.2f %5.2f\n", name, run1<MODE, 1>(1, iters, vinst), run1<MODE, 2>(1, iters, vinst), run1<MODE, 4>(1, iters, vinst), run1<MODE, 4>(2, iters, vinst));
  fflush(stdout);
}

int main() {
  (void)hipMalloc(&d_st, 2 * 8 * 512 * 16 * sizeof(unsigned long long));
  (void)hipMalloc(&d_out, 512 * 1024 * 4);
  printf("SIMD cycles per wave-instruction at %-15s %5d %5d %5d %5d\n", "waves/SIMD =", 1, 2, 4, 8);
#define X(ID, ASM) run<ID>(ASM);
  OPS(X)
#undef X
  run<EPI_M5>("epilogue: sub_f32 fma mul min cvt_pk (x16)", 80, 1000);
  run<EPI_M5_S>("... mul and min with an SGPR operand", 80, 1000);
  run<EPI_PKMIN>("epilogue: sub_f32 fma mul cvt_pk + pk_min_u16 (76 for 16)", 76, 1000);
  run<EPI_SDWA>("epilogue: sub_f32 fma mul min_i16_sdwa cvt_pk", 80, 1000);
  check_sdwa_clamp();
  return 0;
}
