// Microbenchmark: do VALU instructions (softmax-style: v_exp_f32, v_pk_fma_f32, v_max3_f32, v_cvt_pk_f16_f32) overlap with a stream of
// 32x32x16 f16 MFMAs on one SIMD?  Cases per workgroup of 256 or 512 threads (wave w runs on SIMD w % 4):
//   mfma          every wave issues NM independent-chain MFMAs per iteration
//   valu          every wave issues NV VALU instructions per iteration
//   same wave     both streams interleaved in ONE wave (1 MFMA : NV / NM VALU)
//   two waves     waves 0-3 MFMA only, waves 4-7 VALU only (same SIMDs)
// If the matrix pipe and the VALU run side by side, "same wave" / "two waves" cost max(mfma, valu); if they exclude each other, the sum.
// build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/mvv tools/microbench/mfma_vs_valu.hip && /tmp/mvv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int VKIND, int NVI>   // MODE 0 mfma, 1 valu, 2 same wave, 3 two waves (512 threads); VKIND 0 = exp only, 1 = pk_fma only, 2 = softmax mix
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (f16)(0.001f * (lane + e)); b[e] = (f16)(0.002f * (lane - e)); }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float v[32]; f32x2 v2[16]; unsigned vi[32];
  for (int e = 0; e < 32; ++e) { v[e] = 0.01f * (lane + e); vi[e] = 0; }
  for (int e = 0; e < 16; ++e) v2[e] = (f32x2){0.01f * e, 0.02f * lane};
  const float c1 = 0.5f; const f32x2 c2 = {0.5f, 0.25f};
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (do_m) {
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u], 0, 0, 0);
      }
      if (do_v) {
        // NVI VALU instructions of the chosen kind per MFMA slot, each on its own register (re-touched only 32 instructions later), order pinned
#pragma unroll
        for (int e = 0; e < NVI; ++e) {
          const int r = (u * NVI + e) & 31;
          if (VKIND == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
          else if (VKIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[r]) : "v"(c1));
          else if (VKIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v2[r & 15]) : "v"(c2));
          else asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(vi[r]) : "v"(v[r]));
        }
      }
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  for (int e = 0; e < 32; ++e) s += v[e] + (float)vi[e];
  for (int e = 0; e < 16; ++e) s += v2[e].x + v2[e].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int VKIND, int NVI>
static double run(int threads, float* out, long long* cyc, int iters) {
  hipLaunchKernelGGL((k<MODE, VKIND, NVI>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL((k<MODE, VKIND, NVI>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  CHECK(hipDeviceSynchronize());
  long long h[8]; CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  long long m = 0; for (int w = 0; w < threads / 64; ++w) m = h[w] > m ? h[w] : m;
  return (double)m / iters / 4.0;     // s_memtime ticks (100 MHz) would be useless here: clock64 = shader clock on gfx950? report raw
}

int main() {
  float* out; long long* cyc;
  CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&cyc, 256 * 8 * 8));
  const int iters = 4000;
  const char* kn[4] = {"v_exp_f32", "v_fma_f32", "v_pk_fma_f32", "v_cvt_pk_f16_f32"};
  printf("s_memtime ticks per slot; a slot = 1 MFMA 32x32x16 f16 and / or N VALU instructions; 256 workgroups; 4 waves per workgroup = 1 wave per SIMD, 8 = 2 per SIMD\n");
  printf("mfma only, 1 wave / SIMD: %7.2f    2 waves / SIMD: %7.2f per wave-slot\n", run<0, 0, 4>(256, out, cyc, iters), run<0, 0, 4>(512, out, cyc, iters));
#define ROW(VK, N) \
  printf("%-18s x%-2d valu only %7.2f | same wave as the MFMA %7.2f | partner wave (mfma w0-3, valu w4-7) %7.2f | valu only, 2 waves / SIMD %7.2f\n", kn[VK], N, \
         run<1, VK, N>(256, out, cyc, iters), run<2, VK, N>(256, out, cyc, iters), run<3, VK, N>(512, out, cyc, iters), run<1, VK, N>(512, out, cyc, iters));
  ROW(0, 2) ROW(0, 4) ROW(0, 8) ROW(1, 4) ROW(1, 8) ROW(1, 16) ROW(2, 8) ROW(3, 8)
  return 0;
}
