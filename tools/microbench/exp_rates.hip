// Microbenchmark: cycles per instruction (one wave per SIMD, independent registers) of the exponentials a flash-attention softmax can use on gfx950:
// v_exp_f32, v_exp_f16 (on one half of a register), v_exp_f16 on both halves via SDWA / op_sel, plus v_cvt_pk_f16_f32 and v_pk_mul_f16 for reference.
// Result (MI355X, round 4): v_exp_f16 costs exactly what v_exp_f32 costs (8.89 ticks against 7.87 for v_fma_f32 / v_pk_mul_f16 in this loop; the SDWA form
// for the high half 10.3) - there is no cheaper half-precision exponential for the softmax.
// build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/exr tools/microbench/exp_rates.hip && /tmp/exr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float v[32]; unsigned h[32];
  for (int e = 0; e < 32; ++e) { v[e] = -0.01f * (lane + e); h[e] = 0xB800B400u + e; }   // fp16 pairs around -0.5 / -0.25
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      if (KIND == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[e]));
      else if (KIND == 1) asm volatile("v_exp_f16 %0, %0" : "+v"(h[e]));
      else if (KIND == 2) { asm volatile("v_exp_f16 %0, %0" : "+v"(h[e])); asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(h[e])); }
      else if (KIND == 3) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(h[e]) : "v"(v[e]));
      else if (KIND == 4) asm volatile("v_pk_mul_f16 %0, %0, %0" : "+v"(h[e]));
      else if (KIND == 5) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[e]));
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int e = 0; e < 32; ++e) s += v[e] + (float)h[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int per_iter) {
  float* out; long long* cyc;
  CHECK(hipMalloc(&out, 256 * 256 * 4)); CHECK(hipMalloc(&cyc, 256 * 4 * 8));
  const int iters = 2000;
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
  CHECK(hipDeviceSynchronize());
  long long h[4]; CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  printf("%-44s %6.2f cycles per instruction (clock64 ticks; %d instructions per iteration)\n", name, (double)h[0] / iters / per_iter, per_iter);
  CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main() {
  run<5>("v_fma_f32 (reference)", 32);
  run<0>("v_exp_f32", 32);
  run<1>("v_exp_f16 (low half)", 32);
  run<2>("v_exp_f16 low + v_exp_f16_sdwa high", 64);
  run<3>("v_cvt_pk_f16_f32", 32);
  run<4>("v_pk_mul_f16", 32);
  return 0;
}
