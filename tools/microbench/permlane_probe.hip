// What do the gfx950 lane-swap instructions return?  (round 5: quad_rows_sum in kernels/gemm_common.h)  hipcc --offload-arch=gfx950 -O2 permlane_probe.hip -o permlane_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ float quad_rows_sum(float x) {
  // Inline asm, two copies of the value in two registers: through __builtin_amdgcn_permlane16_swap hipcc (ROCm 7.2) either emits `v_permlane16_swap v1, v1`
  // (same value twice: a register swapped with itself) or, with the copy hidden behind an asm barrier, adds result[0] to itself - both give 4 x instead of the
  // sum (tools/microbench/permlane_probe.hip).  s_nop: the VALU-write -> lane-swap-read hazard the compiler would have covered for its own instruction.
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  float c = a + b, d = c;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));
  return c + d;
}
__global__ void k2(float* o) {
  float x = (float)(1 << (threadIdx.x >> 4)) * (1.f + (threadIdx.x & 15));   // row r holds 2^r * (1 + l15): the quad sum is 15 * (1 + l15)
  float y = x * 0.5f;
  x = quad_rows_sum(x); y = quad_rows_sum(y);
  o[threadIdx.x] = x; o[64 + threadIdx.x] = y;
}
__global__ void k(unsigned* o) {
  const unsigned x = threadIdx.x;
  const auto a = __builtin_amdgcn_permlane16_swap(x, x + 100, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(x, x + 100, false, false);
  o[threadIdx.x] = a[0]; o[64 + threadIdx.x] = a[1]; o[128 + threadIdx.x] = b[0]; o[192 + threadIdx.x] = b[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[4] = {"permlane16_swap(x, x+100)[0]", "permlane16_swap(x, x+100)[1]", "permlane32_swap(x, x+100)[0]", "permlane32_swap(x, x+100)[1]"};
  for (int r = 0; r < 4; ++r) { printf("%s:", nm[r]); for (int i = 0; i < 64; ++i) printf(" %u", h[r * 64 + i]); printf("\n"); }
  float* f; hipMalloc(&f, 128 * 4);
  hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, f);
  float hf[128]; hipMemcpy(hf, f, sizeof(hf), hipMemcpyDeviceToHost);
  printf("quad_rows_sum(2^row * (1 + l15)) (expect 15 * (1 + l15) in every row):"); for (int i = 0; i < 64; ++i) printf(" %g", hf[i]); printf("\n");
  printf("... of half of it:"); for (int i = 0; i < 64; ++i) printf(" %g", hf[64 + i]); printf("\n");
  return 0;
}
