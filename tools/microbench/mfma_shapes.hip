// Microbenchmark: sustained chip-wide fp16 MFMA rate by instruction shape (16x16x32 vs 32x32x16), operands in registers, NW waves per SIMD.
// The clip runs at ~2.0 GHz (not 2.4): if the shader clock under a matrix-heavy load depends on the operand-read traffic per FLOP, the
// 32x32x16 shape (half the A/B register reads per MAC) should sustain a higher rate.  Wall-clock (HIP events) over ~50 ms per case.
// build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/mshape tools/microbench/mfma_shapes.hip && /tmp/mshape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int SHAPE, int NCH>   // SHAPE 0: 16x16x32 (16 cycles), 1: 32x32x16 (32 cycles); NCH independent accumulator chains
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { a[i][e] = (f16)(0.001f * (lane + e + i)); b[i][e] = (f16)(0.002f * (lane - e - i)); }
  f32x4 c4[NCH]; f32x16 c16[NCH];
  for (int i = 0; i < NCH; ++i) { for (int e = 0; e < 4; ++e) c4[i][e] = 0.f; for (int e = 0; e < 16; ++e) c16[i][e] = 0.f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int u = 0; u < NCH; ++u) {
        if (SHAPE == 0) c4[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(u + r) & 3], b[u & 3], c4[u], 0, 0, 0);
        else c16[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u + r) & 3], b[u & 3], c16[u], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < NCH; ++i) { for (int e = 0; e < 4; ++e) s += c4[i][e]; for (int e = 0; e < 16; ++e) s += c16[i][e]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int NCH>
static void run(const char* name, int threads, float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<SHAPE, NCH>), dim3(256), dim3(threads), 0, 0, out, 2000);
  CHECK(hipDeviceSynchronize());
  double best = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<SHAPE, NCH>), dim3(256), dim3(threads), 0, 0, out, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = 2.0 * 256 * (threads / 64) * (double)iters * 4 * NCH * (SHAPE == 0 ? 16.0 * 16 * 32 : 32.0 * 32 * 16);
    const double tf = flop / (ms * 1e-3) / 1e12;
    if (tf > best) best = tf;
    printf("%-34s %4d threads/CU: %7.2f ms  %7.1f TFLOP/s  (implied clock %.0f MHz at 4096 FLOP/clk/CU)\n", name, threads, ms, tf, tf * 1e12 / (256.0 * 4096) / 1e6);
  }
}

int main() {
  float* out; CHECK(hipMalloc(&out, 256 * 512 * 4));
  for (int round = 0; round < 2; ++round) {
    run<0, 4>("16x16x32 f16, 4 chains", 256, out);
    run<1, 4>("32x32x16 f16, 4 chains", 256, out);
    run<0, 4>("16x16x32 f16, 4 chains", 512, out);
    run<1, 4>("32x32x16 f16, 4 chains", 512, out);
  }
  return 0;
}
