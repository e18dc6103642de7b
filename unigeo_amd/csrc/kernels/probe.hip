// Calibration probe (not on the product path): the chip-wide fp16 MFMA rate with operands in registers - what the matrix pipes sustain at the clock
// the power management allows under a pure matrix load.  bench.py reports it next to the 2.5 PFLOP/s spec peak ("the builder must
// also report against a measured MFMA micro-benchmark peak"); tools/microbench/mfma_shapes.hip is the stand-alone form.  Round 4, 1 x MI355X:
// 1.95 - 1.99 PFLOP/s with v_mfma_f32_16x16x32_f16 (implied 1.89 GHz), 1.69 PFLOP/s with v_mfma_f32_32x32x16_f16 (1.61 GHz).
#include "../common.h"
#include <algorithm>

__global__ __launch_bounds__(512) void k_mfma_peak(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { a[i][e] = (f16)(0.001f * (lane + e + i)); b[i][e] = (f16)(0.002f * (lane - e - i)); }
  f32x4 c[4];
  for (int i = 0; i < 4; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(u + r) & 3], b[u], c[u], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 4; ++e) s += c[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// TFLOP/s of `iters` x 16 MFMAs per wave on 256 workgroups x 8 waves (best of three ~5 ms launches after a warm-up); scratch >= 256 * 512 floats
float bench_mfma_peak(float* scratch, int iters, hipStream_t s) {
  hipEvent_t e0, e1; UG_CHECK(hipEventCreate(&e0)); UG_CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_mfma_peak, dim3(256), dim3(512), 0, s, scratch, iters / 8);
  float best = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    UG_CHECK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(k_mfma_peak, dim3(256), dim3(512), 0, s, scratch, iters);
    UG_CHECK(hipEventRecord(e1, s)); UG_CHECK(hipEventSynchronize(e1));
    float ms = 0.f; UG_CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = 2.0 * 256 * 8 * (double)iters * 16 * (16.0 * 16 * 32);
    best = std::max(best, (float)(flop / (ms * 1e-3) / 1e12));
  }
  UG_CHECK(hipEventDestroy(e0)); UG_CHECK(hipEventDestroy(e1));
  UG_CHECK(hipGetLastError());
  return best;
}
