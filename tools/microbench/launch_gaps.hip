// Microbenchmark (round 6): the idle time between consecutive dependent launches, measured ON the GPU without a profiler.
// Every kernel stamps the constant 100 MHz clock (wall_clock64) when its first workgroup starts and when each workgroup ends (atomicMax); the gap before launch
// i + 1 is start[i + 1] - end[i].  rocprofv3's kernel trace showed ~10 us gaps on every third launch of the clip (tools/gap_analysis.py) - under the profiler,
// which also stretches the clip by 30 %.  This chain imitates the engine's launch stream (448-byte by-value argument blocks, two alternating kernels, 20 - 200 us
// of work per launch, the host far ahead) and prints the gap histogram, so that what the hardware does on its own can be told from what the profiler adds.
// build: hipcc --offload-arch=gfx950 -O3 launch_gaps.hip -o launch_gaps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
struct Big { int v[104]; };
struct Stamp { unsigned long long start, end; };
#define G 1024   // stamp slots per launch: one per workgroup (plain stores - 1024 same-address atomics would cost 11 us themselves)
__device__ __forceinline__ void body(const Big& b, Stamp* st, int idx, int busy_ticks, int* out) {
  const long long t0 = wall_clock64();
  const int f = b.v[b.v[1] & 63];
  while (wall_clock64() - t0 < busy_ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) { if (f == 12345678) out[0] = f; Stamp s; s.start = t0; s.end = wall_clock64(); st[(long)idx * G + blockIdx.x] = s; }
}
__global__ void k_a(Big b, Stamp* st, int idx, int busy_ticks, int* out) { body(b, st, idx, busy_ticks, out); }
__global__ void k_b(Big b, Stamp* st, int idx, int busy_ticks, int* out) { extern __shared__ char s[]; if (threadIdx.x == 99999) s[0] = 1; body(b, st, idx, busy_ticks, out); }
int main(int argc, char** argv) {
  const int N = 6000;
  const char* e = getenv("HIP_FORCE_DEV_KERNARG");
  printf("HIP_FORCE_DEV_KERNARG=%s\n", e ? e : "(unset)");
  Stamp* st; hipMalloc(&st, sizeof(Stamp) * N * G); int* out; hipMalloc(&out, 64);
  hipFuncSetAttribute((const void*)k_b, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
  Big big; for (int i = 0; i < 104; ++i) big.v[i] = i;
  for (int variant = 0; variant < 3; ++variant) {
    // 0: one kernel, 20 us; 1: two kernels alternating (256 x 768 with 120 KiB LDS / 1024 x 256), 20 / 10 us; 2: mixed durations 5 .. 200 us
    std::vector<Stamp> hg((size_t)N * G);
    hipMemset(st, 0, sizeof(Stamp) * N * G);
    hipDeviceSynchronize();
    for (int i = 0; i < N; ++i) {
      int busy = 2000;
      if (variant == 2) { static const int d[7] = {500, 4000, 1000, 20000, 700, 6000, 2500}; busy = d[i % 7]; }
      if (variant == 0 || (i & 1)) hipLaunchKernelGGL(k_a, dim3(1024), dim3(256), 0, 0, big, st, i, variant == 1 ? 1000 : busy, out);
      else hipLaunchKernelGGL(k_b, dim3(256), dim3(768), 120 * 1024, 0, big, st, i, busy, out);
    }
    hipDeviceSynchronize();
    hipMemcpy(hg.data(), st, sizeof(Stamp) * N * G, hipMemcpyDeviceToHost);
    std::vector<Stamp> h(N);
    for (int i = 0; i < N; ++i) {
      h[i].start = ~0ull; h[i].end = 0;
      for (int g = 0; g < G; ++g) { const Stamp& q = hg[(size_t)i * G + g]; if (q.end) { h[i].start = std::min(h[i].start, q.start); h[i].end = std::max(h[i].end, q.end); } }
    }
    std::vector<double> gap;
    for (int i = 200; i + 1 < N; ++i) gap.push_back(((double)h[i + 1].start - (double)h[i].end) / 100.0);   // us
    std::vector<double> s = gap; std::sort(s.begin(), s.end());
    double sum = 0; for (double g : gap) sum += g;
    printf("variant %d: %zu gaps, mean %.2f us, p1 %.2f p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f\n", variant, gap.size(), sum / gap.size(), s[s.size() / 100], s[s.size() / 10],
           s[s.size() / 2], s[s.size() * 9 / 10], s[s.size() * 99 / 100], s.back());
    int big_n = 0; for (double g : gap) big_n += g > 6.0;
    printf("           gaps > 6 us: %d of %zu; first indices:", big_n, gap.size());
    int shown = 0; for (size_t i = 0; i < gap.size() && shown < 24; ++i) if (gap[i] > 6.0) { printf(" %zu(%.1f)", i + 200, gap[i]); ++shown; }
    printf("\n");
  }
  return 0;
}
