// Microbenchmark (round 6): does the end-of-kernel release (L2 write-back of dirty lines) cost time in proportion to the bytes a kernel has just written?
// Chains of dependent launches; every kernel writes `mb` MB (each workgroup its own contiguous slice, 16-byte stores, plain or non-temporal), stamps the 100 MHz
// clock at its first instruction and after its last store has been ISSUED, and the next kernel reads one value of it.  Reported per case: kernel body time (first
// start -> last end stamp) and the gap to the next kernel's first start.  A gap that grows with the bytes written = write-back serialised at the boundary.
// build: hipcc --offload-arch=gfx950 -O3 kernel_end_flush.hip -o kernel_end_flush
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
struct Stamp { unsigned long long start, end; };
#define G 2048
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ void __launch_bounds__(256) k_write(f32x4* out, const f32x4* prev, long n_vec, Stamp* st, int idx) {
  const long long t0 = wall_clock64();
  const long per = (n_vec + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * per, hi = min(lo + per, n_vec);
  const f32x4 seed = prev ? prev[lo < n_vec ? lo : 0] : (f32x4){1.f, 2.f, 3.f, 4.f};    // dependence on the previous launch
  for (long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const f32x4 v = seed + (float)i;
    if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
  }
  if (threadIdx.x == 0) { Stamp s; s.start = t0; s.end = wall_clock64(); st[(long)idx * G + blockIdx.x] = s; }
}
int main() {
  const int N = 400;
  Stamp* st; hipMalloc(&st, sizeof(Stamp) * N * G);
  f32x4 *a, *b; const long maxb = 256L << 20; hipMalloc(&a, maxb); hipMalloc(&b, maxb);
  std::vector<Stamp> hg((size_t)N * G);
  for (int nt = 0; nt < 2; ++nt)
    for (int mb : {0, 1, 4, 12, 25, 49, 98, 196}) {
      const long n_vec = mb == 0 ? 256 : (long)mb * (1 << 20) / 16;
      const int grid = mb <= 1 ? 256 : 1024;
      hipMemset(st, 0, sizeof(Stamp) * N * G); hipDeviceSynchronize();
      for (int i = 0; i < N; ++i) {
        f32x4* o = (i & 1) ? a : b; const f32x4* pv = i ? ((i & 1) ? b : a) : nullptr;
        if (nt) hipLaunchKernelGGL(k_write<1>, dim3(grid), dim3(256), 0, 0, o, pv, n_vec, st, i);
        else hipLaunchKernelGGL(k_write<0>, dim3(grid), dim3(256), 0, 0, o, pv, n_vec, st, i);
      }
      hipDeviceSynchronize();
      hipMemcpy(hg.data(), st, sizeof(Stamp) * N * G, hipMemcpyDeviceToHost);
      std::vector<double> gap, body;
      unsigned long long ps = 0, pe = 0;
      for (int i = 0; i < N; ++i) {
        unsigned long long s = ~0ull, e = 0;
        for (int g = 0; g < grid; ++g) { const Stamp& q = hg[(size_t)i * G + g]; if (q.end) { s = std::min(s, q.start); e = std::max(e, q.end); } }
        if (i > 50) { gap.push_back(((double)s - (double)pe) / 100.0); body.push_back(((double)e - (double)s) / 100.0); }
        ps = s; pe = e;
      }
      std::sort(gap.begin(), gap.end()); std::sort(body.begin(), body.end());
      printf("%s stores, %3d MB per kernel (%4d workgroups): body %7.2f us (median), gap to the next kernel %6.2f us (median) %6.2f (p90)  -> %6.2f us per launch\n", nt ? "non-temporal" : "plain       ", mb, grid,
             body[body.size() / 2], gap[gap.size() / 2], gap[gap.size() * 9 / 10], body[body.size() / 2] + gap[gap.size() / 2]);
      (void)ps;
    }
  return 0;
}
