// Microbenchmark (round 6): what does a launch boundary cost on the GPU?  Chains of 2000 dependent launches on one stream, HIP events around the chain:
// us per launch for empty kernels of the engine's launch shapes (grid x block x dynamic LDS) and for a kernel that reads one field of a 448-byte by-value argument
// block (what every engine kernel does first).  Run once with HIP_FORCE_DEV_KERNARG=0 and once with =1.
// build: hipcc --offload-arch=gfx950 -O3 launch_ramp.hip -o launch_ramp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct Big { int v[112]; };   // 448 bytes, GemmP-sized
__global__ void k_empty() {}
__global__ void k_lds() { extern __shared__ char s[]; if (threadIdx.x == 9999) s[0] = 1; }
__global__ void k_arg(Big b, int* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = b.v[b.v[0] & 63]; }
__global__ void k_touch(const float* in, float* out, int n) {   // a little dependent work: every block reads what the previous launch wrote
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] + 1.f;
}
// GPU-bound chains (the host runs ahead, as in the engine): every workgroup first reads a field of the argument block, then spins `us` microseconds
__global__ void k_arg_busy(Big b, int* out, int us) {
  const int f = b.v[b.v[1] & 63];
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (long long)us * 100) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = f;
}
template <class F> static void timeit(const char* name, int n, F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 50; ++i) launch(i);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < n; ++i) launch(i);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-64s %7.2f us per launch\n", name, ms * 1e3 / n);
}
int main() {
  const char* e = getenv("HIP_FORCE_DEV_KERNARG");
  printf("HIP_FORCE_DEV_KERNARG=%s\n", e ? e : "(unset)");
  int* out; hipMalloc(&out, 64); float *a, *b; hipMalloc(&a, 1 << 22); hipMalloc(&b, 1 << 22); hipMemset(a, 0, 1 << 22); hipMemset(b, 0, 1 << 22);
  hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
  hipFuncSetAttribute((const void*)k_arg_busy, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
  Big big; for (int i = 0; i < 112; ++i) big.v[i] = i;
  const int N = 2000;
  timeit("empty, 1 x 64", N, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); });
  timeit("empty, 256 x 256", N, [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0); });
  timeit("empty, 256 x 768 threads, 120 KiB LDS (producer / consumer GEMM)", N, [&](int) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(768), 120 * 1024, 0); });
  timeit("empty, 256 x 512 threads, 128 KiB LDS -> 120 (loader GEMM)", N, [&](int) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 120 * 1024, 0); });
  timeit("empty, 4096 x 256 (norm / elementwise grids)", N, [&](int) { hipLaunchKernelGGL(k_empty, dim3(4096), dim3(256), 0, 0); });
  timeit("448-byte argument block, one field read, 1 x 64", N, [&](int) { hipLaunchKernelGGL(k_arg, dim3(1), dim3(64), 0, 0, big, out); });
  timeit("448-byte argument block, one field read, 256 x 768", N, [&](int) { hipLaunchKernelGGL(k_arg, dim3(256), dim3(768), 0, 0, big, out); });
  timeit("dependent touch of 1 MiB, 1024 x 256 (ping-pong buffers)", N, [&](int i) { hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, 1 << 18); });
  for (int us : {5, 10, 20}) {
    char nm[96];
    snprintf(nm, sizeof(nm), "GPU-bound: argument read + %d us spin, 256 x 768, 120 KiB LDS", us);
    timeit(nm, N, [&](int) { hipLaunchKernelGGL(k_arg_busy, dim3(256), dim3(768), 120 * 1024, 0, big, out, us); });
    snprintf(nm, sizeof(nm), "GPU-bound: argument read + %d us spin, 64 x 256", us);
    timeit(nm, N, [&](int) { hipLaunchKernelGGL(k_arg_busy, dim3(64), dim3(256), 0, 0, big, out, us); });
  }
  return 0;
}
