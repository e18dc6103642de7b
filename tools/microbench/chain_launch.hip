// Microbenchmark (round 6): device-side chaining of dependent launches.
// Kernel k + 1 is enqueued WITHOUT the AQL barrier bit (hipExtAnyOrderLaunch), so the packet processor dispatches it while kernel k still runs; its workgroups
// wait on a completion counter that the waves (or workgroups) of kernel k bump after an agent-scope release.  Questions:
//   (1) does the runtime honour the flag on gfx950 (hip_ext.h says "not supported on GFX9xx")?
//   (2) what does a dependent boundary cost then, against the 1.9 us of an in-order launch?
//   (3) is the data of kernel k visible to kernel k + 1 on every XCD (ping-pong increments, checked exactly)?
//   (4) what does the per-wave / per-workgroup release (buffer_wbl2 + atomic) cost a kernel that writes a lot?
// Every wait is BOUNDED (a missed signal ends in a wrong answer and an error flag, never in a hung GPU).
// build: hipcc --offload-arch=gfx950 -O3 chain_launch.hip -o chain_launch
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Big { int v[104]; };   // GemmP-sized argument block
struct Chain { unsigned* ctr; unsigned wait_for; unsigned mode; int* err; };   // mode bit 0: wait, bit 1: signal per wave, bit 2: signal per workgroup

__device__ __forceinline__ void chain_wait(const Chain& c) {
  if (!(c.mode & 1)) return;
  if (threadIdx.x == 0) {
    int spins = __hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? (1 << 14) : 0;   // one time-out ends all waiting
    while (__hip_atomic_load(c.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c.wait_for) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > (1 << 14)) { atomicAdd(c.err, 1); break; }
    }
    (void)__hip_atomic_load(c.ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
}
__device__ __forceinline__ void chain_signal(const Chain& c) {
  if (c.mode & 2) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(c.ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  } else if (c.mode & 4) {
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(c.ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  } else if (c.mode & 8) {     // relaxed atomic only, one address
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(c.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (c.mode & 16) {    // release fence only
    __syncthreads();
    if (threadIdx.x == 0) __atomic_thread_fence(__ATOMIC_RELEASE);   // system scope is the default of the builtin; see mode 32 for agent
  } else if (c.mode & 32) {    // agent-scope release fence only
    __syncthreads();
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  } else if (c.mode & 64) {    // agent release + atomic spread over 64 lines of 128 bytes
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(c.ctr + 32 * (1 + (blockIdx.x & 63)), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  } else if (c.mode & 128) {   // relaxed atomic spread over 64 lines
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(c.ctr + 32 * (1 + (blockIdx.x & 63)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// dependent work: out[i] = in[i] + 1 over n floats (grid-stride), after reading a field of the argument block; optional busy time
__global__ void __launch_bounds__(1024) k_step(Big b, Chain c, const float* in, float* out, int n, int busy_us) {
  const int f = b.v[b.v[1] & 63];
  chain_wait(c);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = in[i] + 1.f + (float)(f - f);
  if (busy_us > 0) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (long long)busy_us * 100) __builtin_amdgcn_s_sleep(8);
  }
  chain_signal(c);
}

static unsigned* g_ctr; static int* g_err;

struct Case { const char* name; int grid, block, lds, n, busy; };

static double run(const Case& cs, int mode, int N, float* a, float* b, bool check) {
  // mode 0: in-order launches, no counter.  1: any-order + wait + per-wave signal.  2: any-order + wait + per-workgroup signal.
  // 3: in-order launches WITH per-wave signal (cost of the release alone).  4: in-order with per-workgroup signal.
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  Big big; for (int i = 0; i < 104; ++i) big.v[i] = i;
  hipMemsetAsync(g_ctr, 0, 4, 0); hipMemsetAsync(g_err, 0, 4, 0);
  hipMemsetAsync(a, 0, (size_t)cs.n * 4, 0); hipMemsetAsync(b, 0, (size_t)cs.n * 4, 0);
  hipDeviceSynchronize();
  const unsigned per = (mode == 1 || mode == 3) ? (unsigned)cs.grid * (cs.block / 64) : (unsigned)cs.grid;
  unsigned total = 0;
  auto launch = [&](int i) {
    Chain c{g_ctr, total, 0u, g_err};
    const float* in = (i & 1) ? b : a; float* out = (i & 1) ? a : b;
    if (mode == 0) { hipLaunchKernelGGL(k_step, dim3(cs.grid), dim3(cs.block), cs.lds, 0, big, c, in, out, cs.n, cs.busy); return; }
    c.mode = (mode == 1 || mode == 3) ? 2u : 4u;
    if (mode >= 5 && mode <= 9) c.mode = 8u << (mode - 5);
    if (mode == 10) {   // any-order, no wait, no signal: do launches overlap at all?
      c.mode = 0;
      hipExtLaunchKernelGGL(k_step, dim3(cs.grid), dim3(cs.block), cs.lds, 0, nullptr, nullptr, i == 0 ? 0u : (unsigned)hipExtAnyOrderLaunch, big, c, in, out, cs.n, cs.busy);
      return;
    }
    if (mode <= 2) {
      c.mode |= 1u;
      // the first launch of a chain keeps the barrier bit: everything before it (the memsets) must be complete
      hipExtLaunchKernelGGL(k_step, dim3(cs.grid), dim3(cs.block), cs.lds, 0, nullptr, nullptr, i == 0 ? 0u : (unsigned)hipExtAnyOrderLaunch, big, c, in, out, cs.n, cs.busy);
    } else {
      hipLaunchKernelGGL(k_step, dim3(cs.grid), dim3(cs.block), cs.lds, 0, big, c, in, out, cs.n, cs.busy);
    }
    total += per;
  };
  hipEventRecord(e0);
  for (int i = 0; i < N; ++i) launch(i);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  if (check) {
    std::vector<float> h(cs.n); hipMemcpy(h.data(), (N & 1) ? b : a, (size_t)cs.n * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < cs.n; ++i) bad += (h[i] != (float)N);
    int err = 0; hipMemcpy(&err, g_err, 4, hipMemcpyDeviceToHost);
    if (bad || err) printf("    !! mode %d: %d wrong of %d (first %.0f, want %d), wait time-outs %d\n", mode, bad, cs.n, h[0], N, err);
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms * 1e3 / N;
}

int main() {
  hipMalloc(&g_ctr, 16384); hipMemset(g_ctr, 0, 16384); hipMalloc(&g_err, 64);
  const int NMAX = 64 << 20;
  float *a, *b; hipMalloc(&a, (size_t)NMAX * 4); hipMalloc(&b, (size_t)NMAX * 4);
  hipFuncSetAttribute((const void*)k_step, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
  const Case cases[] = {
    {"empty-ish: 256 x 256, 1 Ki floats", 256, 256, 0, 1024, 0},
    {"256 x 768, 120 KiB LDS, 5 us busy, 64 Ki floats", 256, 768, 120 * 1024, 65536, 5},
    {"256 x 512, 120 KiB LDS, 20 us busy, 64 Ki floats", 256, 512, 120 * 1024, 65536, 20},
    {"1024 x 256, 1 MiB floats (norm-like, 8 MB moved)", 1024, 256, 0, 1 << 20, 0},
    {"4096 x 256 (more workgroups than slots?), 4 Mi floats (32 MB moved)", 4096, 256, 0, 4 << 20, 0},
    {"16384 x 256, 16 Mi floats (128 MB moved)", 16384, 256, 0, 16 << 20, 0},

    {"500 x 512, 120 KiB LDS (2 rounds of one-per-CU workgroups), 10 us busy", 500, 512, 120 * 1024, 65536, 10},
  };
  const char* mn[] = {"in-order", "any-order, per-wave signal", "any-order, per-workgroup signal", "in-order + per-wave release", "in-order + per-workgroup release",
                      "in-order + relaxed atomic, 1 address", "in-order + release fence (system)", "in-order + release fence (agent)", "in-order + release atomic, 64 lines", "in-order + relaxed atomic, 64 lines",
                      "any-order, no wait (wrong results ok)"};
  for (const Case& cs : cases) {
    printf("%s\n", cs.name);
    const int N = cs.n >= (16 << 20) ? 200 : 1000;
    for (int mode = 0; mode < 11; ++mode) {
      if ((mode == 1 || mode == 3) && cs.grid > 1024) continue;
      run(cs, mode, 20, a, b, false);
      const double us = run(cs, mode, N, a, b, mode != 10);
      printf("  %-36s %8.2f us per launch\n", mn[mode], us);
      fflush(stdout);
    }
  }
  return 0;
}
