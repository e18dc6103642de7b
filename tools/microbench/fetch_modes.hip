// Microbenchmark (round 6): what bounds a CU's operand stream?  One persistent workgroup per CU, W loader waves, every wave keeps D direct-to-LDS loads of
// 1 KiB in flight (counted vmcnt, no barrier) and issues `rounds` x D of them.  Address modes:
//   0 private stream   : every CU reads its own disjoint region of a large buffer (cold lines: the A operand of a GEMM, first touch)
//   1 XCD-shared stream: the 32 CUs of an XCD (blockIdx.x & 7) read the SAME region at the same time (lockstep sharers: hit-on-miss)
//   2 shared by 5      : groups of 5 CUs of one XCD share a stream (the five N tiles of one M tile)
//   3 L2-resident      : every CU re-reads its own 128 KiB window (the W operand once it is in the XCD's L2)
//   4 mix              : 5 of 8 loads from the private stream, 3 of 8 from a 2 MiB window all CUs of the XCD share (a 192 x 128 tile's K step)
//   5 shared by 5, staggered: as 2, but sharer j starts j * 64 KiB further on in the (wrapping) stream - followers find the leader's lines in L2
// span = bytes of the buffer the streams live in (128 MiB: Infinity-Cache resident on the second pass; 4 GiB: HBM).
// Prints B/clk/CU (clock from s_memtime deltas = shader cycles) and TB/s for the chip.
// build: hipcc --offload-arch=gfx950 -O3 fetch_modes.hip -o fetch_modes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lptr_t;

template <int D>
__global__ __launch_bounds__(512) void k(const char* src, unsigned long long* out, int mode, int rounds, unsigned long long span) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const int cu = blockIdx.x, xcd = cu & 7, idx = cu >> 3;          // workgroup w runs on XCD w % 8
  // stream base per CU (bytes); every load of a wave advances by nw KiB so that the waves of a CU interleave in one stream
  unsigned long long base, win = 0;
  const unsigned long long per_cu = span / 256, per_xcd = span / 8;
  if (mode == 0 || mode == 4) base = (unsigned long long)cu * per_cu;
  else if (mode == 1) base = (unsigned long long)xcd * per_xcd;
  else if (mode == 2 || mode == 5) base = (unsigned long long)xcd * per_xcd + (unsigned long long)(idx / 5) * (per_xcd / 8);
  else base = (unsigned long long)cu * (128 << 10);
  if (mode == 4) win = span + (unsigned long long)xcd * (2 << 20);
  // all stream lengths are powers of two (<= 2 GiB): the position wraps with a mask, the address is one 32-bit voffset on a per-stream descriptor
  const unsigned mask = (unsigned)(((mode == 0 || mode == 4) ? per_cu : (mode == 1 ? per_xcd : (mode == 3 ? (128ull << 10) : per_xcd / 8))) - 1);
  unsigned pos = (unsigned)wave * 1024u + (mode == 5 ? (unsigned)(idx % 5) * 65536u : 0u) + lane * 16;
  unsigned wpos = (unsigned)wave * 1024u + (unsigned)(idx & 31) * 8192u + lane * 16;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + base), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(src + win), 0, 0x7fffffff, 0x00020000);
  char* slot = smem + wave * D * 1024;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < rounds; ++it) {
#pragma unroll
    for (int l = 0; l < D; ++l) {
      const bool w = mode == 4 && ((l & 7) >= 5);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");       // at most D in flight
      if (w) { __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(slot + l * 1024), 16, (int)(wpos & ((2u << 20) - 1)), 0, 0, 0); wpos += (unsigned)nw * 1024u; }
      else { __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(slot + l * 1024), 16, (int)(pos & mask), 0, 0, 0); pos += (unsigned)nw * 1024u; }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0 && wave == 0) out[cu] = t1 - t0;
}

template <int D> static void run(const char* d, unsigned long long* o, int mode, int waves, unsigned long long span, const char* name) {
  const int rounds = 16384 / D;
  hipFuncSetAttribute((const void*)k<D>, hipFuncAttributeMaxDynamicSharedMemorySize, waves * D * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f; double cyc = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<D>), dim3(256), dim3(waves * 64), waves * D * 1024, 0, d, o, mode, rounds, span);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) {
      best = ms;
      unsigned long long h[256]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
      cyc = 0; for (int i = 0; i < 256; ++i) cyc += (double)h[i]; cyc /= 256;
    }
  }
  const double bytes_cu = (double)waves * rounds * D * 1024;
  printf("%-26s span %5llu MiB  waves %d  in flight/wave %2d : %6.1f B/clk/CU  %6.2f TB/s  (%.0f kcycles, %.3f ms, %.2f GHz)\n", name, span >> 20, waves, D,
         bytes_cu / cyc, bytes_cu * 256 / (best * 1e-3) / 1e12, cyc / 1e3, best, cyc / (best * 1e-3) / 1e9);
}

int main() {
  char* d; unsigned long long* o;
  const unsigned long long big = 4ull << 30;
  hipMalloc(&d, big + (16 << 20)); hipMemset(d, 1, big + (16 << 20)); hipMalloc(&o, 256 * 8);
  const char* names[] = {"private stream", "XCD-shared stream (32)", "shared by 5 (lockstep)", "L2-resident window", "mix 5 private : 3 L2", "shared by 5, staggered"};
  for (unsigned long long span : {128ull << 20, 4ull << 30}) {
    for (int mode : {0, 1, 2, 5, 3, 4}) {
      if (mode == 3 && span != (128ull << 20)) continue;
      run<8>(d, o, mode, 4, span, names[mode]);
      run<16>(d, o, mode, 4, span, names[mode]);
      run<32>(d, o, mode, 4, span, names[mode]);
      run<16>(d, o, mode, 8, span, names[mode]);
      run<16>(d, o, mode, 1, span, names[mode]);
    }
  }
  return 0;
}
