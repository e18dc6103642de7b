// Microbenchmark: what does ISSUING direct-to-LDS loads (buffer_load_dwordx4 ... lds) cost a wave, and does it depend on
// rewriting M0 between loads?  One workgroup of W waves per CU, every wave issues NL loads of 1 KiB per round.
//   mode 0: a fresh M0 (LDS base) for every load                  - what the GEMM kernels do
//   mode 1: one M0 per 4 loads, the LDS row selected by the instruction's immediate offset (compensated in soffset)
//   mode 2: plain global_load_dwordx4 into VGPRs (no LDS), for reference
// build: hipcc --offload-arch=gfx950 -O3 dma_issue.hip -o dma_issue (the binary is not tracked).  Result on MI355X with L2-resident
// data: a CU moves 64 KiB per ~1410 cycles with 8 waves issuing (46 B/clk of the 64 B/clk path while all 256 CUs hit L2 at once),
// M0 rewrites do not matter, plain global loads cost the same - the fetch of a 256x256x64 K-step is >= 1024 cycles of that path.
// Prints cycles per load seen by wave 0 (s_memtime around the issue loop, and around issue + vmcnt(0)).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;
typedef _Float16 f16;

template <int MODE, int NL>
__global__ __launch_bounds__(512) void k(const f16* src, unsigned long long* out, int rounds, float* sink) {
  extern __shared__ f16 smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
  const int voff = ((blockIdx.x & 7) * 8 + wave) * 65536 + lane * 16;   // 8 distinct 512 KiB windows: L2-resident after the first round
  f16* base = smem + wave * NL * 512;
  float acc = 0.f;
  unsigned long long t0, t1, t2;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < rounds; ++it) {
    const int so = (it & 3) * 8192;
    if (MODE == 0) {
#pragma unroll
      for (int l = 0; l < NL; ++l) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(base + l * 512), 16, voff, so + l * 1024, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int l = 0; l < NL; l += 4) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(base + l * 512), 16, voff, so + l * 1024, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(base + l * 512), 16, voff, so + l * 1024, 1024, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(base + l * 512), 16, voff, so + l * 1024, 2048, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(base + l * 512), 16, voff, so + l * 1024, 3072, 0);
      }
    } else {
      typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int l = 0; l < NL; ++l) { f4 v = *(const f4*)((const char*)src + voff + so + l * 1024); acc += v.x; }
    }
    if (it == rounds - 1) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t2 - t0; out[1] = t1; out[2] = t0; }
  if (acc == 12345.f) sink[0] = acc + (float)smem[lane];
}

template <int MODE, int NL> void run(const f16* d, unsigned long long* o, float* sink, int waves, const char* name) {
  const int rounds = 64;
  hipFuncSetAttribute((const void*)k<MODE, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * NL * 1024);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, NL>), dim3(256), dim3(waves * 64), 8 * NL * 1024, 0, d, o, rounds, sink);
  hipDeviceSynchronize();
  unsigned long long h[3]; hipMemcpy(h, o, 24, hipMemcpyDeviceToHost);
  printf("%-44s waves/CU %d  loads/wave/round %2d : %7.1f cycles per round, %6.1f per load (incl. landing + barrier)\n", name, waves, NL,
         (double)h[0] / rounds, (double)h[0] / rounds / NL);
}

int main() {
  f16* d; unsigned long long* o; float* sink;
  hipMalloc(&d, 1ull << 30); hipMemset(d, 0, 1ull << 30); hipMalloc(&o, 64); hipMalloc(&sink, 64);
  for (int waves : {1, 2, 4, 8}) {
    run<0, 8>(d, o, sink, waves, "LDS-DMA, M0 per load");
    run<1, 8>(d, o, sink, waves, "LDS-DMA, M0 per 4 loads (imm offsets)");
    run<2, 8>(d, o, sink, waves, "global_load_dwordx4 -> VGPR");
  }
  run<0, 16>(d, o, sink, 4, "LDS-DMA, M0 per load");
  run<1, 16>(d, o, sink, 4, "LDS-DMA, M0 per 4 loads (imm offsets)");
  return 0;
}
