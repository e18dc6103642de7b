// Microbenchmark: can the matrix pipe run at full rate while the OTHER wave of the same SIMD issues direct-to-LDS loads?
// One workgroup of 8 waves per CU.  Waves 0-3 (one per SIMD) run a chain-free stream of v_mfma_f32_16x16x32_f16 on
// registers only; waves 4-7 (their SIMD partners) either idle or issue buffer_load_dwordx4 ... lds from an L2-resident
// window, NL per round, with a vmcnt(0) per round.  Reports cycles per MFMA seen by wave 0 in both settings and the
// load rate of wave 4.  Result on MI355X: 16.45 cycles per MFMA alone, 16.29 with the partner issuing DMA (72.8 cycles per load per
// wave = one load per 18 cycles per CU), 16.28 with DMA + ds_read_b128 traffic: the matrix pipe is NOT disturbed by a partner
// wave's fetch - the serialisation seen in the GEMM kernels comes from every wave doing both jobs.   build: hipcc --offload-arch=gfx950 -O3 mfma_vs_dma.hip -o mfma_vs_dma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lptr_t;
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: partners idle, 1: partners issue LDS-DMA, 2: partners issue DMA + ds_read_b128 traffic
__global__ __launch_bounds__(512) void k(const f16* src, unsigned long long* out, int rounds, float* sink) {
  extern __shared__ f16 smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned long long t0, t1;
  if (wave < 4) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (f16)(lane * 0.001f + i); b[i] = (f16)(i * 0.5f); }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    if (s == 12345.f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; }
  } else {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    const int voff = ((blockIdx.x & 7) * 8 + wave) * 65536 + lane * 16;
    f16* base = smem + (wave - 4) * 16 * 512;
    f16x8 v = {};
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (MODE >= 1) {
      for (int it = 0; it < rounds / 4; ++it) {
#pragma unroll
        for (int l = 0; l < 16; ++l) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(base + l * 512), 16, voff, (it & 3) * 16384 + l * 1024, 0, 0);
        if (MODE == 2) {
#pragma unroll
          for (int l = 0; l < 24; ++l) { f16x8 x = *(const f16x8*)(base + ((l * 64 + lane) & 1023) * 8); v = v + x; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (v[0] == (f16)123.f) sink[1] = (float)v[1];
    if (threadIdx.x == 256 && blockIdx.x == 0) { out[1] = t1 - t0; }
  }
}

template <int MODE> void run(const f16* d, unsigned long long* o, float* sink, const char* name) {
  const int rounds = 256;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 64 * 1024, 0, d, o, rounds, sink);
  hipDeviceSynchronize();
  unsigned long long h[2]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
  printf("%-52s: %6.2f cycles per MFMA (wave 0, %d MFMAs)", name, (double)h[0] / (rounds * 64), rounds * 64);
  if (MODE) printf(";  partner: %6.1f cycles per 1 KiB load (%d loads)", (double)h[1] / (rounds / 4 * 16), rounds / 4 * 16);
  printf("\n");
}

int main() {
  f16* d; unsigned long long* o; float* sink;
  hipMalloc(&d, 1ull << 28); hipMemset(d, 0, 1ull << 28); hipMalloc(&o, 64); hipMalloc(&sink, 64);
  run<0>(d, o, sink, "MFMA stream, SIMD partner idle");
  run<1>(d, o, sink, "MFMA stream, SIMD partner issuing LDS-DMA loads");
  run<2>(d, o, sink, "MFMA stream, partner issuing LDS-DMA + ds_read_b128");
  return 0;
}
