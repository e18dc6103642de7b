// Weight-stationary streaming GEMM for the short-K layers of the narrow level (round 4): Out[M, N] = epilogue(A[M, 320] . W[N, 320]^T), N a
// multiple of 320 - the attention output / proj_in / proj_out projections (76800 x 320 x 320) and the fused Q | K | V projection
// (76800 x 960 x 320) of the level-0 transformers.
//
// These layers are memory-shaped (142 MB moved for 15.7 GFLOP): on the tiled kernels every workgroup runs load -> 5 K steps -> epilogue in
// sequence and the chip moves 2 - 3 TB/s (profiles/r03_per_shape_hip_events_25step.txt); GroupNorm, with nothing but loads in flight, moves
// 4.3 - 5.  Here the WEIGHTS never move: a workgroup owns a 320-column group of W for its whole life, held in registers as MFMA fragments
// (5 compute waves x 64 columns x 320 K = 160 VGPRs per lane), and the activation rows stream through a 4-slot LDS ring of 32-row tiles
// (20 KiB each, direct-to-LDS loads by a sixth wave, two to three tiles = 40 - 60 KB in flight per CU at all times).  Per 32-row tile a
// compute wave issues 20 fragment reads and 80 MFMAs (1280 cycles; one 16-row block after the other) against ~6000 cycles of HBM time for the tile's 60 KB: the loop is
// paced by memory, as it should be.  One s_barrier per tile; the epilogue (bias, residuals, 32-byte stores per lane) is the compute waves' own
// plain loads / stores - only the fetch wave has direct-to-LDS loads in flight, so its counted vmcnt never sees them.  One residual (R1), one bias.
//
// Operands as in gemm_kernel: W fragment as the MFMA "A" operand with the same row permutation (every lane ends up with 16 contiguous
// output columns), the activation fragment as "B", K walked in the same order with the same instruction => bit-identical outputs.
// N = 960: three column groups; workgroup b serves group b % 3, so the three workgroups that need an activation tile run side by side and
// two of them find it in the L2.
#include "gemm_common.h"
#include <algorithm>

// Measured and removed (round 4): (i) deeper rings - 5 ... 8 slots, one or two fetch waves - run the same or 2 - 5 % slower than 4 slots: the loop is
// not short of bytes in flight; (ii) the LayerNorm in front of the Q | K | V projection done INSIDE this kernel (two more waves normalising tile
// i + 1 in LDS while tile i is multiplied; op-level tests green): clip 975.3 / 974.4 ms against 972.3 / 971.8 for LayerNorm launch + this kernel.
template <int NSLOT, int NF>   // ring slots, fetch waves (each loads 20 / NF KiB of a tile: vmcnt counts <= 63 loads)
__global__ __launch_bounds__((5 + NF) * 64, 2) void gemm_stream320_kernel(const GemmP p, int ngroups, int wg_per_group) {
  constexpr int BM = 32, KT = 5, NCW = 5;
  constexpr int LPT = 20 / NF;                       // loads per fetch wave per tile
  static_assert((NSLOT - 2) * LPT <= 63 && NSLOT * 20 <= 160 && NSLOT >= 3, "vmcnt range / LDS");
  constexpr int TILE = KT * BM * 64;                 // halves per activation tile: [KT][32 rows][64]
  constexpr unsigned SENT = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) f16 ring[];   // [NSLOT][TILE] = 80 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = blockIdx.x % ngroups, wslot = blockIdx.x / ngroups;
  if (wslot >= wg_per_group) return;
  const int ntiles = (p.M + BM - 1) / BM;
  const int my_tiles = wslot < ntiles ? (ntiles - wslot + wg_per_group - 1) / wg_per_group : 0;

  if (wave >= NCW) {
    // ================================= fetch waves =================================
    const int fw = wave - NCW;                       // fetch wave fw loads row groups j = fw * (4 / NF) .. of every K tile
    const int pc = lane & 7, lrow = lane >> 3;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A0, 0, (int)SENT, 0x00020000);
    constexpr int JPW = 4 / NF;
    unsigned roff[JPW];                              // byte offset of (row j*8 + lrow, swizzled chunk) inside a tile's K tile 0
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
      const int r = (fw * JPW + jj) * 8 + lrow;
      roff[jj] = (unsigned)r * (unsigned)(p.C0 * 2) + (unsigned)((pc ^ swz<64>(r)) * 16);
    }
    auto issue = [&](int ti) {                       // tile index of this workgroup
      const int m0 = (wslot + ti * wg_per_group) * BM;
      f16* dst = ring + (ti % NSLOT) * TILE;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
          const int j = fw * JPW + jj;
          const bool ok = m0 + j * 8 + lrow < p.M;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(dst + (kt * BM + j * 8) * 64), 16, ok ? (int)roff[jj] : (int)SENT, (m0 * p.C0 + kt * 64) * 2, 0, 0);
        }
    };
    // NSLOT - 1 tiles issued ahead; tile it must have landed before the barrier that starts iteration it
    for (int t = 0; t < NSLOT - 1 && t < my_tiles; ++t) issue(t);
    for (int it = 0; it < my_tiles; ++it) {
      // younger tiles that may still be in flight behind tile it: issued so far = up to it + NSLOT - 2
      const int younger = max(0, min(it + NSLOT - 2, my_tiles - 1) - it);
      switch (younger) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPT <= 63 ? 3 * LPT : 63) : "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LPT <= 63 ? 4 * LPT : 63) : "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * LPT <= 63 ? 5 * LPT : 63) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * LPT <= 63 ? 6 * LPT : 63) : "memory"); break;
      }
      __builtin_amdgcn_s_barrier();                  // publishes tile it; everybody has left the slot of tile it - 1
      if (it + NSLOT - 1 < my_tiles) issue(it + NSLOT - 1);   // into the slot of tile it - 1
    }
    return;
  }
  // ================================= compute waves =================================
  const int l15 = lane & 15, g = lane >> 4;
  const int nbase = grp * 320 + wave * 64;           // this wave's 64 output columns
  // W fragments, resident: MFMA "A" operand of n-tile j, K step s = W row nbase + perm(j, l15), K [s*32 + g*8, +8);
  // perm: row i of n-tile j <-> column (j >> 1) * 32 + (i >> 2) * 8 + (j & 1) * 4 + (i & 3): lane group g ends up with the two 8-column runs
  // [g*8, +8) and [32 + g*8, +8) of the wave's 64 columns, so that the four lanes of a row write 64 contiguous bytes per store instruction
  // (16 contiguous columns per lane, as in the tiled kernels' epilogue, make every store instruction write 16-byte pieces 32 bytes apart)
  f16x8 wf[4][10];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = nbase + (j >> 1) * 32 + (l15 >> 2) * 8 + (j & 1) * 4 + (l15 & 3);
    const f16* wr = p.W + (long)n * p.ldw + g * 8;
#pragma unroll
    for (int s = 0; s < 10; ++s) wf[j][s] = n < p.N ? *(const f16x8*)(wr + s * 32) : (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
  }
  const int nc = nbase + g * 8;                      // this lane's columns: nc + [0, 8) and nc + 32 + [0, 8)
  constexpr int NC2 = 32;                            // distance of the lane's second 8-column run
  f16x8 bb[2] = {(f16x8){0, 0, 0, 0, 0, 0, 0, 0}, (f16x8){0, 0, 0, 0, 0, 0, 0, 0}};
  if (p.bias) { bb[0] = *(const f16x8*)(p.bias + nc); bb[1] = *(const f16x8*)(p.bias + nc + NC2); }
  // fragment read offsets (halves) inside a tile: row block b, K step s -> (kt = s >> 1, kk = s & 1)
  int aoff[2][2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) { const int r = b * 16 + l15; aoff[b][kk] = r * 64 + (((kk * 4 + g) ^ swz<64>(r)) * 8); }

  // residual rows: the loads of tile i + 1 are issued at the END of iteration i (into the registers the epilogue has just consumed), so they have
  // the barrier wait and the MFMAs of the next iteration to land
  f16x8 r1[2][2];
  auto load_r1 = [&](int ti) {
    const int m0 = (wslot + ti * wg_per_group) * BM;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const long m = m0 + b * 16 + l15;
      const f16* rp = p.R1 + (m < p.M ? m : 0) * p.ldr1 + nc;
      r1[b][0] = *(const f16x8*)rp; r1[b][1] = *(const f16x8*)(rp + 32);
    }
  };
  // The resident W fragments / bias must be complete HERE: otherwise the compiler's wait for them sits at the first MFMA of the loop body, as a
  // vmcnt(0) that every iteration executes - and that also waits for the previous tile's stores to be acknowledged (~2 us per tile: the loop ran at
  // 3.3 us per tile whatever the bytes; seen in the ISA).  The builtin form is the one the wait-count pass takes into account.
  __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0)
  if (p.R1 && my_tiles > 0) load_r1(0);
  for (int i = 0; i < my_tiles; ++i) {
    const int m0 = (wslot + i * wg_per_group) * BM;
    __builtin_amdgcn_s_barrier();                    // tile i is in its slot
    asm volatile("" ::: "memory");
    const f16* T = ring + (i % NSLOT) * TILE;
    // one 16-row block at a time: 16 accumulator registers live instead of 32 (with both blocks live the kernel spilled a W fragment, and the
    // reload's vmcnt(0) inside the loop made every tile wait for the previous tile's stores to be acknowledged)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      f32x4 acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 10; ++s) {
        const f16x8 af = *(const f16x8*)(T + (s >> 1) * BM * 64 + aoff[b][s & 1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j][s], af, acc[j], 0, 0, 0);
      }
      const long m = m0 + b * 16 + l15;
      float o[16];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[j * 4 + r] = p.c0 * (acc[j][r] + (float)bb[(j * 4 + r) >> 3][(j * 4 + r) & 7]);
      if (p.R1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { o[e] += p.c1 * (float)r1[b][0][e]; o[8 + e] += p.c1 * (float)r1[b][1][e]; }
      }
      f16x8 h0, h1;
#pragma unroll
      for (int e = 0; e < 8; ++e) { h0[e] = (f16)o[e]; h1[e] = (f16)o[8 + e]; }
      if (m < p.M) {
        f16* op = (f16*)p.Out + m * p.ldo + nc;
        *(f16x8*)op = h0; *(f16x8*)(op + 32) = h1;
      }
    }
    if (p.R1 && i + 1 < my_tiles) load_r1(i + 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

bool gemm_stream_supported(const GemmP& p, int batch) {
  // tools/ab_stream.py: 0.63 - 0.72 x the tiled time at M = 76800, 0.9 at 19200, 1.2 - 1.3 x at 5184.  (A GEGLU form for the feed-forward tail rows,
  // 11264 x 2560 x 320, measured 40.3 against 41.4 us in the clip and was removed.)
  if (p.conv || batch != 1 || p.K != 320 || p.N % 320 || p.N > 960 || p.M < 16384) return false;
  if (p.act != UG_ACT_NONE || (p.flags & (UG_F_GEGLU | UG_F_OUT_F32 | UG_F_R1_F32)) || p.splitk > 1 || p.up_phase || p.R2 || p.bias2) return false;
  if ((p.C0 & 7) || (p.ldw & 7) || (p.ldo & 7) || (p.R1 && (p.ldr1 & 7))) return false;
  return (long)p.M * p.C0 * 2 < (1L << 31) - 64;
}

void launch_gemm_stream(const GemmP& p, hipStream_t s) {
  UG_REQUIRE(gemm_stream_supported(p, 1), "streaming GEMM: K = 320, N a multiple of 320 (<= 960), dense, fp16 out, no activation");
  const int ngroups = p.N / 320;
  const int ntiles = cdiv(p.M, 32);
  const int wpg = std::max(1, std::min(256 / ngroups, ntiles));
  auto go = [&](auto kern, int nslot, int nwave) {
    const size_t lds = (size_t)nslot * 5 * 32 * 64 * sizeof(f16);
    UG_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(ngroups * wpg), dim3(nwave * 64), lds, s, p, ngroups, wpg);
  };
  go(gemm_stream320_kernel<4, 1>, 4, 6);
}
