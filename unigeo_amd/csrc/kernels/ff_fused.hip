// Fused GEGLU feed-forward for the narrow (level-0, C = 320) transformer blocks on gfx950:
//
//   Out[M, C] = c0 * ( GEGLU( X[M,C] . W1^T + b1 )[M, 4C] . W2^T + b2 ) + c1 * R1 + c2 * R2
//
// As two GEMM launches this pair is the worst-behaved part of the clip (profiles/r02_per_shape_hip_events_25step.txt): the
// projection (K = 320: five K steps under a GEGLU epilogue) runs at 640 TFLOP/s and writes a 197 MB intermediate that the
// down-projection (N = 320) then re-reads once per 128-column tile from the Infinity Cache / HBM at 490 TFLOP/s.  Here the
// [M, 4C] intermediate never leaves the CU:
//
// * workgroup = 8 wave64 (4 along M x 2 along N), BM = 128 token rows; the X tile [128 x C] is staged ONCE into LDS
//   (C/64 K tiles of [128 x 64], 16-byte chunks XOR-swizzled on the source address like the GEMM's tiles);
// * the inner dimension is walked in chunks of 64 GEGLU outputs (= 128 rows of W1, already row-interleaved [8 value | 8 gate]
//   at bind time).  Phase A: acc1[128 x 128] = X . W1c^T over C/64 K steps -> + b1 -> GEGLU in registers -> fp16 -> G tile
//   [128 x 64] in LDS.  Phase B: acc2[128 x C] += G . W2c^T, W2c = the 64 matching columns of W2, taken in pieces of 128 output
//   columns (+ one of 64) so that EVERY streamed operand packet is a [<=128 x 64] fp16 tile = 16 KiB;
// * all weight packets flow through one 3-slot LDS ring by direct-to-LDS loads issued two packets ahead, one barrier per packet
//   (same protocol as gemm_kernel: raw s_barrier, counted vmcnt, LDS-DMA stays in flight across barriers);
// * swapped MFMA operands + permuted weight rows give every lane 16 (8 in the last piece) contiguous output columns: bias,
//   GEGLU, residuals and stores are 16-byte vectors.
// * optional pre-LayerNorm (FFusedP::ln_g): the block's LayerNorm (+ the broadcast cross-attention / frame-embedding row that is
//   added to the residual stream just before it) is applied to the X tile in LDS, so the normalised [M, C] tensor and the
//   updated residual stream are never written to / re-read from HBM (one 3-tensor pass per feed-forward saved).
// LDS: X 16 KiB * C/64 + ring 48 KiB + G 16 KiB = 144 KiB at C = 320; registers: acc2 80 + acc1 32 per lane.
#include "../common.h"
#include <algorithm>
#include <type_traits>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// same packed-fp32 GEGLU as the GEMM epilogue (kernels/gemm.hip): h * g * Phi(g), erfc by Abramowitz-Stegun 7.1.26
__device__ __forceinline__ f32x2 ff_geglu2(f32x2 h, f32x2 g) {
  const f32x2 ag = {__builtin_fabsf(g.x), __builtin_fabsf(g.y)};
  const f32x2 u = ag * 0.23164189f + 1.0f;
  const f32x2 t = {__builtin_amdgcn_rcpf(u.x), __builtin_amdgcn_rcpf(u.y)};
  f32x2 y = t * 0.5307027145f - 0.7265760135f;
  y = y * t + 0.7107068705f;
  y = y * t - 0.142248368f;
  y = y * t + 0.127414796f;
  y = y * t;
  const f32x2 w = (g * g) * -0.72134752044f;
  const f32x2 e = {__builtin_amdgcn_exp2f(w.x), __builtin_amdgcn_exp2f(w.y)};
  const f32x2 q = 0.5f - y * e;
  const f32x2 phi = {0.5f + __builtin_copysignf(q.x, g.x), 0.5f + __builtin_copysignf(q.y, g.y)};
  return h * g * phi;
}

__device__ __forceinline__ int ffswz(int row) { return (row >> 1) & 7; }   // 128-byte rows, 16-byte chunks

// ABL (tools/bench_ff.py ablate): timing-only variants of the kernel with one ingredient removed - WRONG RESULTS, never launched by the engine.
// 1 = no GEGLU arithmetic, 2 = no weight-packet loads after the first two, 4 = no MFMAs, 8 = no fragment reads, 16 = no per-packet barrier
// XT (round 3): the weight-packet ring runs CONTINUOUSLY across the tiles a workgroup walks (the packets do not depend on the tile), and the
// next tile's X rows are fetched into Xs as soon as the last chunk's phase A is over - behind the two phase-B packets still to come and the
// whole epilogue.  The ablation (tools/bench_ff.py ablate, profiles/r03_ff_ablation.txt) showed 29 % of this kernel in the per-tile skeleton:
// every CU of the chip sits in its prologue (80 KB of X), then in its epilogue (80 KB of residual in, 80 KB out) at the same moment.
template <int KT, int ABL = 0, bool XT = false>   // KT = C / 64 K tiles of the X operand (C = 64 * KT <= 320)
__global__ __launch_bounds__(512, 2) void ff_fused_kernel(const FFusedP p) {
  constexpr int C = KT * 64;
  constexpr int NP = (C + 127) / 128;          // phase-B pieces: NPF full ones of 128 output columns + (C % 128 == 64) one of 64
  constexpr int NPF = C / 128;
  constexpr bool TAIL = (C % 128) != 0;
  constexpr int STEPS = KT + NP;               // packets per chunk
  constexpr int TILE = 128 * 64;               // halves per [128 x 64] tile
  extern __shared__ __attribute__((aligned(16))) f16 smem[];
  f16* Xs = smem;                              // [KT][128][64]
  f16* ring = smem + KT * TILE;                // [3][128][64]
  f16* Gs = ring + 3 * TILE;                   // [128][64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;
  const int sw = ffswz(l15);
  const int pc = lane & 7, lrow = lane >> 3;
  const int I = 4 * C;                         // inner width (GEGLU outputs); W1 has 2*I rows
  const int nchunk = I / 64;
  const int ntiles = (p.M + 127) / 128;

  // ---- buffer-addressed direct-to-LDS loads: the per-lane byte offset of every load is fixed for the whole kernel (row permutation,
  // swizzled 16-byte chunk), the packet position is a scalar offset - no address arithmetic on the issue path
  constexpr unsigned SENT = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rW1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W1, 0, (int)SENT, 0x00020000);
  const __amdgpu_buffer_rsrc_t rW2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W2, 0, (int)SENT, 0x00020000);
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, (int)((long)p.M * C * 2), 0x00020000);   // rows >= M read zeros
  unsigned vo1[2], vo2[2], vot, vox;
#pragma unroll
  for (int u = 0; u < 2; ++u) {   // LDS row lr holds weight row perm(lr): wave tile 64, 16 contiguous columns per lane
    const int lr = (wave * 2 + u) * 8 + lrow;
    const int part = lr >> 6, rem = lr & 63, jj = rem >> 4, i = rem & 15;
    const int n = part * 64 + (i >> 2) * 16 + jj * 4 + (i & 3);
    vo1[u] = (unsigned)((n * C + (pc ^ ffswz(lr)) * 8) * 2);
    vo2[u] = (unsigned)((n * I + (pc ^ ffswz(lr)) * 8) * 2);
  }
  {   // tail piece: 64 output columns, wave tile 32 (8 columns per lane)
    const int lr = wave * 8 + lrow;
    const int part = lr >> 5, rem = lr & 31, jj = rem >> 4, i = rem & 15;
    const int n = NPF * 128 + part * 32 + (i >> 2) * 8 + jj * 4 + (i & 3);
    vot = (unsigned)((n * I + (pc ^ ffswz(lr)) * 8) * 2);
  }
  vox = (unsigned)((lrow * C + 0) * 2);   // X rows: + (r*8) rows and the swizzled chunk are added per load (the swizzle depends on r)
  auto issue_packet = [&](int j, int s, int slot) {
    f16* dst = ring + slot * TILE;
    if (s < KT) {   // W1 rows [j*128, +128), K tile s
      const int so = (j * 128 * C + s * 64) * 2;
#pragma unroll
      for (int u = 0; u < 2; ++u)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW1, (lptr_t)(dst + (wave * 2 + u) * 8 * 64), 16, (int)vo1[u], so, 0, 0);
    } else if (s - KT < NPF) {   // W2 rows [pp*128, +128) (output columns), K columns [j*64, +64)
      const int so = ((s - KT) * 128 * I + j * 64) * 2;
#pragma unroll
      for (int u = 0; u < 2; ++u)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW2, (lptr_t)(dst + (wave * 2 + u) * 8 * 64), 16, (int)vo2[u], so, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW2, (lptr_t)(dst + wave * 8 * 64), 16, (int)vot, j * 64 * 2, 0, 0);
    }
  };
  // loads per wave of packet s (compile-time when s is)
  auto nload = [](int s) { return (TAIL && s == STEPS - 1) ? 1 : 2; };

  static_assert(!XT || (KT >= 2 && NP >= 2), "cross-tile prefetch needs two phase-B packets after the last phase A");
  constexpr int NX = KT * 2;                    // X-tile loads per wave
  // ---- X tile: KT x 16 wave-instructions of 1 KiB, spread over the 8 waves
  auto issue_x = [&](int m0_) {
    for (int t = wave; t < KT * 16; t += 8) {
      const int kt = t >> 4, r = t & 15;
      const int lr = r * 8 + lrow;
      const unsigned vo = vox + (unsigned)((r * 8 * C + (pc ^ ffswz(lr)) * 8) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lptr_t)(Xs + kt * TILE + r * 8 * 64), 16, (int)vo, (m0_ * C + kt * 64) * 2, 0, 0);
    }
  };
  int slot = 0;                       // ring slot of the packet being consumed (XT: runs on across tiles)
  if (XT && (int)blockIdx.x < ntiles) { issue_x(blockIdx.x * 128); issue_packet(0, 0, 0); issue_packet(0, 1, 1); }

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = tile * 128;
    const bool more = XT && tile + (int)gridDim.x < ntiles;      // this workgroup has another tile after this one
    if (!XT) {
      issue_x(m0);
      issue_packet(0, 0, 0);
      issue_packet(0, 1, 1);
      slot = 0;
    }

    if (p.ln_g) {
      // ---- pre-norm on the LDS tile: 4 threads per token row (2 chunks of 8 channels in each of the KT K tiles), exact two-pass
      // statistics in fp32 - the values launch_layernorm would have written to HBM and this kernel would have read back.
      // gamma / beta / the broadcast row are fetched BEFORE the wait for the X tile, so their latency hides behind the tile's.
      const int row = tid >> 2, q4 = tid & 3;
      const long mr = (long)m0 + row < p.M ? (long)m0 + row : (long)p.M - 1;
      const f16* av = p.addvec ? p.addvec + (long)((int)mr / p.rows_per_vec) * C : nullptr;
      f16x8 ga[KT][2], be[KT][2], ad[KT][2];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          ga[kt][u] = *(const f16x8*)(p.ln_g + kt * 64 + q4 * 16 + u * 8);
          be[kt][u] = *(const f16x8*)(p.ln_b + kt * 64 + q4 * 16 + u * 8);
          if (av) ad[kt][u] = *(const f16x8*)(av + kt * 64 + q4 * 16 + u * 8);
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      f16* xr = Xs + row * 64;
      const int cs[2] = {((q4 * 2) ^ ffswz(row)) * 8, ((q4 * 2 + 1) ^ ffswz(row)) * 8};   // this thread's two chunks of every K tile
      f16x8 hx[KT][2];
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          hx[kt][u] = *(const f16x8*)(xr + kt * TILE + cs[u]);
          if (av) {
#pragma unroll
            for (int e = 0; e < 8; ++e) hx[kt][u][e] = (f16)((float)hx[kt][u][e] + (float)ad[kt][u][e]);
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) sum += (float)hx[kt][u][e];
        }
      sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2);
      const float mean = sum / C;
      float var = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = (float)hx[kt][u][e] - mean; var += d * d; }
      var += __shfl_xor(var, 1); var += __shfl_xor(var, 2);
      const float rstd = rsqrtf(var / C + p.ln_eps);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          f16x8 y;
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] = (f16)(((float)hx[kt][u][e] - mean) * rstd * (float)ga[kt][u][e] + (float)be[kt][u][e]);
          *(f16x8*)(xr + kt * TILE + cs[u]) = y;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the first packet's barrier below publishes the normalised tile
    }

    f32x4 acc1[2][4], acc2[NP][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) {
        acc1[i][jn] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) acc2[pp][i][jn] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }

    bool first = true;                  // the X tile's loads are still in flight before the very first packet
    for (int j = 0; j < nchunk; ++j) {
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        // packet (j, s) must have landed; the next packet (issued one step ago) may stay in flight
        const bool last = (j == nchunk - 1) && (s == STEPS - 1);
        if (XT && more && j == nchunk - 1 && s > KT && s <= KT + 2) {
          // the next tile's X loads (NX per wave, issued at s == KT behind packet KT + 2) sit between this tile's last packets in the queue:
          // packet s must have landed, the X loads and the packet behind them may stay in flight
          if (s == KT + 1) { if (nload((KT + 2) % STEPS) == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NX + 1) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NX + 2) : "memory"); }
          else { if (KT + 3 < STEPS && nload(KT + 3) == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NX + 1) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NX + 2) : "memory"); }
        }
        else if (first || (last && !more)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (nload((s + 1) % STEPS) == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        if (first) {   // steady state below assumes exactly one younger packet in flight: re-establish it
          first = false;
        }
        if (s == KT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's G-tile writes (end of phase A) are done before the barrier publishes them
        if (!(ABL & 16)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (!(ABL & 2)) {   // fetch the packet two ahead into the slot everybody finished reading in the previous step
          int s2 = s + 2, j2 = j;
          if (s2 >= STEPS) { s2 -= STEPS; ++j2; }
          int slot2 = slot + 2; if (slot2 >= 3) slot2 -= 3;
          if (j2 < nchunk) issue_packet(j2, s2, slot2);
          else if (more) issue_packet(0, s2, slot2);            // XT: the stream runs on into the next tile's first packets
        }
        if (XT && more && j == nchunk - 1 && s == KT) issue_x(m0 + (int)gridDim.x * 128);   // every wave is past its last read of Xs (this packet's barrier)
        const f16* Wt = ring + slot * TILE;
        if (++slot == 3) slot = 0;
        if (s < KT) {
          // ---- phase A, K tile s: acc1 += X[:, s] . W1c[:, s]^T   (wave: 32 rows x 64 W1 rows)
          const f16* Ab = Xs + s * TILE + (wm * 32 + l15) * 64;
          const f16* Bb = Wt + (wn * 64 + l15) * 64;
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const int ch = ((kk * 4 + g) ^ sw) * 8;
            f16x8 bf[4], af[2];
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) bf[jn] = (ABL & 8) ? (f16x8){(f16)(float)(lane + s + jn), 0, 0, 0, 0, 0, 0, 0} : *(const f16x8*)(Bb + jn * 16 * 64 + ch);
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = (ABL & 8) ? (f16x8){(f16)(float)(lane + i), 0, 0, 0, 0, 0, 0, 0} : *(const f16x8*)(Ab + i * 16 * 64 + ch);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int jn = 0; jn < 4; ++jn) {
                if (ABL & 4) acc1[i][jn][0] += (float)bf[jn][0] * (float)af[i][0];
                else acc1[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[jn], af[i], acc1[i][jn], 0, 0, 0);
              }
          }
          if (s == KT - 1) {
            // ---- GEGLU: lane holds W1 rows j*128 + wn*64 + g*16 + [0,16) = [8 value | 8 gate] of inner columns j*64 + wn*32 + g*8 + [0,8)
            const f16x8 b0 = *(const f16x8*)(p.b1 + j * 128 + wn * 64 + g * 16), b1v = *(const f16x8*)(p.b1 + j * 128 + wn * 64 + g * 16 + 8);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              float v[16];
#pragma unroll
              for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[jn * 4 + r] = acc1[i][jn][r] + (float)(jn < 2 ? b0[jn * 4 + r] : b1v[(jn - 2) * 4 + r]);
              f16x8 o;
#pragma unroll
              for (int e = 0; e < 8; e += 2) {
                const f32x2 r2 = (ABL & 1) ? (f32x2){v[e] + v[8 + e], v[e + 1] + v[9 + e]} : ff_geglu2((f32x2){v[e], v[e + 1]}, (f32x2){v[8 + e], v[9 + e]});
                o[e] = (f16)r2.x; o[e + 1] = (f16)r2.y;
              }
              const int row = wm * 32 + i * 16 + l15;
              *(f16x8*)(Gs + row * 64 + (((wn * 4 + g) ^ ffswz(row)) * 8)) = o;
#pragma unroll
              for (int jn = 0; jn < 4; ++jn) acc1[i][jn] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
          }
        } else {
          // ---- phase B, piece pp: acc2[pp] += G . W2c[piece]^T
          const int pp = s - KT;
          const f16* Ab = Gs + (wm * 32 + l15) * 64;
          const bool tailp = TAIL && pp == NP - 1;
          const f16* Bb = Wt + ((tailp ? wn * 32 : wn * 64) + l15) * 64;
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const int ch = ((kk * 4 + g) ^ sw) * 8;
            f16x8 bf[4], af[2];
#pragma unroll
            for (int jn = 0; jn < 4; ++jn)
              if (!tailp || jn < 2) bf[jn] = (ABL & 8) ? (f16x8){(f16)(float)(lane + s + jn), 0, 0, 0, 0, 0, 0, 0} : *(const f16x8*)(Bb + jn * 16 * 64 + ch);
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = (ABL & 8) ? (f16x8){(f16)(float)(lane + i), 0, 0, 0, 0, 0, 0, 0} : *(const f16x8*)(Ab + i * 16 * 64 + ch);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int jn = 0; jn < 4; ++jn)
                if (!tailp || jn < 2) {
                  if (ABL & 4) acc2[pp][i][jn][0] += (float)bf[jn][0] * (float)af[i][0];
                  else acc2[pp][i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[jn], af[i], acc2[pp][i][jn], 0, 0, 0);
                }
          }
        }
      }
    }

    // ---- tile epilogue: out = c0 * (acc2 + b2) + c1 * R1 + c2 * R2; lane: rows wm*32 + i*16 + l15, 16 (tail: 8) contiguous columns
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      const bool tailp = TAIL && pp == NP - 1;
      const int wid = tailp ? 8 : 16;
      const int n0 = pp * 128 + (tailp ? wn * 32 + g * 8 : wn * 64 + g * 16);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 32 + i * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int e0 = 0; e0 < 16; e0 += 8) {
          if (e0 >= wid) continue;
          const f16x8 b = p.b2 ? *(const f16x8*)(p.b2 + n0 + e0) : (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
          float o[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) o[q] = p.c0 * (acc2[pp][i][(e0 + q) >> 2][(e0 + q) & 3] + (float)b[q]);
          if (p.R1) {
            f16x8 r = *(const f16x8*)(p.R1 + (long)m * C + n0 + e0);
            if (p.ln_g && p.addvec) {   // the residual stream is x' = fp16(X + addvec) (see FFusedP)
              const f16x8 a = *(const f16x8*)(p.addvec + ((long)m / p.rows_per_vec) * C + n0 + e0);
#pragma unroll
              for (int q = 0; q < 8; ++q) r[q] = (f16)((float)r[q] + (float)a[q]);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] += p.c1 * (float)r[q];
          }
          if (p.R2) {
            const f16x8 r = *(const f16x8*)(p.R2 + (long)m * C + n0 + e0);
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] += p.c2 * (float)r[q];
          }
          f16x8 h;
#pragma unroll
          for (int q = 0; q < 8; ++q) h[q] = (f16)o[q];
          *(f16x8*)(p.Out + (long)m * C + n0 + e0) = h;
        }
      }
    }
    // the next tile's X loads overwrite Xs / the ring: everybody must be past this tile's LDS reads, and the epilogue's
    // loads / stores must not be counted against the next tile's packets (XT: the first packet of the next tile waits for everything)
    if (!XT) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
}

// FFusedP::variant (per launch; the engine passes its context's - ug_tune_ff): 0 = the kernel above with the cross-tile prefetch (default), 1 = without
// it (the round-2 kernel, A/B); UG_EXPERIMENTS builds: 100 + mask = the timing-only ablations of tools/bench_ff.py (WRONG results)
template <int KT>
static void launch_ff_t(const FFusedP& p, hipStream_t s) {
  const size_t lds = (size_t)(KT + 3 + 1) * 128 * 64 * sizeof(f16);
  static bool attr[32] = {};
  bool& at = attr[ug_dev_slot()];
  if (!at) { UG_CHECK(hipFuncSetAttribute((const void*)ff_fused_kernel<KT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); at = true; }
  const int ntiles = (p.M + 127) / 128;
  const int variant = p.variant;
#ifdef UG_EXPERIMENTS
  if (variant >= 100) {
    if constexpr (KT == 5) {
      auto go = [&](auto kern) {
        UG_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(std::min(ntiles, 256)), dim3(512), lds, s, p);
      };
      switch (variant - 100) {
        case 1: go(ff_fused_kernel<5, 1>); break;  case 2: go(ff_fused_kernel<5, 2>); break;   case 4: go(ff_fused_kernel<5, 4>); break;
        case 8: go(ff_fused_kernel<5, 8>); break;  case 16: go(ff_fused_kernel<5, 16>); break; case 12: go(ff_fused_kernel<5, 12>); break;
        case 3: go(ff_fused_kernel<5, 3>); break;  case 14: go(ff_fused_kernel<5, 14>); break; case 30: go(ff_fused_kernel<5, 30>); break;
        case 31: go(ff_fused_kernel<5, 31>); break; case 18: go(ff_fused_kernel<5, 18>); break; case 10: go(ff_fused_kernel<5, 10>); break;
        default: go(ff_fused_kernel<5, 0>); break;
      }
      return;
    }
  }
#else
  UG_REQUIRE(variant < 100, "ff_fused: the ablation variants (100 + mask) exist only in the UG_EXPERIMENTS build (make experiments -> build/exp/libunigeo_exp.so, UG_LIB_PATH)");
#endif
  if (variant == 0) {
    if constexpr (KT >= 2 && (KT * 64 + 127) / 128 >= 2) {
      static bool attrx[32] = {};
      bool& atx = attrx[ug_dev_slot()];
      if (!atx) { UG_CHECK(hipFuncSetAttribute((const void*)ff_fused_kernel<KT, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); atx = true; }
      const int per_cu_x = std::max(1, std::min(2, (int)((160 * 1024) / lds)));
      hipLaunchKernelGGL((ff_fused_kernel<KT, 0, true>), dim3(std::min(ntiles, per_cu_x * 256)), dim3(512), lds, s, p);
      return;
    }
  }
  const int per_cu = std::max(1, std::min(2, (int)((160 * 1024) / lds)));
  const int grid = std::min(ntiles, per_cu * 256);
  hipLaunchKernelGGL(ff_fused_kernel<KT>, dim3(grid), dim3(512), lds, s, p);
}

bool ff_fused_supported(int C) { return C == 64 || C == 128 || C == 192 || C == 256 || C == 320; }

void launch_ff_fused(const FFusedP& p, hipStream_t s) {
  UG_REQUIRE(ff_fused_supported(p.C) && p.M > 0 && p.zero, "ff_fused: C must be a multiple of 64 up to 320");
  if (p.ln_g) UG_REQUIRE(p.ln_b && (!p.addvec || p.rows_per_vec >= 1) && (!p.addvec || !p.R1 || p.R1 == p.X), "ff_fused pre-norm: beta missing / residual is not the normalised stream");
  switch (p.C / 64) {
    case 1: launch_ff_t<1>(p, s); break;
    case 2: launch_ff_t<2>(p, s); break;
    case 3: launch_ff_t<3>(p, s); break;
    case 4: launch_ff_t<4>(p, s); break;
    default: launch_ff_t<5>(p, s); break;
  }
  UG_CHECK(hipGetLastError());
}
