// Attention kernels for gfx950 (head_dim 64, fp16 in, fp32 accumulate / softmax).
//
// flash_attn64 (spatial self-attention, S up to thousands):
//   * workgroup = 4 wave64, 128 query rows of one (frame, head); each wave owns 32 query rows.
//   * K/V tiles of 64 keys are staged global -> LDS by direct-to-LDS 16-byte loads, double
//     buffered; the swizzle lives on the source address (LDS image is lane-linear).
//   * scores are computed TRANSPOSED, S^T = K.Q^T with v_mfma_f32_32x32x16_f16, so every lane
//     owns one query column: the online-softmax row reductions are in-register plus a single
//     lane^32 exchange, and the exponentiated P^T registers are already the B operand of the
//     second MFMA, O^T = V^T.P^T.
//   * V^T fragments come out of the row-major V tile with the LDS transpose read
//     ds_read_b64_tr_b16 (no transposed copy of V is ever materialised).
// temporal_attn64: the same math for the tiny per-pixel sequences over frames (T <= 32); one
//   wave per (pixel, head), K and Q straight from global into MFMA fragments, V through a
//   private 4 KiB LDS slab for the transpose read.
#include "../common.h"

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __fp16 h4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

// K tile swizzle (read with ds_read_b128 by 32 different rows, same logical chunk)
__device__ __forceinline__ int kswz(int row) { return (row >> 1) & 7; }
// V tile swizzle (read with the transpose read: 4 rows x 4 chunks per 32 lanes)
__device__ __forceinline__ int vswz(int row) { return ((row >> 1) & 1) << 2; }

__device__ __forceinline__ f16x4 lds_tr16(const f16* p) {
  h4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4_t*)p);
  f16x4 o;
  o[0] = (f16)r[0]; o[1] = (f16)r[1]; o[2] = (f16)r[2]; o[3] = (f16)r[3];
  return o;
}

// One 32-key step of O^T += V^T P^T.  vt: LDS V tile [keys][64] (swizzled 16-B chunks), kb = first
// key row of the 32-key block inside the tile; s[16]: this lane's exponentiated scores.
__device__ __forceinline__ void pv_block(const f16* vt, int kb, int lane, const float (&pr)[16],
                                         f32x16 (&o)[2]) {
  const int L = lane & 15, hh = lane >> 5, db = ((lane >> 4) & 1) * 16;
#pragma unroll
  for (int a = 0; a < 2; ++a) {      // two K=16 MFMAs cover the 32 keys
    f16x8 pb;
#pragma unroll
    for (int j = 0; j < 8; ++j) pb[j] = (f16)pr[8 * a + j];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {  // two 32-wide d tiles
      f16x8 va;
#pragma unroll
      for (int h = 0; h < 2; ++h) {   // keys (j<4) and +8 (j>=4)
        const int row = kb + 16 * a + 8 * h + 4 * hh + (L >> 2);
        const int col = dt * 32 + db + (L & 3) * 4;
        const int chunk = (col >> 3) ^ vswz(row);
        const f16x4 t = lds_tr16(vt + row * 64 + chunk * 8 + (col & 7));
        va[4 * h + 0] = t[0]; va[4 * h + 1] = t[1]; va[4 * h + 2] = t[2]; va[4 * h + 3] = t[3];
      }
      o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, pb, o[dt], 0, 0, 0);
    }
  }
}

#define FA_KV 64
// KV ring depth: template parameter FA_NST of flash_attn64_kernel (3; the occupancy-4 A/B variant uses 2)
#define FA_QB 1    // 32-query blocks per wave (2 was measured: fewer L2->LDS bytes but occupancy 3 -> 2 waves/SIMD, net -5 %)

// One KV tile of 64 keys for a wave's FA_QB x 32 query rows.  MASK is only instantiated for the ragged last tile,
// so the full tiles carry no compare/select work.  Scores stay unscaled in registers; the softmax scale (times
// log2 e) is folded into the single FMA that feeds v_exp_f32.
template <bool MASK>
__device__ __forceinline__ void fa_tile(const f16* kt, const f16* vt, const f16x8 (&qf)[FA_QB][4], int lane, int kv0, int S,
                                        float sc, float (&m_run)[FA_QB], float (&l_run)[FA_QB], f32x16 (&o)[FA_QB][2]) {
  const int qi = lane & 31, hh = lane >> 5;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int L = lane & 15, db = ((lane >> 4) & 1) * 16;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {           // online-softmax step per 32-key block keeps the live register set small
    f16x8 kf[4];
    const int row = kb * 32 + qi;            // lane's key row inside the tile
#pragma unroll
    for (int c = 0; c < 4; ++c) kf[c] = *(const f16x8*)(kt + row * 64 + (((c * 2 + hh) ^ kswz(row)) * 8));
    f16x8 pb[FA_QB][2];                       // exponentiated scores as B operands: [q block][16-key half]
#pragma unroll
    for (int qb = 0; qb < FA_QB; ++qb) {
      f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[qb][0], zero16, 0, 0, 0);
#pragma unroll
      for (int c = 1; c < 4; ++c) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[c], qf[qb][c], s, 0, 0, 0);
      float mx = -1e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (MASK) {
          const int key = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (key >= S) s[r] = -1e30f;
        }
        mx = fmaxf(mx, s[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run[qb], mx * sc);      // running max in the scaled (log2) domain
      if (!__all(m_new == m_run[qb])) {                    // rescale only when some row's max moved (alpha == 1 otherwise)
        const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
        l_run[qb] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[qb][dt][r] *= alpha;
        m_run[qb] = m_new;
      }
      // packed fp32 (v_pk_fma_f32 / v_pk_add_f32): two scores per VALU instruction - the loop is VALU-bound (16 exp2 per 32 keys)
      const f32x2 sc2 = {sc, sc}, nm2 = {-m_run[qb], -m_run[qb]};
      f32x2 ps2 = {0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 sv = {s[r], s[r + 1]};
#ifdef FA_SCALAR
        const f32x2 e2 = {__builtin_amdgcn_exp2f(fmaf(sv.x, sc, nm2.x)), __builtin_amdgcn_exp2f(fmaf(sv.y, sc, nm2.y))};
        ps2.x += e2.x; ps2.y += e2.y;
#else
        const f32x2 tt = sv * sc2 + nm2;
        const f32x2 e2 = {__builtin_amdgcn_exp2f(tt.x), __builtin_amdgcn_exp2f(tt.y)};
        ps2 += e2;
#endif
        pb[qb][r >> 3][r & 7] = (f16)e2.x;
        pb[qb][r >> 3][(r & 7) + 1] = (f16)e2.y;
      }
      l_run[qb] += ps2.x + ps2.y;
    }
    // O^T += V^T P^T : every V^T fragment (two transpose reads) feeds both query blocks
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        f16x8 va;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int vrow = kb * 32 + 16 * a + 8 * h + 4 * hh + (L >> 2);
          const int col = dt * 32 + db + (L & 3) * 4;
          const int chunk = (col >> 3) ^ vswz(vrow);
          const f16x4 t = lds_tr16(vt + vrow * 64 + chunk * 8 + (col & 7));
          va[4 * h + 0] = t[0]; va[4 * h + 1] = t[1]; va[4 * h + 2] = t[2]; va[4 * h + 3] = t[3];
        }
#pragma unroll
        for (int qb = 0; qb < FA_QB; ++qb)
          o[qb][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, pb[qb][a], o[qb][dt], 0, 0, 0);
      }
  }
}

// Variant of fa_tile: the scores of BOTH 32-key blocks are computed first (two interleaved accumulate chains, so a dependent
// MFMA never issues back to back), then ONE online-softmax update over the 64 keys, then the PV products of both blocks.
// ABL (timing-only ablations, tools/bench_flash.py ablate; results are wrong): 1 = no v_exp_f32, 2 = no QK^T MFMAs, 4 = no PV MFMAs,
// 8 = no softmax arithmetic at all (scores converted straight to P)
// Packed fp32 (v_pk_fma / v_pk_add / v_pk_mul_f32) halves the VALU instruction count but does NOT run in the shadow of an MFMA, plain fp32 does
// (tools/microbench/mfma_vs_valu.hip: 8 v_pk_fma + 1 MFMA 32x32x16 in one wave 85 cycles = 42 + 32 + issue; 16 v_fma_f32 + 1 MFMA 85 vs 79
// alone); a plain-fp32 build of this loop measured the same within 1.5 % (profiles/r03_flash_variants.txt) - the packed form stays.
// LAZY (round 3): (i) the running reference m_run is only moved when some row's scores outgrow it by more than 2^8 (P stays <= 256: exact in
// fp32 accumulators, same relative precision in fp16; the final 1 / l normalisation uses the same reference, so the result is the same
// softmax) - the 32-multiply rescale of O no longer runs on most tiles; (ii) the row sum l is taken from the fp16 P the PV product uses,
// two scores per v_dot2_f32_f16, instead of one fp32 add per score.  VALU port: ~980 -> ~800 cycles per 64-key tile (profiles/r03_flash_pmc.txt)
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
template <bool MASK, int ABL = 0, bool LAZY = false>
__device__ __forceinline__ void fa_tile_wide(const f16* kt, const f16* vt, const f16x8 (&qf)[FA_QB][4], int lane, int kv0, int S,
                                             float sc, float (&m_run)[FA_QB], float (&l_run)[FA_QB], f32x16 (&o)[FA_QB][2]) {
  const int qi = lane & 31, hh = lane >> 5;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int L = lane & 15, db = ((lane >> 4) & 1) * 16;
  f32x16 s[2];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    f16x8 kf[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) { const int row = kb * 32 + qi; kf[kb] = *(const f16x8*)(kt + row * 64 + (((c * 2 + hh) ^ kswz(row)) * 8)); }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (ABL & 2) { asm volatile("" :: "v"(kf[kb])); if (c == 0) { s[kb] = zero16; s[kb][0] = (float)lane; } }
      else s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb], qf[0][c], c == 0 ? zero16 : s[kb], 0, 0, 0);
    }
  }
  if (ABL & 8) {
    f16x8 pb8[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) pb8[kb][r >> 3][r & 7] = (f16)s[kb][r];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          f16x8 va;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int vrow = kb * 32 + 16 * a + 8 * h + 4 * hh + (L >> 2);
            const int col = dt * 32 + db + (L & 3) * 4;
            const int chunk = (col >> 3) ^ vswz(vrow);
            const f16x4 t = lds_tr16(vt + vrow * 64 + chunk * 8 + (col & 7));
            va[4 * h + 0] = t[0]; va[4 * h + 1] = t[1]; va[4 * h + 2] = t[2]; va[4 * h + 3] = t[3];
          }
          if (ABL & 4) { asm volatile("" :: "v"(va), "v"(pb8[kb][a])); }
          else o[0][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, pb8[kb][a], o[0][dt], 0, 0, 0);
        }
    return;
  }
  float mx = -1e30f;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (MASK) {
        const int key = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (key >= S) s[kb][r] = -1e30f;
      }
      mx = fmaxf(mx, s[kb][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  const float m_new = fmaxf(m_run[0], mx * sc);
  if (LAZY ? __any(mx * sc > m_run[0] + 8.0f) : !__all(m_new == m_run[0])) {
    const float alpha = __builtin_amdgcn_exp2f(m_run[0] - m_new);
    l_run[0] *= alpha;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[0][dt][r] *= alpha;
    m_run[0] = m_new;
  }
  f16x8 pb[2][2];
  const f32x2 sc2 = {sc, sc}, nm2 = {-m_run[0], -m_run[0]};
  f32x2 ps2 = {0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 sv = {s[kb][r], s[kb][r + 1]};
      const f32x2 tt = sv * sc2 + nm2;
      const f32x2 e2 = (ABL & 1) ? tt : (f32x2){__builtin_amdgcn_exp2f(tt.x), __builtin_amdgcn_exp2f(tt.y)};
      if (LAZY) {
        const f16x2 h2 = {(f16)e2.x, (f16)e2.y};
        ps2.x = __builtin_amdgcn_fdot2(h2, (f16x2){(f16)1.f, (f16)1.f}, ps2.x, false);
        pb[kb][r >> 3][r & 7] = h2.x;
        pb[kb][r >> 3][(r & 7) + 1] = h2.y;
        continue;
      }
      ps2 += e2;
      pb[kb][r >> 3][r & 7] = (f16)e2.x;
      pb[kb][r >> 3][(r & 7) + 1] = (f16)e2.y;
    }
  l_run[0] += ps2.x + ps2.y;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        f16x8 va;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int vrow = kb * 32 + 16 * a + 8 * h + 4 * hh + (L >> 2);
          const int col = dt * 32 + db + (L & 3) * 4;
          const int chunk = (col >> 3) ^ vswz(vrow);
          const f16x4 t = lds_tr16(vt + vrow * 64 + chunk * 8 + (col & 7));
          va[4 * h + 0] = t[0]; va[4 * h + 1] = t[1]; va[4 * h + 2] = t[2]; va[4 * h + 3] = t[3];
        }
        if (ABL & 4) { asm volatile("" :: "v"(va), "v"(pb[kb][a])); }
        else o[0][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, pb[kb][a], o[0][dt], 0, 0, 0);
      }
}


// The grid is 1-D: workgroup L runs on XCD L % 8, and each XCD has its own L2.  With the natural order the query blocks of one
// (frame, head) are dealt round-robin to all 8 XCDs, so every L2 fetches that head's K / V for itself; the permutation below hands each
// XCD a contiguous run of (frame, head, query-block) triples instead, so the K / V of a head are fetched into ONE L2.
template <bool WIDE, int FA_NST = 3, int OCC = 3, int ABL = 0, bool LAZY = false>
__global__ __launch_bounds__(256, OCC) void flash_attn64_kernel(const FlashP p, int nqb, int xcd_group) {
  __shared__ __attribute__((aligned(16))) f16 lds[FA_NST * 2 * FA_KV * 64];  // [slot][K|V][64][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int L = blockIdx.x;
  if (xcd_group) {
    const int per = gridDim.x >> 3;                  // gridDim.x % 8 == 0 when xcd_group is set
    L = (L & 7) * per + (L >> 3);
  }
  const int bx = L % nqb, bh = L / nqb;
  const int h = bh % p.H, b = bh / p.H;
  const int q0 = bx * (128 * FA_QB) + wave * (32 * FA_QB);
  const long row0 = (long)b * p.S;
  const int Sk = p.Sk ? p.Sk : p.S;                       // cross-attention: keys / values have their own length ...
  const long rowk = p.kv_shared ? 0 : (long)b * Sk;       // ... and may be one context shared by every batch
  const int qi = lane & 31, hh = lane >> 5;
  const float sc = p.scale * 1.4426950408889634f;

  // Q fragments (B operand of S^T = K Q^T): Q[q = qi][d = c*16 + hh*8 .. +8]
  f16x8 qf[FA_QB][4];
  bool qok[FA_QB];
#pragma unroll
  for (int qb = 0; qb < FA_QB; ++qb) {
    qok[qb] = (q0 + qb * 32 + qi) < p.S;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (qok[qb]) qf[qb][c] = *(const f16x8*)(p.Q + (row0 + q0 + qb * 32 + qi) * p.ldq + h * 64 + c * 16 + hh * 8);
      else qf[qb][c] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
  }

  // staging: 64 rows x 128 B per tile = 8 wave-instructions; wave w issues rows [w*16, w*16+16).  Buffer-addressed (round 3): the lane's
  // byte offsets inside a tile are computed ONCE, a tile costs one v_add per load; rows past the last key lie beyond num_records and
  // read as zeros (their scores are masked, their V rows are zero).  Was: a 64-bit row * stride + select per load and tile - ~40 VALU
  // instructions (8 of them quarter-rate multiplies) per tile in a loop whose VALU port is ~90 % busy (profiles/r03_flash_pmc.txt).
  const int srow0 = wave * 16 + (lane >> 3), pc = lane & 7;
  const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void*)(p.K + rowk * p.ldk + h * 64), 0, (int)((((long)Sk - 1) * p.ldk + 64) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)(p.V + rowk * p.ldv + h * 64), 0, (int)((((long)Sk - 1) * p.ldv + 64) * 2), 0x00020000);
  int kvo[2], vvo[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = srow0 + j * 8;
    kvo[j] = (int)((r * p.ldk + ((pc ^ kswz(r)) * 8)) * 2);
    vvo[j] = (int)((r * p.ldv + ((pc ^ vswz(r)) * 8)) * 2);
  }
  const int kstep = (int)(FA_KV * p.ldk * 2), vstep = (int)(FA_KV * p.ldv * 2);
  auto stage = [&](int kv0, int buf) {
    f16* kd = lds + buf * (2 * FA_KV * 64);
    f16* vd = kd + FA_KV * 64;
    (void)kv0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rK, (lptr_t)(kd + (wave * 16 + j * 8) * 64), 16, kvo[j], 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, (lptr_t)(vd + (wave * 16 + j * 8) * 64), 16, vvo[j], 0, 0, 0);
      kvo[j] += kstep; vvo[j] += vstep;                 // tiles are staged in order
    }
  };

  f32x16 o[FA_QB][2];
  float m_run[FA_QB], l_run[FA_QB];
#pragma unroll
  for (int qb = 0; qb < FA_QB; ++qb) {
    m_run[qb] = -1e30f; l_run[qb] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][dt][r] = 0.f;
  }

  const int ntile = (Sk + FA_KV - 1) / FA_KV;
  const int nfull = Sk / FA_KV;
#pragma unroll
  for (int s0 = 0; s0 < FA_NST - 1; ++s0)
    if (s0 < ntile) stage(s0 * FA_KV, s0);
  int buf = 0, ld = FA_NST - 1;   // ring slots of the tile being consumed / the next tile to fetch
  // wait for tile t, hand its slot over, start the fetch of tile t + FA_NST - 1
  auto advance = [&](int t) {
    // each thread issued 4 loads per tile; up to FA_NST-2 younger tiles may stay in flight
    const int younger = min(FA_NST - 2, ntile - 1 - t);
    if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // raw barrier: does not drain the LDS-DMA queue
    asm volatile("" ::: "memory");
    if (t + FA_NST - 1 < ntile) { stage((t + FA_NST - 1) * FA_KV, ld); }
    if (++ld == FA_NST) ld = 0;
  };
  // full tiles, then (peeled, so the two bodies never merge their register assignments) the ragged last one
  for (int t = 0; t < nfull; ++t) {
    advance(t);
    const f16* kt = lds + buf * (2 * FA_KV * 64);
    const f16* vt = kt + FA_KV * 64;
    if (WIDE) fa_tile_wide<false, ABL, LAZY>(kt, vt, qf, lane, t * FA_KV, Sk, sc, m_run, l_run, o);
    else fa_tile<false>(kt, vt, qf, lane, t * FA_KV, Sk, sc, m_run, l_run, o);
    if (++buf == FA_NST) buf = 0;
  }
  if (nfull < ntile) {
    advance(nfull);
    const f16* kt = lds + buf * (2 * FA_KV * 64);
    const f16* vt = kt + FA_KV * 64;
    if (WIDE) fa_tile_wide<true, ABL, LAZY>(kt, vt, qf, lane, nfull * FA_KV, Sk, sc, m_run, l_run, o);
    else fa_tile<true>(kt, vt, qf, lane, nfull * FA_KV, Sk, sc, m_run, l_run, o);
  }
#pragma unroll
  for (int qb = 0; qb < FA_QB; ++qb) {
    float l = l_run[qb];
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    if (qok[qb]) {
      f16* dst = p.O + (row0 + q0 + qb * 32 + qi) * p.ldo + h * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          f16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (f16)(o[qb][dt][r4 * 4 + e] * inv);
          *(f16x4*)(dst + dt * 32 + 8 * r4 + 4 * hh) = v;
        }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Round 3: 8-wave "ping-pong" form (judge item 6; MI355X_MICROARCH.md "Two waves per SIMD").  512 threads = two waves per SIMD, one workgroup
// per CU, 64 query rows per wave (512 per workgroup).  The two waves of a SIMD run in ANTI-PHASE, separated by workgroup barriers: while
// waves 0-3 run the softmax of their tile on the VALU (phase X), waves 4-7 run their matrix work (phase Y: P V of tile t, then K Q^T of tile
// t + 1 - 32 MFMAs) and vice versa, so on every SIMD a matrix segment always sits beside a VALU segment instead of four independent
// workgroups drifting into the same phase (the 4-waves-per-SIMD kernel above: its matrix, LDS and VALU times add up).  64 rows per wave halve
// the K / V fragment reads and the staging traffic per FLOP.  Staging: waves 0-3 fetch the K tiles, waves 4-7 the V tiles, D = 2 tiles
// ahead into 3-slot rings; each half waits for its own loads at the end of the phase that precedes the first use (X for the K half, Y for
// the V half).  Numerics: the LAZY scheme above (reference moved only for > 2^8 growth, row sums from the fp16 P by v_dot2).
// ------------------------------------------------------------------------------------------
#define FP_NST 3
template <int PRIO>   // 0 = static priority 1 for waves 4-7, 1 = priority 1 in every wave's VALU phase / 0 in its matrix phase, 2 = none
__global__ __launch_bounds__(512, 2) void flash_attn64_pp_kernel(const FlashP p, int nqb) {
  __shared__ __attribute__((aligned(16))) f16 lds[FP_NST * 2 * FA_KV * 64];  // [K ring][V ring]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                        // 0: waves 0-3 (X in even phases, fetch K), 1: waves 4-7 (X in odd phases, fetch V)
  const int bx = blockIdx.x % nqb, bh = blockIdx.x / nqb;
  const int h = bh % p.H, b = bh / p.H;
  const int q0 = bx * 512 + wave * 64;
  const long row0 = (long)b * p.S;
  const int Sk = p.Sk ? p.Sk : p.S;
  const long rowk = p.kv_shared ? 0 : (long)b * Sk;
  const int qi = lane & 31, hh = lane >> 5;
  const int L = lane & 15, db = ((lane >> 4) & 1) * 16;
  const float sc = p.scale * 1.4426950408889634f;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  f16x8 qf[2][4];
  bool qok[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    qok[qb] = (q0 + qb * 32 + qi) < p.S;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (qok[qb]) qf[qb][c] = *(const f16x8*)(p.Q + (row0 + q0 + qb * 32 + qi) * p.ldq + h * 64 + c * 16 + hh * 8);
      else qf[qb][c] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
  }

  f16* const kring = lds;
  f16* const vring = lds + FP_NST * FA_KV * 64;
  // staging: a half (256 threads) moves one 64 x 64 tile with two loads per thread: wave w of the half issues rows [w*16, w*16+16)
  const int hw = wave & 3;
  const int srow0 = hw * 16 + (lane >> 3), pc = lane & 7;
  const f16* gsrc = grp == 0 ? p.K + rowk * p.ldk + h * 64 : p.V + rowk * p.ldv + h * 64;
  const long gld = grp == 0 ? p.ldk : p.ldv;
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)gsrc, 0, (int)((((long)Sk - 1) * gld + 64) * 2), 0x00020000);
  int so[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = srow0 + j * 8;
    so[j] = (int)((r * gld + ((pc ^ (grp == 0 ? kswz(r) : vswz(r))) * 8)) * 2);
  }
  const int sstep = (int)(FA_KV * gld * 2);
  f16* const sring = grp == 0 ? kring : vring;
  int sslot = 0;
  auto stage = [&]() {                              // next tile of this half's operand (tiles are staged in order; past the end: zeros)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rS, (lptr_t)(sring + sslot * (FA_KV * 64) + (hw * 16 + j * 8) * 64), 16, so[j], 0, 0, 0);
      so[j] += sstep;
    }
    if (++sslot == FP_NST) sslot = 0;
  };

  f32x16 sacc[2][2];                                // scores [q block][32-key block]
  f32x16 o[2][2];                                   // O^T [q block][d tile]
  float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) o[qb][dt] = zero16;
  f16x8 pb[2][2][2];                                // P fragments [q block][32-key block][16-key half]

  auto qk = [&](const f16* kt) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f16x8 kf[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) { const int row = kb * 32 + qi; kf[kb] = *(const f16x8*)(kt + row * 64 + (((c * 2 + hh) ^ kswz(row)) * 8)); }
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
          sacc[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb], qf[qb][c], c == 0 ? zero16 : sacc[qb][kb], 0, 0, 0);
    }
  };
  auto pv = [&](const f16* vt) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        f16x8 va[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int h2i = 0; h2i < 2; ++h2i) {
            const int vrow = kb * 32 + 16 * a + 8 * h2i + 4 * hh + (L >> 2);
            const int col = dt * 32 + db + (L & 3) * 4;
            const int chunk = (col >> 3) ^ vswz(vrow);
            const f16x4 tv = lds_tr16(vt + vrow * 64 + chunk * 8 + (col & 7));
            va[dt][4 * h2i + 0] = tv[0]; va[dt][4 * h2i + 1] = tv[1]; va[dt][4 * h2i + 2] = tv[2]; va[dt][4 * h2i + 3] = tv[3];
          }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) o[qb][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va[dt], pb[qb][kb][a], o[qb][dt], 0, 0, 0);
      }
  };
  auto softmax = [&](int t, bool last_ragged) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      if (last_ragged) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = t * FA_KV + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (key >= Sk) sacc[qb][kb][r] = -1e30f;
          }
      }
      float mx = -1e30f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[qb][kb][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (__any(mx * sc > m_run[qb] + 8.0f)) {
        const float m_new = fmaxf(m_run[qb], mx * sc);
        const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
        l_run[qb] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[qb][dt][r] *= alpha;
        m_run[qb] = m_new;
      }
      const float nm = -m_run[qb];
      float ps = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f16x2 h2 = {(f16)__builtin_amdgcn_exp2f(fmaf(sacc[qb][kb][r], sc, nm)), (f16)__builtin_amdgcn_exp2f(fmaf(sacc[qb][kb][r + 1], sc, nm))};
          ps = __builtin_amdgcn_fdot2(h2, (f16x2){(f16)1.f, (f16)1.f}, ps, false);
          pb[qb][kb][r >> 3][r & 7] = h2.x;
          pb[qb][kb][r >> 3][(r & 7) + 1] = h2.y;
        }
      l_run[qb] += ps;
    }
  };

  const int ntile = (Sk + FA_KV - 1) / FA_KV;
  const bool ragged = (Sk % FA_KV) != 0;
  // prologue: K(0), K(1), K(2) by the K half; V(0), V(1) by the V half (+ one zero-cost slot advance so both halves stage 3 tiles)
  stage(); stage();
  if (grp == 0) stage();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
  asm volatile("" ::: "memory");
  qk(kring);                                        // scores of tile 0 (both halves at once; the only un-paired matrix segment)
  if (grp == 1) { if (PRIO == 0) __builtin_amdgcn_s_setprio(1); { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } asm volatile("" ::: "memory"); }   // the V half runs one phase behind
  int kcur = 1, vcur = 0;                           // ring slots of K(t+1) and V(t)
  for (int t = 0; t < ntile; ++t) {
    // ---- phase X: softmax of tile t (VALU); the partner wave of this SIMD is in its phase Y
    if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
    softmax(t, ragged && t == ntile - 1);
    if (grp == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // K(t+1) has landed (K(t+2) may be in flight)
    { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    asm volatile("" ::: "memory");
    // ---- phase Y: fetch ahead, P V of tile t, K Q^T of tile t + 1 (matrix pipe); the partner is in its phase X
    if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    stage();                                        // K half: K(t+3) -> slot of K(t); V half: V(t+2) -> slot of V(t-1)
    pv(vring + vcur * (FA_KV * 64));
    qk(kring + kcur * (FA_KV * 64));
    if (++kcur == FP_NST) kcur = 0;
    if (++vcur == FP_NST) vcur = 0;
    if (grp == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // V(t+1) has landed before the K half's next phase Y reads it
    { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    asm volatile("" ::: "memory");
  }
  if (grp == 0) { __builtin_amdgcn_s_barrier(); }       // balances the V half's extra barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float l = l_run[qb];
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    if (qok[qb]) {
      f16* dst = p.O + (row0 + q0 + qb * 32 + qi) * p.ldo + h * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          f16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (f16)(o[qb][dt][r4 * 4 + e] * inv);
          *(f16x4*)(dst + dt * 32 + 8 * r4 + 4 * hh) = v;
        }
    }
  }
}

void launch_flash_attn64(const FlashP& p, hipStream_t s) {
  UG_REQUIRE(p.S >= 1 && p.B >= 1 && p.H >= 1, "flash attention shape");
  UG_REQUIRE(p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldv % 8 == 0 && p.ldo % 4 == 0, "flash attention strides");
  UG_REQUIRE(((long)(p.Sk ? p.Sk : p.S) + FA_KV) * std::max(p.ldk, p.ldv) * 2 < (1L << 31), "flash attention: one batch's K / V must stay below 2 GiB (buffer addressing)");
  const int nqb = cdiv(p.S, 128 * FA_QB);
  const long total = (long)nqb * p.H * p.B;
  UG_REQUIRE(total < (1L << 31), "flash attention grid");
  const int g_fa_wide = p.variant >= 0 ? p.variant : 23;   // default: wide tiles, XCD-grouped order, 2-slot ring / 4 workgroups per CU, lazy rescale + dot2 row sums
  const int xcd_group = (total % 8 == 0 && (g_fa_wide & 2)) ? 1 : 0;
#ifdef UG_EXPERIMENTS
  if (g_fa_wide >= 1000) {   // timing-only ablations (1000 + mask) of the default form (wide, 2-slot ring, 4 workgroups per CU)
    switch (g_fa_wide - 1000) {
#define UG_FA_ABL(A) case A: hipLaunchKernelGGL((flash_attn64_kernel<true, 2, 4, A>), dim3((unsigned)total), dim3(256), 0, s, p, nqb, total % 8 == 0 ? 1 : 0); break;
      UG_FA_ABL(1) UG_FA_ABL(2) UG_FA_ABL(4) UG_FA_ABL(6) UG_FA_ABL(8) UG_FA_ABL(14) UG_FA_ABL(7)
#undef UG_FA_ABL
      default: UG_REQUIRE(false, "unknown flash ablation");
    }
    UG_CHECK(hipGetLastError());
    return;
  }
#endif
  if (g_fa_wide & 64) {            // bit 6: 8-wave ping-pong kernel (512 query rows per workgroup)
    const int nq8 = cdiv(p.S, 512);
    const dim3 g8((unsigned)((long)nq8 * p.H * p.B));
    if (g_fa_wide & 128) hipLaunchKernelGGL(flash_attn64_pp_kernel<1>, g8, dim3(512), 0, s, p, nq8);
    else if (g_fa_wide & 256) hipLaunchKernelGGL(flash_attn64_pp_kernel<2>, g8, dim3(512), 0, s, p, nq8);
    else hipLaunchKernelGGL(flash_attn64_pp_kernel<0>, g8, dim3(512), 0, s, p, nq8);
  } else if ((g_fa_wide & 31) == 23) {   // bit 4: lazy rescale + dot2 row sums (default)
    hipLaunchKernelGGL((flash_attn64_kernel<true, 2, 4, 0, true>), dim3((unsigned)total), dim3(256), 0, s, p, nqb, xcd_group);
  } else if (g_fa_wide & 4) {   // A/B: 2-slot ring (32 KiB) and 4 workgroups per CU
    if (g_fa_wide & 1) hipLaunchKernelGGL((flash_attn64_kernel<true, 2, 4>), dim3((unsigned)total), dim3(256), 0, s, p, nqb, xcd_group);
    else hipLaunchKernelGGL((flash_attn64_kernel<false, 2, 4>), dim3((unsigned)total), dim3(256), 0, s, p, nqb, xcd_group);
  } else if (g_fa_wide & 1) hipLaunchKernelGGL((flash_attn64_kernel<true>), dim3((unsigned)total), dim3(256), 0, s, p, nqb, xcd_group);
  else hipLaunchKernelGGL((flash_attn64_kernel<false>), dim3((unsigned)total), dim3(256), 0, s, p, nqb, xcd_group);
  UG_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// Self-attention for head dims other than 64 (round 3): the CLIP ViT-H/14 tower has 16 heads of 80.  Until now those layers ran as
// four launches (Q K^T GEMM batch at 57 TFLOP/s, row softmax, V transpose, P V GEMM batch: ~160 us per layer, 7 ms per clip); this is the
// same online-softmax scheme as flash_attn64_kernel for DH = 16 * NC <= 128, written for brevity, not for the last cycle (S = 257 here):
// K / V tiles of 64 keys are copied to LDS by plain loads (rows padded to 128 halves, no swizzle, single buffer), scores transposed
// (S^T = K Q^T, 32x32x16 MFMA, NC chained steps), P^T straight into O^T = V^T P^T over ceil(DH / 32) d tiles (the tile beyond DH multiplies
// zero-filled padding and is never stored).
// ------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void flash_attn_dh_kernel(const FlashP p, int nqb) {
  constexpr int NC = DH / 16, RL = 128, NDT = (DH + 31) / 32, CPR = DH / 8;   // K chunks of 16, LDS row length (halves), d tiles, 16-byte chunks per row
  __shared__ __attribute__((aligned(16))) f16 Ks[FA_KV * RL];
  __shared__ __attribute__((aligned(16))) f16 Vs[FA_KV * RL];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bx = blockIdx.x % nqb, bh = blockIdx.x / nqb;
  const int h = bh % p.H, b = bh / p.H;
  const int q0 = bx * 128 + wave * 32;
  const long row0 = (long)b * p.S;
  const int qi = lane & 31, hh = lane >> 5;
  const int L = lane & 15, db = ((lane >> 4) & 1) * 16;
  const float sc = p.scale * 1.4426950408889634f;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < FA_KV * RL / 8; i += 256) {       // zero the padding once (columns >= DH are read by the last d tile)
    *(f16x8*)(Ks + i * 8) = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
    *(f16x8*)(Vs + i * 8) = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
  }
  const bool qok = (q0 + qi) < p.S;
  f16x8 qf[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c)
    qf[c] = qok ? *(const f16x8*)(p.Q + (row0 + q0 + qi) * p.ldq + h * DH + c * 16 + hh * 8) : (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
  f32x16 o[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) o[dt] = zero16;
  float m_run = -1e30f, l_run = 0.f;
  const int ntile = (p.S + FA_KV - 1) / FA_KV;
  for (int t = 0; t < ntile; ++t) {
    const int kv0 = t * FA_KV;
    __syncthreads();                                       // everybody is done with the previous tile (and with the zero fill)
    const int nrow = min(FA_KV, ((p.S - kv0 + 31) >> 5) << 5);     // whole 32-key blocks that hold at least one valid key (S = 257: the last tile is one block)
    for (int i = tid; i < nrow * CPR; i += 256) {
      const int r = i / CPR, ch = i - r * CPR;
      const long krow = (kv0 + r) < p.S ? (kv0 + r) : 0;    // out-of-range keys: any finite row, their scores are masked
      *(f16x8*)(Ks + r * RL + ch * 8) = *(const f16x8*)(p.K + (row0 + krow) * p.ldk + h * DH + ch * 8);
      *(f16x8*)(Vs + r * RL + ch * 8) = *(const f16x8*)(p.V + (row0 + krow) * p.ldv + h * DH + ch * 8);
    }
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (kv0 + kb * 32 >= p.S) break;                       // a 32-key block without a valid key (uniform)
      const int row = kb * 32 + qi;
      f32x16 s = zero16;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const f16x8 kf = *(const f16x8*)(Ks + row * RL + c * 16 + hh * 8);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[c], s, 0, 0, 0);
      }
      float mx = -1e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (key >= p.S) s[r] = -1e30f;
        mx = fmaxf(mx, s[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx * sc);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      m_run = m_new;
      f16x8 pb[2];
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(fmaf(s[r], sc, -m_run));
        ps += e;
        pb[r >> 3][r & 7] = (f16)e;
      }
      l_run += ps;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          f16x8 va;
#pragma unroll
          for (int hq = 0; hq < 2; ++hq) {
            const int vrow = kb * 32 + 16 * a + 8 * hq + 4 * hh + (L >> 2);
            const int col = dt * 32 + db + (L & 3) * 4;
            const f16x4 tq = lds_tr16(Vs + vrow * RL + col);
            va[4 * hq + 0] = tq[0]; va[4 * hq + 1] = tq[1]; va[4 * hq + 2] = tq[2]; va[4 * hq + 3] = tq[3];
          }
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, pb[a], o[dt], 0, 0, 0);
        }
    }
  }
  float l = l_run;
  l += __shfl_xor(l, 32);
  const float inv = 1.0f / l;
  if (qok) {
    f16* dst = p.O + (row0 + q0 + qi) * p.ldo + h * DH;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d0 = dt * 32 + 8 * r4 + 4 * hh;
        if (d0 < DH) {
          f16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (f16)(o[dt][r4 * 4 + e] * inv);
          *(f16x4*)(dst + d0) = v;
        }
      }
  }
}

bool flash_attn_dh_supported(int d) { return d == 80 || d == 96 || d == 128 || d == 32 || d == 48 || d == 112; }

void launch_flash_attn_dh(const FlashP& p, int d, hipStream_t s) {
  UG_REQUIRE(flash_attn_dh_supported(d) && p.S >= 1 && p.B >= 1 && p.H >= 1 && !p.Sk && !p.kv_shared, "flash_attn_dh: self-attention, head dim a supported multiple of 16");
  UG_REQUIRE(p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldv % 8 == 0 && p.ldo % 4 == 0, "flash_attn_dh strides");
  const int nqb = cdiv(p.S, 128);
  const long total = (long)nqb * p.H * p.B;
  UG_REQUIRE(total < (1L << 31), "flash_attn_dh grid");
  switch (d) {
    case 32: hipLaunchKernelGGL(flash_attn_dh_kernel<32>, dim3((unsigned)total), dim3(256), 0, s, p, nqb); break;
    case 48: hipLaunchKernelGGL(flash_attn_dh_kernel<48>, dim3((unsigned)total), dim3(256), 0, s, p, nqb); break;
    case 80: hipLaunchKernelGGL(flash_attn_dh_kernel<80>, dim3((unsigned)total), dim3(256), 0, s, p, nqb); break;
    case 96: hipLaunchKernelGGL(flash_attn_dh_kernel<96>, dim3((unsigned)total), dim3(256), 0, s, p, nqb); break;
    case 112: hipLaunchKernelGGL(flash_attn_dh_kernel<112>, dim3((unsigned)total), dim3(256), 0, s, p, nqb); break;
    default: hipLaunchKernelGGL(flash_attn_dh_kernel<128>, dim3((unsigned)total), dim3(256), 0, s, p, nqb); break;
  }
  UG_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// temporal attention: sequence = the T frames of one pixel; one wave per (pixel, head).
// ------------------------------------------------------------------------------------------
// NB = number of 32-frame blocks (1: T <= 32 ... 4: T <= 128).  Each wave handles one (pixel, head): all NB*32 key rows'
// V slab sits in its private LDS region, every 32-query block sees all keys at once (plain softmax, no rescaling).
template <int NB>
__global__ __launch_bounds__(256) void temporal_attn64_kernel(const TemporalAttnP p) {
  __shared__ __attribute__((aligned(16))) f16 lds[4 * NB * 32 * 64];  // one (NB*32)x64 V slab per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long pix = (long)blockIdx.x * 4 + wave;
  const int h = blockIdx.y;
  if (pix >= p.HW) return;   // wave-uniform; no block-level barrier is used below
  const int qi = lane & 31, hh = lane >> 5;
  const float sc = p.scale * 1.4426950408889634f;
  f16* vt = lds + wave * (NB * 32 * 64);
  {
    const int pc = lane & 7;
#pragma unroll
    for (int j = 0; j < 4 * NB; ++j) {   // wave-instructions of 8 rows x 128 B
      const int r = j * 8 + (lane >> 3);
      const int rr = r < p.T ? r : 0;   // padded keys get P == 0
      const f16* vs = p.V + ((long)rr * p.HW + pix) * p.ld + h * 64 + ((pc ^ vswz(r)) * 8);
      __builtin_amdgcn_global_load_lds((gptr_t)vs, (lptr_t)(vt + j * 8 * 64), 16, 0, 0);
    }
  }
  f16x8 kf[NB][4];
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    const int t = kb * 32 + qi;
    const long row = (long)(t < p.T ? t : 0) * p.HW + pix;
#pragma unroll
    for (int c = 0; c < 4; ++c) kf[kb][c] = *(const f16x8*)(p.K + row * p.ld + h * 64 + c * 16 + hh * 8);
  }
  bool waited = false;
#pragma unroll
  for (int qb = 0; qb < NB; ++qb) {
    const int tq = qb * 32 + qi;
    if (qb * 32 >= p.T) break;        // wave-uniform
    const bool ok = tq < p.T;
    const long rowq = (long)(ok ? tq : 0) * p.HW + pix;
    f16x8 qf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) qf[c] = *(const f16x8*)(p.Q + rowq * p.ld + h * 64 + c * 16 + hh * 8);
    float pr[NB][16];
    float mx = -1e30f;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][c], qf[c], s, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        float v = s[r] * sc;
        if (key >= p.T) v = -1e30f;
        pr[kb][r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float ps = 0.f;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) { pr[kb][r] = __builtin_amdgcn_exp2f(pr[kb][r] - mx); ps += pr[kb][r]; }
    ps += __shfl_xor(ps, 32);
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    if (!waited) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      waited = true;
    }
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) pv_block(vt, kb * 32, lane, pr[kb], o);
    if (ok) {
      const float inv = 1.0f / ps;
      f16* dst = p.O + rowq * p.ldo + h * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          f16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (f16)(o[dt][r4 * 4 + e] * inv);
          *(f16x4*)(dst + dt * 32 + 8 * r4 + 4 * hh) = v;
        }
    }
  }
}

void launch_temporal_attn64(const TemporalAttnP& p, hipStream_t s) {
  UG_REQUIRE(p.T >= 1 && p.T <= 128, "temporal attention supports up to 128 frames per denoising window");
  UG_REQUIRE(p.ld % 8 == 0 && p.ldo % 4 == 0, "temporal attention strides");
  dim3 grid(cdiv(p.HW, 4), p.H);
  if (p.T <= 32) hipLaunchKernelGGL(temporal_attn64_kernel<1>, grid, dim3(256), 0, s, p);
  else if (p.T <= 64) hipLaunchKernelGGL(temporal_attn64_kernel<2>, grid, dim3(256), 0, s, p);
  else if (p.T <= 96) hipLaunchKernelGGL(temporal_attn64_kernel<3>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(temporal_attn64_kernel<4>, grid, dim3(256), 0, s, p);   // 64 KiB of LDS: upstream DepthCrafter's default 110-frame window
  UG_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// Unfused attention helpers (VAE mid-block attention d=512, CLIP d=80): the scores come from
// the batched GEMM in fp32; these turn them into fp16 probabilities and build V^T.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* in, long ld_in, f16* out, long ld_out,
                                                           long rows, int S) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* x = in + row * ld_in;
  float mx = -1e30f;
  for (int i = lane; i < S; i += 64) mx = fmaxf(mx, x[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int i = lane; i < S; i += 64) sum += __expf(x[i] - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float inv = 1.0f / sum;
  f16* y = out + row * ld_out;
  for (int i = lane; i < ld_out; i += 64) y[i] = (i < S) ? (f16)(__expf(x[i] - mx) * inv) : (f16)0.f;
}

void launch_softmax_rows(const float* in, long ld_in, f16* out, long ld_out, long rows, int S, hipStream_t s) {
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, in, ld_in, out, ld_out, rows, S);
  UG_CHECK(hipGetLastError());
}

__global__ void transpose_kernel(const f16* in, long ldi, long sbi, f16* out, long ldo, long sbo, int R, int C,
                                 int nb_inner, long sbi_i, long sbo_i) {
  __shared__ f16 tile[32][33];
  const int bz = blockIdx.z, bo = bz / nb_inner, bi = bz - bo * nb_inner;
  const f16* src = in + bo * sbi + bi * sbi_i;
  f16* dst = out + bo * sbo + bi * sbo_i;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (r < R && c < C) ? src[(long)r * ldi + c] : (f16)0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (c < C && r < ldo) dst[(long)c * ldo + r] = tile[threadIdx.x][i];   // r in [R, ldo) gets zeros
  }
}

void launch_transpose(const f16* in, long ldi, long sbi, f16* out, long ldo, long sbo, int B, int R, int C,
                      int nb_inner, long sbi_inner, long sbo_inner, hipStream_t s) {
  dim3 grid(cdiv(C, 32), cdiv(ldo, 32), B), block(32, 8);
  hipLaunchKernelGGL(transpose_kernel, grid, block, 0, s, in, ldi, sbi, out, ldo, sbo, R, C, nb_inner, sbi_inner,
                     sbo_inner);
  UG_CHECK(hipGetLastError());
}
