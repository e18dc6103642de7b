// Device helpers shared by the GEMM kernels (kernels/gemm.hip) and the halo-staged 3x3 convolution (kernels/conv_halo.hip):
// activation functions, the LDS chunk swizzle, the tile walk and the per-tile epilogue.
#pragma once
#include "../common.h"
#include <type_traits>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// erf by Abramowitz-Stegun 7.1.26 (|abs err| < 1.5e-7, far below the fp16 output spacing): ~12 VALU ops + one
// exp instead of libm's erff - the GEGLU epilogue evaluates it 8x per output row per lane.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(1.0f + 0.3275911f * ax);
  const float y = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const float r = 1.0f - y * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   // hardware reciprocal, as kernels/norm.hip

// GEGLU on two (value, gate) pairs in packed fp32 (v_pk_fma/mul/add_f32 issue two lanes' worth per instruction):
//   h * g * Phi(g),  Phi(g) = 1/2 + sign(g) (1/2 - erfc(|g|/sqrt2)/2),  erfc by the same Abramowitz-Stegun 7.1.26 form as
// erf_as (coefficients pre-halved, argument scaling folded in).  Per pair of outputs: 2 rcp + 2 exp2 + ~16 VALU instead of
// ~50 for the scalar form - in the K = 320 projections the epilogue is as long as the K loop, so this is wall time.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 geglu2(f32x2 h, f32x2 g) {
  const f32x2 ag = {__builtin_fabsf(g.x), __builtin_fabsf(g.y)};
  const f32x2 u = ag * 0.23164189f + 1.0f;   // 1 + 0.3275911 |g| / sqrt(2)
  const f32x2 t = {__builtin_amdgcn_rcpf(u.x), __builtin_amdgcn_rcpf(u.y)};
  f32x2 y = t * 0.5307027145f - 0.7265760135f;
  y = y * t + 0.7107068705f;
  y = y * t - 0.142248368f;
  y = y * t + 0.127414796f;
  y = y * t;                                   // erfc(|g|/sqrt2) / 2 / exp(-g^2/2)
  const f32x2 w = (g * g) * -0.72134752044f;   // -g^2/2 * log2(e)
  const f32x2 e = {__builtin_amdgcn_exp2f(w.x), __builtin_amdgcn_exp2f(w.y)};
  const f32x2 q = 0.5f - y * e;
  const f32x2 phi = {0.5f + __builtin_copysignf(q.x, g.x), 0.5f + __builtin_copysignf(q.y, g.y)};
  return h * g * phi;
}

// 16-byte-chunk swizzle of a [rows][BK] fp16 LDS tile: makes 16 consecutive rows reading the same logical
// chunk land on 16 distinct 16-byte slots of the 256-byte bank row.
template <int BK> __device__ __forceinline__ int swz(int row) {
  if (BK == 64) return (row >> 1) & 7;   // 128-B rows, 2 rows per bank row
  else return (row >> 2) & 3;            // 64-B rows, 4 rows per bank row
}

// Tile id -> (tm, tn).  group_m <= 1: row-major (all N tiles of one M tile are neighbours: the workgroups of an XCD share ONE
// activation panel - right for im2col, whose A operand is re-read per tap, and for weights that fit the 4 MiB L2).  group_m = g > 1:
// ids walk g M-tiles x all N tiles column by column, so the ~32 tiles an XCD works on at a time form a g x (32/g) block and touch
// g + 32/g operand panels instead of 1 + 32 - for dense layers with weights far larger than the L2 the row-major walk re-fetched the
// whole weight matrix once per M tile (profiles/r02_pmc_traffic_per_shape.txt: 4800x10240x1280 read 524 MB for 88 MB of operands).
__device__ __forceinline__ void tile_coord(int tile, int ntm, int ntn, int group_m, int& tm, int& tn) {
  if (group_m <= 1) { tm = tile / ntn; tn = tile - tm * ntn; return; }
  const int per = group_m * ntn;
  const int grp = tile / per, local = tile - grp * per;
  const int first = grp * group_m;
  const int gm = min(ntm - first, group_m);
  tn = local / gm;
  tm = first + (local - tn * gm);
}
// + the frame-fastest permutation of the M tiles for temporal convolutions (GemmP::tm_T)
__device__ __forceinline__ void tile_coord_p(const GemmP& p, int tile, int ntm, int ntn, int& tm, int& tn) {
  tile_coord(tile, ntm, ntn, p.group_m, tm, tn);
  if (p.tm_T) { const int q = tm / p.tm_T; tm = (tm - q * p.tm_T) * p.tm_nb + q; }
}

// Persistent tile walk of workgroup w of nwg (workgroup w is dispatched to XCD w % 8; each XCD has its own 4 MiB L2).
//   owned (default since round 5): XCD x owns the CONTIGUOUS run [x * ntiles / 8, (x + 1) * ntiles / 8) of the walk order for the whole launch and its nwg / 8
//     workgroups step through it nwg / 8 tiles at a time - every round of an XCD continues where its previous round stopped, so an operand panel (the A rows of
//     an M tile with its N tiles, a group of tile_coord) is pulled into exactly one L2, except for the seven panels that straddle two XCDs' runs;
//   round-strided (rounds 1 - 4, UG_F_XCDROUND / knob 128): round i of the launch is the run [i * nwg, (i + 1) * nwg) cut into eight pieces - the XCD's second
//     round starts nwg tiles further on, and every round boundary x every XCD boundary can split a panel between two L2s (19200 x 640 x 2560 on 192 x 128 tiles,
//     groups of 4 x 5: 15 of the 25 groups were read by two XCDs);
//   no remap (UG_F_NOXCD / knob 2, or a grid that is not a multiple of 8): tile w + i * nwg.
// The number of rounds is the same for all three: ceil(ceil(ntiles / 8) / (nwg / 8)) == ceil(ntiles / nwg) when nwg % 8 == 0.
struct TileWalk { int first, step, count; };
__device__ __forceinline__ TileWalk tile_walk(int flags, int ntiles, int nwg, int w) {
  TileWalk t;
  if (nwg % 8 != 0 || (flags & UG_F_NOXCD)) { t.first = w; t.step = nwg; t.count = (ntiles - w + nwg - 1) / nwg; return t; }
  const int x = w & 7, j = w >> 3, npx = nwg >> 3;
  if (flags & UG_F_XCDROUND) { t.first = x * npx + j; t.step = nwg; t.count = (ntiles - t.first + nwg - 1) / nwg; return t; }
  const int lo = (int)(((long)x * ntiles) >> 3), hi = (int)(((long)(x + 1) * ntiles) >> 3);
  t.first = lo + j; t.step = npx; t.count = max(0, (hi - lo - j + npx - 1) / npx);
  return t;
}

template <int N> struct HVec;
template <> struct HVec<8> { typedef f16x8 type; };
template <> struct HVec<4> { typedef f16x4 type; };

// Epilogue operands fetched AHEAD of the epilogue (round 5, the producer / consumer kernel): a tile of a short-K layer lasts 4 - 8 us and every dependent round
// trip to L2 / HBM at the start of its epilogue (bias; the residual rows) keeps the CU's matrix pipes idle for 1 - 2 us of it (profiles/r05_ln_fold_and_epilogue_prefetch.txt).
// The compute waves issue these loads into registers while K steps are still to come - the column operands at the tile's first step, the row operands three steps
// before its last - and the epilogue finds them landed.  Raw loaded vectors only: any arithmetic on them would pull the wait forward.
template <int MT, int NT> struct EpiPre {
  static constexpr int WID = 4 * NT, CH = (WID % 8 == 0) ? 8 : 4, NV = WID / CH;
  typedef typename HVec<CH>::type hvec;
  hvec b1[NV], b2[NV];                                   // bias, bias2 (one vector for all rows) of the lane's columns
  hvec r1[MT][NV];                                       // R1 rows (fp16, not with GEGLU)
};
template <int MT, int NT, int WTM, int WTN>
__device__ __forceinline__ void epi_pre_cols(const GemmP& p, int n0, int wn, int lane, EpiPre<MT, NT>& pre) {
  typedef EpiPre<MT, NT> P;
  typedef typename P::hvec hvec;
  const int g = lane >> 4;
  const int nb = n0 + wn * WTN + g * P::WID;
  if (nb + P::WID > p.N) return;                        // tile_epilogue takes its scalar path for this lane
  if (p.bias) {
#pragma unroll
    for (int v = 0; v < P::NV; ++v) pre.b1[v] = *(const hvec*)(p.bias + nb + v * P::CH);
  }
  if (p.bias2) {
#pragma unroll
    for (int v = 0; v < P::NV; ++v) pre.b2[v] = *(const hvec*)(p.bias2 + nb + v * P::CH);
  }
}
// the residual-row prefetch applies exactly where tile_epilogue's vector path would load them: (see `vec` there)
__device__ __forceinline__ bool epi_pre_r1_ok(const GemmP& p) {
  return p.R1 && !(p.flags & (UG_F_R1_F32 | UG_F_GEGLU)) && (p.ldo & 7) == 0 && (p.ldr1 & 7) == 0 && (!p.R2 || (p.ldr2 & 7) == 0);
}
template <int MT, int NT, int WTM, int WTN>
__device__ __forceinline__ void epi_pre_rows(const GemmP& p, int m0, int n0, int wm, int wn, int lane, EpiPre<MT, NT>& pre) {
  typedef EpiPre<MT, NT> P;
  typedef typename P::hvec hvec;
  const int l15 = lane & 15, g = lane >> 4;
  const int nb = n0 + wn * WTN + g * P::WID;
  if (nb + P::WID > p.N) return;
  if (epi_pre_r1_ok(p)) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wm * WTM + i * 16 + l15;
      if (m < p.M) {
#pragma unroll
        for (int v = 0; v < P::NV; ++v) pre.r1[i][v] = *(const hvec*)(p.R1 + (long)m * p.ldr1 + nb + v * P::CH);
      }
    }
  }
}

// Per-tile epilogue shared by the GEMM kernels: the lane holds WID = 4*NT contiguous columns of rows
// m0 + wm*WTM + i*16 + (lane & 15).  Clears the accumulators for the next tile.
// Anything new in here has to be a COMPILE-TIME variant instantiated only where it is used: run-time branches in this function cost every GEMM kernel that
// inlines it its register allocation (round 5: the LayerNorm-fold extensions as run-time branches, ~20 VGPRs and an occupancy step; round 6: a split-K
// fix-up path that was never taken, 960 -> 1284 ms per clip - profiles/r06_split_k_ticket_fixup_rejected.txt).
// PRE: bias / bias2 / the R1 rows come from `pre` (filled by epi_pre_cols / epi_pre_rows for THIS tile) instead of being loaded here.
template <int MT, int NT, int WTM, int WTN, bool PRE = false>
__device__ __forceinline__ void tile_epilogue(const GemmP& p, f32x4 (&acc)[MT][NT], int m0, int n0, int wm, int wn, int lane,
                                              long out_off, const EpiPre<MT, NT>* pre = nullptr) {
  constexpr int WID = 4 * NT;
  constexpr int CH = (WID % 8 == 0) ? 8 : 4;   // vector width of the epilogue's loads / stores (WID = 20: 80-column wave tiles)
  typedef typename HVec<CH>::type hvec;
  const int l15 = lane & 15, g = lane >> 4;
  const bool geglu = (p.flags & UG_F_GEGLU) != 0;
  const bool of32 = (p.flags & UG_F_OUT_F32) != 0;
  const int Nout = geglu ? p.N / 2 : p.N;
  const int OW = geglu ? WID / 2 : WID;        // output columns of this lane
  const int nb = n0 + wn * WTN + g * WID;      // lane holds WID contiguous columns of its rows
  if (p.splitk > 1) {   // raw fp32 partials [split][M][N]
    float* P = p.partial + ((long)blockIdx.y * p.M) * p.N;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wm * WTM + i * 16 + l15;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = nb + j * 4;
        if (m < p.M && n + 4 <= p.N) *(f32x4*)(P + (long)m * p.N + n) = acc[i][j];
        acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    return;
  }
  const int ob = geglu ? nb / 2 : nb;          // first output column of this lane
  const bool full = (nb + WID <= p.N);
  const bool vec = full && ((p.ldo & 7) == 0) && (!p.R1 || (p.ldr1 & 7) == 0) &&
                   (!p.R2 || (p.ldr2 & 7) == 0) && (OW % CH == 0);
  float bv[WID];
#pragma unroll
  for (int e = 0; e < WID; ++e) bv[e] = 0.f;
  if (full) {
    if (p.bias) {
#pragma unroll
      for (int e = 0; e < WID; e += CH) {
        hvec b;
        if constexpr (PRE) b = pre->b1[e / CH]; else b = *(const hvec*)(p.bias + nb + e);
#pragma unroll
        for (int q = 0; q < CH; ++q) bv[e + q] += (float)b[q];
      }
    }
    if (p.bias2) {
#pragma unroll
      for (int e = 0; e < WID; e += CH) {
        hvec b;
        if constexpr (PRE) b = pre->b2[e / CH]; else b = *(const hvec*)(p.bias2 + nb + e);
#pragma unroll
        for (int q = 0; q < CH; ++q) bv[e + q] += (float)b[q];
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < WID; ++e)
      if (nb + e < p.N) {
        if (p.bias) bv[e] += (float)p.bias[nb + e];
        if (p.bias2) bv[e] += (float)p.bias2[nb + e];
      }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int m = m0 + wm * WTM + i * 16 + l15;
    if (p.halo_tw) { const int r = wm * WTM + i * 16 + l15; m = m0 + (r >> p.halo_lg) * p.Wo + (r & (p.halo_tw - 1)); }   // halo conv: 2-D pixel tile
    long orow = m;
    if (p.up_phase) {   // sub-pixel phase of a nearest-2x upsample conv: scatter to the (2y+a, 2x+b) output pixel
      const int hw = p.Ho * p.Wo;
      const int t = m / hw, rem = m - t * hw;
      const int y = rem / p.Wo, x = rem - y * p.Wo;
      const int ph = p.up_phase - 1;
      orow = ((long)t * 2 * p.Ho + 2 * y + (ph >> 1)) * (2 * p.Wo) + 2 * x + (ph & 1);
    }
    float v[WID];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[j * 4 + r] = acc[i][j][r] + bv[j * 4 + r];
      }
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (m >= p.M) continue;
    if (geglu) {
      if (WID == 16) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const f32x2 r = geglu2((f32x2){v[e], v[e + 1]}, (f32x2){v[8 + e], v[9 + e]});
          v[e] = r.x; v[e + 1] = r.y;
        }
      }
    }
    if (vec) {
#pragma unroll
      for (int e = 0; e < OW; e += CH) {
        float o[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) o[q] = p.c0 * v[e + q];
        if (p.R1) {
          if (p.flags & UG_F_R1_F32) {
            const float* R = (const float*)p.R1 + (long)m * p.ldr1 + ob + e;
#pragma unroll
            for (int q = 0; q < CH; q += 4) { const f32x4 r = *(const f32x4*)(R + q); o[q] += p.c1 * r[0]; o[q + 1] += p.c1 * r[1]; o[q + 2] += p.c1 * r[2]; o[q + 3] += p.c1 * r[3]; }
          } else {
            hvec r;
            if constexpr (PRE) { if (!geglu) r = pre->r1[i][e / CH]; else r = *(const hvec*)(p.R1 + (long)m * p.ldr1 + ob + e); }
            else r = *(const hvec*)(p.R1 + (long)m * p.ldr1 + ob + e);
#pragma unroll
            for (int q = 0; q < CH; ++q) o[q] += p.c1 * (float)r[q];
          }
        }
        if (p.R2) {
          const hvec r = *(const hvec*)(p.R2 + (long)m * p.ldr2 + ob + e);
#pragma unroll
          for (int q = 0; q < CH; ++q) o[q] += p.c2 * (float)r[q];
        }
        if (p.act == UG_ACT_SILU) {
#pragma unroll
          for (int q = 0; q < CH; ++q) o[q] = silu_f(o[q]);
        } else if (p.act == UG_ACT_GELU) {
#pragma unroll
          for (int q = 0; q < CH; ++q) o[q] = gelu_f(o[q]);
        }
        if (of32) {
          float* O = (float*)p.Out + out_off + orow * p.ldo + ob + e;
#pragma unroll
          for (int q = 0; q < CH; q += 4) *(f32x4*)(O + q) = (f32x4){o[q], o[q + 1], o[q + 2], o[q + 3]};
        } else {
          hvec h;
#pragma unroll
          for (int q = 0; q < CH; ++q) h[q] = (f16)o[q];
          *(hvec*)((f16*)p.Out + out_off + orow * p.ldo + ob + e) = h;
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < OW; ++e) {
        const int n = ob + e;
        if (n < Nout) {
          float o = p.c0 * v[e];
          if (p.R1) o += p.c1 * ((p.flags & UG_F_R1_F32) ? ((const float*)p.R1)[(long)m * p.ldr1 + n] : (float)p.R1[(long)m * p.ldr1 + n]);
          if (p.R2) o += p.c2 * (float)p.R2[(long)m * p.ldr2 + n];
          if (p.act == UG_ACT_SILU) o = silu_f(o);
          else if (p.act == UG_ACT_GELU) o = gelu_f(o);
          if (of32) ((float*)p.Out)[out_off + orow * p.ldo + n] = o;
          else ((f16*)p.Out)[out_off + orow * p.ldo + n] = (f16)o;
        }
      }
    }
  }
}

// The epilogue of the common case without its run-time variants (round 6): fp16 output, bias / bias2, at most ONE fp16 residual (c0 == 1), no activation, no GEGLU,
// no split-K, no scatter, 8-aligned leading dimensions, c1 == 1 (so that either contraction of c0 v + c1 r gives the same bits), N a multiple of the wave's column count (every wave is wholly inside or wholly outside N).  tile_epilogue
// decides all of that per 8-column chunk of every row block with uniform branches - ~110 branches and ~700 VALU instructions per wave and tile, which a
// short-K tile (K = 640: 240 MFMAs per wave) pays as 20 - 30 % of its time.  Same arithmetic in the same order as tile_epilogue's vector path (c0 * v with
// c0 == 1 is exact), so the outputs are bit-identical; the launch-uniform test is tile_epilogue_lean_ok.
__device__ __forceinline__ bool tile_epilogue_lean_ok(const GemmP& p, int wtn, bool halo = false) {
  return p.splitk <= 1 && !(p.flags & (UG_F_GEGLU | UG_F_OUT_F32 | UG_F_R1_F32)) && !p.R2 && p.act == UG_ACT_NONE && p.c0 == 1.0f && (halo || !p.halo_tw) && !p.up_phase &&
         (p.ldo & 7) == 0 && (!p.R1 || ((p.ldr1 & 7) == 0 && p.c1 == 1.0f)) && p.N % wtn == 0;
}
template <int MT, int NT, int WTM, int WTN, bool PRE, bool HALO = false>   // HALO: the halo convolution's 2-D pixel tile (rows -> pixels as in tile_epilogue)
__device__ __forceinline__ void tile_epilogue_lean(const GemmP& p, f32x4 (&acc)[MT][NT], int m0, int n0, int wm, int wn, int lane, long out_off,
                                                   const EpiPre<MT, NT>* pre) {
  constexpr int WID = 4 * NT;
  constexpr int CH = (WID % 8 == 0) ? 8 : 4, NV = WID / CH;
  typedef typename HVec<CH>::type hvec;
  const int l15 = lane & 15, g = lane >> 4;
  const int nb = n0 + wn * WTN + g * WID;
  const bool in = n0 + wn * WTN < p.N;          // uniform per wave (N % WTN == 0)
  float bv[WID];
#pragma unroll
  for (int e = 0; e < WID; ++e) bv[e] = 0.f;
  if (in) {
    if (p.bias) {
#pragma unroll
      for (int e = 0; e < WID; e += CH) {
        hvec b;
        if constexpr (PRE) b = pre->b1[e / CH]; else b = *(const hvec*)(p.bias + nb + e);
#pragma unroll
        for (int q = 0; q < CH; ++q) bv[e + q] += (float)b[q];
      }
    }
    if (p.bias2) {
#pragma unroll
      for (int e = 0; e < WID; e += CH) {
        hvec b;
        if constexpr (PRE) b = pre->b2[e / CH]; else b = *(const hvec*)(p.bias2 + nb + e);
#pragma unroll
        for (int q = 0; q < CH; ++q) bv[e + q] += (float)b[q];
      }
    }
  }
  f16* const Ob = (f16*)p.Out + out_off + nb;
  auto rows = [&](auto HAS_R1) {
    constexpr bool R1 = decltype(HAS_R1)::value;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      int m = m0 + wm * WTM + i * 16 + l15;
      if constexpr (HALO) { const int rr = wm * WTM + i * 16 + l15; m = m0 + (rr >> p.halo_lg) * p.Wo + (rr & (p.halo_tw - 1)); }
      const bool ok = in && m < p.M;
      hvec r[NV];
      if constexpr (R1 && !PRE) {
        if (ok) {
#pragma unroll
          for (int v = 0; v < NV; ++v) r[v] = *(const hvec*)(p.R1 + (long)m * p.ldr1 + nb + v * CH);
        }
      }
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        float o[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) o[q] = acc[i][(v * CH + q) >> 2][(v * CH + q) & 3] + bv[v * CH + q];
        if constexpr (R1) {
          hvec rr;
          if constexpr (PRE) rr = pre->r1[i][v]; else rr = r[v];
#pragma unroll
          for (int q = 0; q < CH; ++q) o[q] += p.c1 * (float)rr[q];
        }
        hvec h;
#pragma unroll
        for (int q = 0; q < CH; ++q) {
          asm("" : "+v"(o[q]));      // keeps the fp32 sum and its fp16 rounding two instructions: hipcc otherwise folds `(f16)fma(c1, r, s)` into ONE v_fma_mixlo_f16
          h[q] = (f16)o[q];          // (a single rounding - one fp16 ulp away from every other kernel's result in the double-rounding cases; the cross-config test found it)
        }
        if (ok) *(hvec*)(Ob + (long)m * p.ldo + v * CH) = h;
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  if (p.R1) rows(std::true_type{}); else rows(std::false_type{});
}

// The GEGLU projection's epilogue without its run-time variants (round 6; see tile_epilogue_lean): bias (+ bias2), GEGLU on the lane's 8 (value, gate) pairs, fp16
// store of 8 outputs per row - tile_epilogue's vector path with c0 == 1 and no residual / activation / fp32 output.
__device__ __forceinline__ bool tile_epilogue_geglu_lean_ok(const GemmP& p, int wtn) {
  return p.splitk <= 1 && (p.flags & UG_F_GEGLU) && !(p.flags & (UG_F_OUT_F32 | UG_F_R1_F32)) && !p.R1 && !p.R2 && p.act == UG_ACT_NONE && p.c0 == 1.0f && !p.halo_tw &&
         !p.up_phase && (p.ldo & 7) == 0 && p.N % wtn == 0;
}
template <int MT, int NT, int WTM, int WTN>
__device__ __forceinline__ void tile_epilogue_geglu_lean(const GemmP& p, f32x4 (&acc)[MT][NT], int m0, int n0, int wm, int wn, int lane, long out_off) {
  constexpr int WID = 4 * NT;
  static_assert(WID == 16, "GEGLU: 8 value + 8 gate columns per lane");
  const int l15 = lane & 15, g = lane >> 4;
  const int nb = n0 + wn * WTN + g * WID;
  const bool in = n0 + wn * WTN < p.N;
  float bv[WID];
#pragma unroll
  for (int e = 0; e < WID; ++e) bv[e] = 0.f;
  if (in) {
    if (p.bias) {
#pragma unroll
      for (int e = 0; e < WID; e += 8) { const f16x8 b = *(const f16x8*)(p.bias + nb + e);
#pragma unroll
        for (int q = 0; q < 8; ++q) bv[e + q] += (float)b[q]; }
    }
    if (p.bias2) {
#pragma unroll
      for (int e = 0; e < WID; e += 8) { const f16x8 b = *(const f16x8*)(p.bias2 + nb + e);
#pragma unroll
        for (int q = 0; q < 8; ++q) bv[e + q] += (float)b[q]; }
    }
  }
  f16* const Ob = (f16*)p.Out + out_off + nb / 2;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + wm * WTM + i * 16 + l15;
    float v[WID];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[j * 4 + r] = acc[i][j][r] + bv[j * 4 + r];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 h;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      f32x2 r = geglu2((f32x2){v[e], v[e + 1]}, (f32x2){v[8 + e], v[9 + e]});
      asm("" : "+v"(r));           // (no fusing of the last multiply with the fp16 rounding: tile_epilogue_lean)
      h[e] = (f16)r.x; h[e + 1] = (f16)r.y;
    }
    if (in && m < p.M) *(f16x8*)(Ob + (long)m * p.ldo) = h;
  }
}

// sum over the 16 lanes of a DPP row (the lanes that hold the 16 rows of one MFMA block for the same columns): xor 1, xor 2 inside the quads,
// then the mirrored half row and the mirrored row - every lane ends up with the total
__device__ __forceinline__ float row16_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true));
  return x;
}

// tile_epilogue for the launches that also hand GroupNorm its statistics (GemmP::stat_part, round 4): the wave's WTM rows are one statistics block
// (index stat_blk); per 8-column chunk the rows are stored as tile_epilogue stores them (same arithmetic, same roundings - bit-identical outputs)
// while the fp16 values written are summed per column (sum, sum of squares) in registers, reduced over the 16 row lanes and stored by lane 0 of each
// column group: stat_part[stat_blk * N + n].  Chunk-outer instead of row-outer order: 16 accumulators instead of 2 x 16 beside the MFMA accumulators
// (row-outer spills in the 256-row tiles).  launch_gemm sends a launch here only when every in-range lane would take tile_epilogue's fp16 vector
// path: whole tiles, 8-aligned leading dimensions, N % 16 == 0, no GEGLU / fp32 output / fp32 residual / split-K / sub-pixel scatter.
template <int MT, int NT, int WTM, int WTN>
__device__ __forceinline__ void tile_epilogue_stats(const GemmP& p, f32x4 (&acc)[MT][NT], int m0, int n0, int wm, int wn, int lane, int stat_blk) {
  constexpr int WID = 4 * NT;
  static_assert(WID % 8 == 0, "statistics epilogue: 8-column chunks");
  static_assert(WID <= 16, "statistics epilogue: a lane's WID-column run is stored whole or not at all - launch_gemm only guarantees N % 16 == 0");
  const int l15 = lane & 15, g = lane >> 4;
  const int nb = n0 + wn * WTN + g * WID;
  const bool full = (nb + WID <= p.N);
  // MODE 0: every run-time variant (second residual, activation, scale factors); 1 / 2 (round 6): the common case - c0 == 1, no activation, no second residual -
  // with (1) / without (2) the fp16 residual (c1 == 1), decided once per tile instead of per row block and chunk; the same arithmetic, bit-identical outputs
  auto body = [&](auto MODE_) {
  constexpr int MODE = decltype(MODE_)::value;
#pragma unroll
  for (int e = 0; e < WID; e += 8) {
    float bv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) bv[q] = 0.f;
    if (full) {
      if (p.bias) { const f16x8 b = *(const f16x8*)(p.bias + nb + e);
#pragma unroll
        for (int q = 0; q < 8; ++q) bv[q] += (float)b[q]; }
      if (p.bias2) { const f16x8 b = *(const f16x8*)(p.bias2 + nb + e);
#pragma unroll
        for (int q = 0; q < 8; ++q) bv[q] += (float)b[q]; }
    }
    float ssum[8], ssq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { ssum[q] = 0.f; ssq[q] = 0.f; }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      int m = m0 + wm * WTM + i * 16 + l15;
      if (p.halo_tw) { const int r = wm * WTM + i * 16 + l15; m = m0 + (r >> p.halo_lg) * p.Wo + (r & (p.halo_tw - 1)); }
      float o[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if constexpr (MODE == 0) o[q] = p.c0 * (acc[i][(e + q) >> 2][(e + q) & 3] + bv[q]);
        else o[q] = acc[i][(e + q) >> 2][(e + q) & 3] + bv[q];
        acc[i][(e + q) >> 2][(e + q) & 3] = 0.f;
      }
      if (!full) continue;
      if (MODE == 1 || (MODE == 0 && p.R1)) {
        const f16x8 r = *(const f16x8*)(p.R1 + (long)m * p.ldr1 + nb + e);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] += p.c1 * (float)r[q];
      }
      if constexpr (MODE == 0) {
        if (p.R2) {
          const f16x8 r = *(const f16x8*)(p.R2 + (long)m * p.ldr2 + nb + e);
#pragma unroll
          for (int q = 0; q < 8; ++q) o[q] += p.c2 * (float)r[q];
        }
        if (p.act == UG_ACT_SILU) {
#pragma unroll
          for (int q = 0; q < 8; ++q) o[q] = silu_f(o[q]);
        } else if (p.act == UG_ACT_GELU) {
#pragma unroll
          for (int q = 0; q < 8; ++q) o[q] = gelu_f(o[q]);
        }
      }
      f16x8 h;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if constexpr (MODE != 0) asm("" : "+v"(o[q]));   // (no fusing of the residual add with the fp16 rounding: tile_epilogue_lean)
        h[q] = (f16)o[q];
      }
      *(f16x8*)((f16*)p.Out + (long)m * p.ldo + nb + e) = h;
#pragma unroll
      for (int q = 0; q < 8; ++q) { const float f = (float)h[q]; ssum[q] += f; ssq[q] += f * f; }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) { ssum[q] = row16_sum(ssum[q]); ssq[q] = row16_sum(ssq[q]); }
    if (l15 == 0 && full) {
      float2* dst = p.stat_part + (long)stat_blk * p.N + nb + e;
#pragma unroll
      for (int q = 0; q < 8; q += 2) *(f32x4*)(dst + q) = (f32x4){ssum[q], ssq[q], ssum[q + 1], ssq[q + 1]};
    }
  }
  };
  const bool lean = !p.R2 && p.act == UG_ACT_NONE && p.c0 == 1.0f && (!p.R1 || p.c1 == 1.0f) && !(p.tune_knobs & 8388608);
  if (!lean) body(std::integral_constant<int, 0>{});
  else if (p.R1) body(std::integral_constant<int, 1>{});
  else body(std::integral_constant<int, 2>{});
}
