// GroupNorm (+SiLU) and LayerNorm for channels-last fp16 activations on gfx950.
// Both are HBM-bound: 16-byte vector loads/stores, fp32 statistics, wave64 reductions.
//
// GroupNorm is three launches:
//   1. gn_stats    grid (nchunk, T): every thread owns one (or a few) 8-channel vectors and walks
//                  the rows of its chunk accumulating per-channel sum / sum-of-squares in
//                  registers; a deterministic LDS tree reduces them to per-group partials.
//   2. gn_finalize grid (T): fp64 combine of the chunk partials (per frame, or pooled over all
//                  frames for the temporal res-blocks) -> per-(frame, channel) scale/shift
//                  a = rstd*gamma, b = beta - mean*rstd*gamma.
//   3. gn_apply    same decomposition as 1: y = silu(x*a + b), scale/shift held in registers.
// The input may be the virtual channel-concat of two tensors (UNet up-block skip connections)
// - group boundaries straddle the two sources there, so the concat cannot be factored out.
#include "../common.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#define GN_THREADS 256
#define GN_MAXV 3  // vectors per thread => C <= 3*256*8

// SiLU with the hardware reciprocal (1 ulp) instead of the correctly rounded fp32 division hipcc emits for `/` (14 instructions per element, which made the apply
// passes VALU-bound: round 6).  Every GroupNorm form and the GEMM epilogue share this definition (gemm_common.h carries the same one).
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

struct GnGeom { int nvec, tpr, rpi, vpt; };
// nvec 8-channel vectors per row; tpr threads cooperate on a row, rpi rows per iteration,
// vpt vectors per thread.
static __host__ __device__ inline GnGeom gn_geom(int C) {
  GnGeom g;
  g.nvec = C / 8;
  if (g.nvec <= GN_THREADS) { g.tpr = g.nvec; g.rpi = GN_THREADS / g.nvec; g.vpt = 1; }
  else { g.tpr = GN_THREADS; g.rpi = 1; g.vpt = (g.nvec + GN_THREADS - 1) / GN_THREADS; }
  return g;
}

__device__ __forceinline__ f16x8 gn_load(const GroupNormP& p, long m, int c) {
  return (c < p.C0) ? *(const f16x8*)(p.X0 + m * p.C0 + c) : *(const f16x8*)(p.X1 + m * p.C1 + (c - p.C0));
}

// phase 1 of GroupNorm for workgroup (chunk, t): per-group sum / sum-of-squares partials of the chunk's rows -> p.ws
__device__ __forceinline__ void gn_stats_body(const GroupNormP& p, int t, int chunk, int nchunk, int rows_per_chunk, float* red) {
  const int C = p.C0 + p.C1;
  const GnGeom gg = gn_geom(C);
  const int tid = threadIdx.x;
  const int rsub = tid / gg.tpr, v0 = tid - rsub * gg.tpr;
  const bool active = rsub < gg.rpi;
  const int r0 = chunk * rows_per_chunk;
  const int r1 = min(r0 + rows_per_chunk, p.HW);
  float s[GN_MAXV][8], q[GN_MAXV][8];
#pragma unroll
  for (int k = 0; k < GN_MAXV; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[k][e] = 0.f; q[k][e] = 0.f; }
  if (active) {
    if (gg.vpt == 1) {   // common case: one vector per thread -> keep 4 rows' loads in flight
      const int c = v0 * 8;
      int r = r0 + rsub;
      for (; r + 3 * gg.rpi < r1; r += 4 * gg.rpi) {
        f16x8 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = gn_load(p, (long)t * p.HW + r + u * gg.rpi, c);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float f = (float)x[u][e]; s[0][e] += f; q[0][e] += f * f; }
      }
      for (; r < r1; r += gg.rpi) {
        const f16x8 x = gn_load(p, (long)t * p.HW + r, c);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = (float)x[e]; s[0][e] += f; q[0][e] += f * f; }
      }
    } else
    for (int r = r0 + rsub; r < r1; r += gg.rpi) {
      const long m = (long)t * p.HW + r;
#pragma unroll
      for (int k = 0; k < GN_MAXV; ++k) {
        const int v = v0 + k * GN_THREADS;
        if (k < gg.vpt && v < gg.nvec) {
          const f16x8 x = gn_load(p, m, v * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float f = (float)x[e]; s[k][e] += f; q[k][e] += f * f; }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < GN_MAXV; ++k) {
      const int v = v0 + k * GN_THREADS;
      if (k < gg.vpt && v < gg.nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          red[((long)rsub * C + v * 8 + e) * 2 + 0] = s[k][e];
          red[((long)rsub * C + v * 8 + e) * 2 + 1] = q[k][e];
        }
      }
    }
  }
  __syncthreads();
  const int cpg = C / p.G;
  for (int g = tid; g < p.G; g += GN_THREADS) {
    float a = 0.f, b = 0.f;
    for (int rs = 0; rs < gg.rpi; ++rs)
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        a += red[((long)rs * C + c) * 2 + 0];
        b += red[((long)rs * C + c) * 2 + 1];
      }
    float* dst = p.ws + (((long)t * nchunk + chunk) * p.G + g) * 2;
    dst[0] = a; dst[1] = b;
  }
}

__global__ __launch_bounds__(GN_THREADS) void gn_stats(const GroupNormP p, int nchunk, int rows_per_chunk) {
  extern __shared__ float red[];  // [rpi][C][2]
  gn_stats_body(p, blockIdx.y, blockIdx.x, nchunk, rows_per_chunk, red);
}

__global__ __launch_bounds__(1024) void gn_finalize(const GroupNormP p, int nchunk, float* ab) {
  // grid (T), NT = 256 or 1024 threads: NT/G threads cooperate on one group's chunk partials (fp64, fixed order)
  __shared__ double sa[1024], sb[1024];
  __shared__ float mr[2 * 256];
  const int C = p.C0 + p.C1, cpg = C / p.G, G = p.G;
  const int t = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
  const int SUB = NT / G;
  const int g = tid % G, sub = tid / G;
  const int tlo = p.temporal ? 0 : t, thi = p.temporal ? p.T : t + 1;
  const int nitem = (thi - tlo) * nchunk;
  double a = 0.0, b = 0.0;
  if (sub < SUB) {
    // partial (tt, ch) lives at ((tt*nchunk + ch)*G + g)*2 and (tt, ch) pairs are contiguous in `it`
    const float2* src = (const float2*)p.ws + (long)tlo * nchunk * G + g;
    int it = sub;
    for (; it + 3 * SUB < nitem; it += 4 * SUB) {      // 4 independent loads in flight, fixed summation order
      const float2 v0 = src[(long)it * G], v1 = src[(long)(it + SUB) * G];
      const float2 v2 = src[(long)(it + 2 * SUB) * G], v3 = src[(long)(it + 3 * SUB) * G];
      a += (double)v0.x; b += (double)v0.y; a += (double)v1.x; b += (double)v1.y;
      a += (double)v2.x; b += (double)v2.y; a += (double)v3.x; b += (double)v3.y;
    }
    for (; it < nitem; it += SUB) { const float2 v = src[(long)it * G]; a += (double)v.x; b += (double)v.y; }
  }
  sa[tid] = a; sb[tid] = b;
  __syncthreads();
  if (tid < G) {
    for (int s2 = 1; s2 < SUB; ++s2) { a += sa[tid + s2 * G]; b += sb[tid + s2 * G]; }
    const double n = (double)cpg * p.HW * (p.temporal ? p.T : 1);
    const double mean = a / n;
    double var = b / n - mean * mean;
    if (var < 0.0) var = 0.0;
    mr[2 * tid] = (float)mean;
    mr[2 * tid + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
  }
  __syncthreads();
  for (int c = tid; c < C; c += NT) {
    const int gg = c / cpg;
    const float ga = p.gamma ? (float)p.gamma[c] : 1.f, be = p.beta ? (float)p.beta[c] : 0.f;
    const float sc = mr[2 * gg + 1] * ga;
    ab[((long)t * C + c) * 2 + 0] = sc;
    ab[((long)t * C + c) * 2 + 1] = be - mr[2 * gg] * sc;
  }
}

// phase 2 when the statistics come from the producing GEMM's epilogue (GroupNormP::part: per block of part_rb rows and per channel the sum / sum of
// squares of the fp16 values stored): grid (G, temporal ? 1 : T) - workgroup (g, t) combines the blocks of its frame (all frames: pooled) x the
// channels of its group in fp64, fixed order, and writes the scale / shift of the group's channels (pooled: for every frame).
// FRAME_PART (pooled statistics over many blocks - the VAE decoder's 196608-pixel frames): the workgroup of frame t only writes its (sum, sum of squares)
// to scratch[t * G + g]; gn_finalize_pool combines the T entries.  One stage: 32 workgroups x 98 K items = 72 us; two: 12 + 5.
template <bool FRAME_PART>
__global__ __launch_bounds__(1024) void gn_finalize_cols(const GroupNormP p, float* ab, double2* scratch) {
  __shared__ double sa[1024], sb[1024];
  const int C = p.C0, cpg = C / p.G, g = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
  const int S = p.HW / p.part_rb;                              // blocks per frame
  const bool pooled = p.temporal && !FRAME_PART;
  const int tlo = pooled ? 0 : blockIdx.y, thi = pooled ? p.T : blockIdx.y + 1;
  const int nitem = (thi - tlo) * S * cpg;
  const float2* src = p.part + (long)tlo * S * C + g * cpg;
  double a = 0.0, b = 0.0;
  auto at = [&](int it) { const int bs = it / cpg, cc = it - bs * cpg; return src[(long)bs * C + cc]; };
  int it = tid;
  for (; it + 3 * NT < nitem; it += 4 * NT) {                  // 4 independent loads in flight, fixed summation order
    const float2 v0 = at(it), v1 = at(it + NT), v2 = at(it + 2 * NT), v3 = at(it + 3 * NT);
    a += (double)v0.x; b += (double)v0.y; a += (double)v1.x; b += (double)v1.y;
    a += (double)v2.x; b += (double)v2.y; a += (double)v3.x; b += (double)v3.y;
  }
  for (; it < nitem; it += NT) { const float2 v = at(it); a += (double)v.x; b += (double)v.y; }
  sa[tid] = a; sb[tid] = b;
  __syncthreads();
  for (int st = NT / 2; st > 0; st >>= 1) {
    if (tid < st) { sa[tid] += sa[tid + st]; sb[tid] += sb[tid + st]; }
    __syncthreads();
  }
  if (FRAME_PART) {
    if (tid == 0) scratch[(long)blockIdx.y * p.G + g] = make_double2(sa[0], sb[0]);
    return;
  }
  const double n = (double)cpg * p.HW * (thi - tlo);
  const double mean = sa[0] / n;
  double var = sb[0] / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + (double)p.eps));
  for (int i = tid; i < (thi - tlo) * cpg; i += NT) {
    const int t = tlo + i / cpg, c = g * cpg + i % cpg;
    const float ga = p.gamma ? (float)p.gamma[c] : 1.f, be = p.beta ? (float)p.beta[c] : 0.f;
    const float sc = rf * ga;
    ab[((long)t * C + c) * 2 + 0] = sc;
    ab[((long)t * C + c) * 2 + 1] = be - mf * sc;
  }
}
// second stage of the pooled form: grid (G), one wave - frame sums in fixed order, scale / shift of the group's channels for every frame
__global__ __launch_bounds__(64) void gn_finalize_pool(const GroupNormP p, const double2* scratch, float* ab) {
  const int C = p.C0, cpg = C / p.G, g = blockIdx.x, tid = threadIdx.x;
  double a = 0.0, b = 0.0;
  for (int t = 0; t < p.T; ++t) { const double2 v = scratch[(long)t * p.G + g]; a += v.x; b += v.y; }
  const double n = (double)cpg * p.HW * p.T;
  const double mean = a / n;
  double var = b / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + (double)p.eps));
  for (int i = tid; i < p.T * cpg; i += 64) {
    const int t = i / cpg, c = g * cpg + i % cpg;
    const float ga = p.gamma ? (float)p.gamma[c] : 1.f, be = p.beta ? (float)p.beta[c] : 0.f;
    const float sc = rf * ga;
    ab[((long)t * C + c) * 2 + 0] = sc;
    ab[((long)t * C + c) * 2 + 1] = be - mf * sc;
  }
}

// phase 3 for workgroup (chunk, t): y = silu(x * a[c] + b[c]); coef(c, a, b) supplies the per-channel scale / shift
template <bool SILU, typename F>
__device__ __forceinline__ void gn_apply_body(const GroupNormP& p, int t, int chunk, int rows_per_chunk, F coef) {
  const int C = p.C0 + p.C1;
  const GnGeom gg = gn_geom(C);
  const int tid = threadIdx.x;
  const int rsub = tid / gg.tpr, v0 = tid - rsub * gg.tpr;
  if (rsub >= gg.rpi) return;
  const int r0 = chunk * rows_per_chunk;
  const int r1 = min(r0 + rows_per_chunk, p.HW);
  float a[GN_MAXV][8], b[GN_MAXV][8];
#pragma unroll
  for (int k = 0; k < GN_MAXV; ++k) {
    const int v = v0 + k * GN_THREADS;
    if (k < gg.vpt && v < gg.nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        coef(v * 8 + e, a[k][e], b[k][e]);
      }
    }
  }
  if (gg.vpt == 1) {
    const int c = v0 * 8;
    int r = r0 + rsub;
    for (; r + 3 * gg.rpi < r1; r += 4 * gg.rpi) {
      f16x8 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = gn_load(p, (long)t * p.HW + r + u * gg.rpi, c);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f16x8 y;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float f = (float)x[u][e] * a[0][e] + b[0][e];
          if (SILU) f = silu_f(f);
          y[e] = (f16)f;
        }
        *(f16x8*)(p.Y + ((long)t * p.HW + r + u * gg.rpi) * C + c) = y;
      }
    }
    for (; r < r1; r += gg.rpi) {
      const f16x8 x = gn_load(p, (long)t * p.HW + r, c);
      f16x8 y;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = (float)x[e] * a[0][e] + b[0][e];
        if (SILU) f = silu_f(f);
        y[e] = (f16)f;
      }
      *(f16x8*)(p.Y + ((long)t * p.HW + r) * C + c) = y;
    }
    return;
  }
  for (int r = r0 + rsub; r < r1; r += gg.rpi) {
    const long m = (long)t * p.HW + r;
#pragma unroll
    for (int k = 0; k < GN_MAXV; ++k) {
      const int v = v0 + k * GN_THREADS;
      if (k < gg.vpt && v < gg.nvec) {
        const f16x8 x = gn_load(p, m, v * 8);
        f16x8 y;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float f = (float)x[e] * a[k][e] + b[k][e];
          if (SILU) f = silu_f(f);
          y[e] = (f16)f;
        }
        *(f16x8*)(p.Y + m * C + v * 8) = y;
      }
    }
  }
}

template <bool SILU>   // compile time: a run-time flag is a branch per ELEMENT in the unrolled loops
__global__ __launch_bounds__(GN_THREADS) void gn_apply(const GroupNormP p, int rows_per_chunk, const float* ab) {
  const int C = p.C0 + p.C1, t = blockIdx.y;
  gn_apply_body<SILU>(p, t, blockIdx.x, rows_per_chunk, [&](int c, float& a, float& b) { a = ab[((long)t * C + c) * 2 + 0]; b = ab[((long)t * C + c) * 2 + 1]; });
}
static void launch_gn_apply(const GroupNormP& p, int nchunk, int rpc, const float* ab, hipStream_t s) {
  if (p.silu) hipLaunchKernelGGL(gn_apply<true>, dim3(nchunk, p.T), dim3(GN_THREADS), 0, s, p, rpc, ab);
  else hipLaunchKernelGGL(gn_apply<false>, dim3(nchunk, p.T), dim3(GN_THREADS), 0, s, p, rpc, ab);
}

// rows per chunk: enough workgroups (T * nchunk >= ~1024) to fill 256 CUs several times over, but at
// least 4 row-iterations per thread so the unrolled loads stay in flight; at most 128 chunks per frame.
// ------------------------------------------------------------------------------------------
// Small tensors (the low-resolution UNet levels): ONE launch.  A workgroup owns one group of one frame
// (or of all frames for the temporal variant): pass 1 accumulates sum / sum-of-squares over its slab
// (fixed-order tree => deterministic), pass 2 re-reads the slab (L2 resident) and applies.
// 8-byte vectors: needs channels-per-group % 4 == 0 and C0 % 4 == 0.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_small(const GroupNormP p) {
  __shared__ float rs[256], rq[256];
  __shared__ float mr[2];
  const int C = p.C0 + p.C1, cpg = C / p.G, vpg = cpg / 4;
  const int g = blockIdx.x, tid = threadIdx.x;
  const long row0 = p.temporal ? 0 : (long)blockIdx.y * p.HW;
  const long rows = p.temporal ? (long)p.T * p.HW : p.HW;
  const long nitem = rows * vpg;
  auto ld = [&](long m, int c) -> f16x4 {
    return (c < p.C0) ? *(const f16x4*)(p.X0 + m * p.C0 + c) : *(const f16x4*)(p.X1 + m * p.C1 + (c - p.C0));
  };
  float s = 0.f, q = 0.f;
  for (long it = tid; it < nitem; it += 256) {
    const long r = it / vpg; const int v = (int)(it - r * vpg);
    const f16x4 x = ld(row0 + r, g * cpg + v * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float f = (float)x[e]; s += f; q += f * f; }
  }
  rs[tid] = s; rq[tid] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { rs[tid] += rs[tid + o]; rq[tid] += rq[tid + o]; }
    __syncthreads();
  }
  if (tid == 0) {
    const double n = (double)rows * cpg;
    const double mean = (double)rs[0] / n;
    double var = (double)rq[0] / n - mean * mean;
    if (var < 0.0) var = 0.0;
    mr[0] = (float)mean; mr[1] = (float)(1.0 / sqrt(var + (double)p.eps));
  }
  __syncthreads();
  const float mean = mr[0], rstd = mr[1];
  for (long it = tid; it < nitem; it += 256) {
    const long r = it / vpg; const int v = (int)(it - r * vpg);
    const int c = g * cpg + v * 4;
    const f16x4 x = ld(row0 + r, c);
    const f16x4 ga = *(const f16x4*)(p.gamma + c), be = *(const f16x4*)(p.beta + c);
    f16x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = rstd * (float)ga[e];
      const float f = (float)x[e] * a + ((float)be[e] - mean * a);
      y[e] = (f16)(p.silu ? silu_f(f) : f);
    }
    *(f16x4*)(p.Y + (row0 + r) * C + c) = y;
  }
}

// ------------------------------------------------------------------------------------------
// Round 3, slab form: ONE launch, ONE read, no hand-off.  A workgroup owns one group of one frame (or of all frames: temporal variant) and
// keeps its whole slab in REGISTERS: <= VMAX 4-channel vectors per thread, all loads issued before the first use (one memory round trip
// instead of one per loop iteration), statistics as an exact two-pass over the registers (mean, then sum of squared deviations; wave
// butterfly + fixed-order combine of the wave partials => deterministic), normalise + SiLU from the registers.  Thread (row r, vector v):
// vpg = cpg / 4 neighbouring lanes read one row's cpg channels, NT / vpg rows per step.  Replaces gn_small's two strided passes on the
// low-resolution levels (12 MB tensors: 19.5 -> ~8 us) and the three launches of the temporal (pooled) variant there (20 - 29 us).
// ------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ float gn_block_sum(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();                       // red is reused by the second reduction
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) t += red[w];
  return t;
}

// MODE 0: self-contained (statistics of the workgroup's own slab).  MODE 1 / 2 (round 4): the pooled (temporal) GroupNorm of the low-resolution
// levels as TWO launches of per-frame slabs instead of stats / finalize / apply: MODE 1 writes (mean, sum of squared deviations) of slab
// (group, frame) to p.ws; MODE 2 re-reads its slab (L2 / Infinity-Cache resident: <= 12 MB tensors), combines the T frame partials of its
// group (Chan's formula in fp64, fixed order, the same in every workgroup) while the loads are in flight, and applies.
template <int NT, int VMAX, int MODE>
__global__ __launch_bounds__(NT) void gn_slab(const GroupNormP p) {
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  __shared__ float red[NT / 64];
  const int C = p.C0 + p.C1, cpg = C / p.G, vpg = cpg / 4;
  const int g = blockIdx.x, tid = threadIdx.x;
  const bool pooled = MODE == 0 && p.temporal;
  const long row0 = pooled ? 0 : (long)blockIdx.y * p.HW;
  const int R = pooled ? p.T * p.HW : p.HW;
  const int rpi = NT / vpg;
  const int rsub = tid / vpg, v = tid - rsub * vpg;
  const bool act = rsub < rpi;
  const int c = g * cpg + v * 4;
  // buffer addressing: rows >= R (and idle lanes) land past num_records -> loads return zeros, stores are dropped.  The row step is added to
  // the VECTOR offset: the range check covers voffset + the instruction offset, not the scalar offset operand.
  const bool s0 = g * cpg < p.C0;                          // the group lies in ONE source (C0 % cpg == 0, checked by gn_slab_plan)
  const int ld = s0 ? p.C0 : p.C1;
  const f16* base = (s0 ? p.X0 : p.X1) + row0 * ld;
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, R * ld * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Y + row0 * C), 0, R * C * 2, 0x00020000);
  const unsigned OOB = 0x7fffffffu;
  const unsigned voff = act ? (unsigned)(rsub * ld + (s0 ? c : c - p.C0)) * 2u : OOB;
  const unsigned yoff = act ? (unsigned)(rsub * C + c) * 2u : OOB;
  const unsigned stepx = rpi * ld * 2, stepy = rpi * C * 2;
  f16x4 x[VMAX];
#pragma unroll
  for (int k = 0; k < VMAX; ++k)
    x[k] = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(rX, act ? (int)(voff + k * stepx) : (int)OOB, 0, 0));
  const f16x4 ga = *(const f16x4*)(p.gamma + (act ? c : 0)), be = *(const f16x4*)(p.beta + (act ? c : 0));   // with the slab loads, not behind the reductions (MODE 1 never uses them)
  float mean, rstd;
  const double n = (double)R * cpg;
  if (MODE == 2) {
    // pooled statistics of group g from the T per-frame (mean, M2) partials; uniform addresses, every thread the same arithmetic
    // (round 6: lane t of every wave loads frame t's partial - one memory round trip instead of T dependent scalar loads - and the sums run over the lanes in
    // frame order, the same order and the same arithmetic as before; T <= 64 is checked by the launcher, else the serial form)
    const float2* part = (const float2*)p.ws + g;
    double msum = 0.0, m2 = 0.0, pm;
    if (p.T <= 64) {
      const int ln = tid & 63;
      const float2 mine = part[(long)min(ln, p.T - 1) * p.G];
      auto lane_f = [](float v, int t) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), t)); };   // uniform t: v_readlane
      for (int t = 0; t < p.T; ++t) msum += (double)lane_f(mine.x, t);
      pm = msum / p.T;
      for (int t = 0; t < p.T; ++t) { const double d = (double)lane_f(mine.x, t) - pm; m2 += (double)lane_f(mine.y, t) + n * d * d; }
    } else {
      for (int t = 0; t < p.T; ++t) msum += (double)part[(long)t * p.G].x;
      pm = msum / p.T;
      for (int t = 0; t < p.T; ++t) { const float2 v = part[(long)t * p.G]; const double d = (double)v.x - pm; m2 += (double)v.y + n * d * d; }
    }
    mean = (float)pm;
    rstd = (float)(1.0 / sqrt(m2 / (n * p.T) + (double)p.eps));
  } else {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < VMAX; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += (float)x[k][e];
    mean = (float)((double)gn_block_sum<NT>(s, red) / n);
    // keep the slab PACKED between the passes (the compiler would otherwise hold the fp32 images: twice the registers)
#pragma unroll
    for (int k = 0; k < VMAX; ++k) asm volatile("" : "+v"(x[k]));
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < VMAX; ++k)
      if (act && rsub + k * rpi < R) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = (float)x[k][e] - mean; q += d * d; }
      }
    const float qs = gn_block_sum<NT>(q, red);
    if (MODE == 1) {
      if (tid == 0) ((float2*)p.ws)[(long)blockIdx.y * p.G + g] = make_float2(mean, qs);
      return;
    }
    rstd = (float)(1.0 / sqrt((double)qs / n + (double)p.eps));
#pragma unroll
    for (int k = 0; k < VMAX; ++k) asm volatile("" : "+v"(x[k]));
  }
  float a[4], b[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { a[e] = rstd * (float)ga[e]; b[e] = (float)be[e] - mean * a[e]; }
  auto out = [&](auto SILU) {      // the flag is uniform: one branch per launch, not one per element of the unrolled loop
#pragma unroll
    for (int k = 0; k < VMAX; ++k) {
      f16x4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float f = (float)x[k][e] * a[e] + b[e];
        if (decltype(SILU)::value) f = silu_f(f);
        y[e] = (f16)f;
      }
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, y), rY, act ? (int)(yoff + k * stepy) : (int)OOB, 0, 0);   // rows >= R: dropped
    }
  };
  if (p.silu) out(std::true_type{}); else out(std::false_type{});
}

template <int MODE>
static void gn_slab_launch(const GroupNormP& p, int snt, int svmax, dim3 grid, hipStream_t s) {
  if (snt == 256) {
    if (svmax <= 4) hipLaunchKernelGGL((gn_slab<256, 4, MODE>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gn_slab<256, 8, MODE>), grid, dim3(256), 0, s, p);
  } else {
    switch (svmax) {
      case 4: hipLaunchKernelGGL((gn_slab<1024, 4, MODE>), grid, dim3(1024), 0, s, p); break;
      case 8: hipLaunchKernelGGL((gn_slab<1024, 8, MODE>), grid, dim3(1024), 0, s, p); break;
      case 16: hipLaunchKernelGGL((gn_slab<1024, 16, MODE>), grid, dim3(1024), 0, s, p); break;
      case 32: hipLaunchKernelGGL((gn_slab<1024, 32, MODE>), grid, dim3(1024), 0, s, p); break;
      default: hipLaunchKernelGGL((gn_slab<1024, 48, MODE>), grid, dim3(1024), 0, s, p); break;
    }
  }
}

// slab-form plan: threads per workgroup and register vectors per thread, 0 = does not fit
static inline int gn_slab_plan(const GroupNormP& p, int& vmax, bool per_frame = false) {
  const int C = p.C0 + p.C1, cpg = C / p.G;
  const bool pooled = p.temporal && !per_frame;
  if (cpg % 4 || p.C0 % cpg || !p.gamma || !p.beta || cpg / 4 > 64) return 0;
  if ((pooled ? (long)p.T * p.HW : (long)p.HW) * C * 2 >= (1L << 31)) return 0;
  const int vpg = cpg / 4;
  const long R = pooled ? (long)p.T * p.HW : p.HW;
  for (int nt : {256, 1024}) {
    const long k = (R + nt / vpg - 1) / (nt / vpg);
    const int cap = nt == 256 ? 8 : 48;
    if (k <= cap) { vmax = k <= 4 ? 4 : k <= 8 ? 8 : k <= 16 ? 16 : k <= 32 ? 32 : 48; return nt; }
  }
  return 0;
}

static inline void gn_chunks2(int T, int HW, int C, int& nchunk, int& rpc) {
  const GnGeom gg = gn_geom(C);
  static const int total_want = getenv("UG_GN_WANT") ? atoi(getenv("UG_GN_WANT")) : 1024;   // A/B aid
  const int want = cdiv(total_want, T);                 // chunks per frame we would like
  rpc = cdiv(HW, want);
  const int min_rpc = 4 * gg.rpi;
  if (rpc < min_rpc) rpc = min_rpc;
  nchunk = cdiv(HW, rpc);
  if (nchunk > 128) { nchunk = 128; rpc = cdiv(HW, nchunk); nchunk = cdiv(HW, rpc); }
}

size_t groupnorm_ws_floats(int T, int HW, int C, int G) {
  int nchunk, rpc;
  gn_chunks2(T, HW, C, nchunk, rpc);
  // three-launch scheme: chunk partials + per-(frame, channel) scale / shift (the per-frame slab scheme needs T x G x 2 of them)
  return (size_t)T * nchunk * G * 2 + (size_t)T * C * 2;
}

// Launch scheme (p.mode 0 = pick, 1 / 2 / 4 / 6 force; tools/bench_groupnorm.py, profiles/r01_groupnorm_variants.txt):
//   4: gn_slab, one workgroup per (group, frame) with the slab in registers (round 3; replaces 2 wherever it fits)
//   6: pooled statistics from per-frame slabs, two launches (round 4).  Round 4 also re-measured gn_apply combining the chunk partials itself (no
//      gn_finalize launch, first loads issued before the combine): 41.2 vs 40.3 us isolated, +3 us on the one-frame StableNormal tensors, +-0 in the clip - removed.
//   1: gn_stats / gn_finalize / gn_apply   2: gn_small, one workgroup per (group, frame) - wins on the low-resolution levels,
//      loses when a row contributes < 64 B to a group and there are many rows (T25 x HW768 x C640: 47 vs 28 us).
// Also measured and removed: apply with an in-block finalize (2 launches; +3.5 us of serial latency per workgroup, no
// gain) and a single launch with a device-scope barrier (agent-scope release/acquire flushes and invalidates the whole
// L2: 162 ms of GroupNorm per clip; with the partials as coherent atomics instead: 140 ms; three launches: 113 ms).
// launch scheme launch_groupnorm takes for p (p.mode == 0: the measured rules below); *svmax_o / *tvmax_o / *snt_o / *tnt_o = the slab plans
static int gn_pick_mode(const GroupNormP& p, int* snt_o = nullptr, int* svmax_o = nullptr, int* tnt_o = nullptr, int* tvmax_o = nullptr) {
  const int C = p.C0 + p.C1;
  const int cpg = C / p.G;
  const long slab = (long)p.HW * cpg * (p.temporal ? p.T : 1);
  const bool small_ok = !p.temporal && cpg % 4 == 0 && p.C0 % 4 == 0 && p.gamma && p.beta;
  int mode = p.mode;
  int svmax = 0, tvmax = 0;
  const int snt = gn_slab_plan(p, svmax);
  const int tnt = p.temporal ? gn_slab_plan(p, tvmax, true) : 0;   // per-frame slabs of the pooled variant (mode 6)
  if (mode == 0) {
    // tools/bench_groupnorm.py, profiles/r03_groupnorm_slab.txt: the register-slab form wins wherever gn_small did (T25 x HW192 x C1280 12.2 vs
    // 16.5 us, x C2560 19.7 vs 27.4, T1 x HW256 x C1280 5.9 vs 10.9) and up to 64 K elements per slab when a row gives a group >= 64 B and
    // the grid fills the chip (T25 x HW768 x C1280 38.7 vs 41.4); pooled (temporal) slabs have only G workgroups: two launches of per-frame
    // slabs on the two low-resolution levels (mode 6; profiles/r04_groupnorm_modes.txt: 22.4 vs 23.8 us, 15.2 vs 16.0; the 768-row level loses 53 vs 30)
    const bool narrow_ok = p.HW <= 256 || cpg >= 32;
    static const bool noslab = getenv("UG_GN_NOSLAB") != nullptr;   // A/B aids
    static const bool not2 = getenv("UG_GN_NOT2") != nullptr;
    if (small_ok && slab <= 16384 && narrow_ok) mode = (snt && !noslab) ? 4 : 2;
    else if (!p.temporal && snt && svmax <= 16 && narrow_ok && p.G * p.T >= 256 && !noslab) mode = 4;
    else if (p.temporal && tnt && tvmax <= 16 && p.HW <= 256 && p.G * p.T >= 256 && !not2) mode = 6;
    else mode = 1;
  }
  if (mode == 4 && !snt) mode = p.temporal ? 1 : 2;
  if (mode == 2 && !small_ok) mode = 1;
  if (mode == 3) mode = 1;   // (3 was round 3's one-launch ticket scheme: slower than three launches everywhere, removed in round 4)
  if (mode == 6 && !tnt) mode = 1;
  if (snt_o) *snt_o = snt;
  if (svmax_o) *svmax_o = svmax;
  if (tnt_o) *tnt_o = tnt;
  if (tvmax_o) *tvmax_o = tvmax;
  return mode;
}
// true when launch_groupnorm would take the statistics from GroupNormP::part (the producing GEMM's epilogue) for this tensor - the same rule the launcher
// applies, so that a caller (engine.hip: stat_alloc, the op test) never makes a producer write partial sums that the slab / small forms then ignore
bool groupnorm_uses_part(const GroupNormP& p0, int part_rb) {
  GroupNormP p = p0;
  if ((p.C0 + p.C1) % p.G != 0 || p.C1 != 0 || p.mode != 0 || part_rb <= 0 || p.HW % part_rb != 0) return false;
  const int mode = gn_pick_mode(p);
  return mode == 1 || mode == 6;
}

bool launch_groupnorm(const GroupNormP& p, hipStream_t s) {
  const int C = p.C0 + p.C1;
  UG_REQUIRE(p.C0 % 8 == 0 && p.C1 % 8 == 0, "GroupNorm channels must be multiples of 8");
  UG_REQUIRE(C % p.G == 0 && p.G <= 256 && 256 % p.G == 0, "GroupNorm group count must divide 256");
  UG_REQUIRE(C <= GN_MAXV * GN_THREADS * 8, "GroupNorm too many channels");
  const int cpg = C / p.G;
  int snt = 0, svmax = 0, tnt = 0, tvmax = 0;
  int mode = gn_pick_mode(p, &snt, &svmax, &tnt, &tvmax);
  if (p.part && p.mode == 0 && (mode == 1 || mode == 6)) {
    // the producing GEMM's epilogue left per-block column sums (GemmP::stat_part): no statistics pass over X - combine them, apply
    UG_REQUIRE(p.C1 == 0 && p.part_rb > 0 && p.HW % p.part_rb == 0, "GroupNorm: epilogue statistics need a single source and whole blocks per frame");
    int nchunk, rpc;
    gn_chunks2(p.T, p.HW, C, nchunk, rpc);
    float* ab = p.ws + (size_t)p.T * nchunk * p.G * 2;
    const long nitem = (long)(p.temporal ? p.T : 1) * (p.HW / p.part_rb) * cpg;
    const long per_frame = (long)(p.HW / p.part_rb) * cpg;
    if (p.temporal && p.T > 1 && per_frame >= 2048 && (size_t)p.T * p.G * 4 <= (size_t)p.T * nchunk * p.G * 2) {   // pooled over many blocks: per-frame sums, then one wave per group
      double2* scratch = (double2*)p.ws;
      hipLaunchKernelGGL(gn_finalize_cols<true>, dim3(p.G, p.T), dim3(1024), 0, s, p, ab, scratch);
      hipLaunchKernelGGL(gn_finalize_pool, dim3(p.G), dim3(64), 0, s, p, (const double2*)scratch, ab);
    } else {
      hipLaunchKernelGGL(gn_finalize_cols<false>, dim3(p.G, p.temporal ? 1 : p.T), dim3(nitem > 2048 ? 1024 : 256), 0, s, p, ab, (double2*)nullptr);
    }
    launch_gn_apply(p, nchunk, rpc, (const float*)ab, s);
    UG_CHECK(hipGetLastError());
    return true;
  }
  if (mode == 4) {
    gn_slab_launch<0>(p, snt, svmax, dim3(p.G, p.temporal ? 1 : p.T), s);
  } else if (mode == 6) {
    gn_slab_launch<1>(p, tnt, tvmax, dim3(p.G, p.T), s);
    gn_slab_launch<2>(p, tnt, tvmax, dim3(p.G, p.T), s);
  } else if (mode == 2) {
    hipLaunchKernelGGL(gn_small, dim3(p.G, p.T), dim3(256), 0, s, p);
  } else {
    int nchunk, rpc;
    gn_chunks2(p.T, p.HW, C, nchunk, rpc);
    const GnGeom gg = gn_geom(C);
    const size_t lds = (size_t)gg.rpi * C * 2 * sizeof(float);
    float* ab = p.ws + (size_t)p.T * nchunk * p.G * 2;
    hipLaunchKernelGGL(gn_stats, dim3(nchunk, p.T), dim3(GN_THREADS), lds, s, p, nchunk, rpc);
    const int fin_threads = ((p.temporal ? p.T : 1) * nchunk > 64) ? 1024 : 256;
    hipLaunchKernelGGL(gn_finalize, dim3(p.T), dim3(fin_threads), 0, s, p, nchunk, ab);
    launch_gn_apply(p, nchunk, rpc, (const float*)ab, s);
  }
  UG_CHECK(hipGetLastError());
  return false;
}

// ------------------------------------------------------------------------------------------
// LayerNorm: one wave64 per row, row held in registers, exact two-pass statistics.  (Round 4 re-measured two / four rows per wave with all loads issued
// before the first use - more bytes in flight per wave: 15.6 / 17.1 us per call against 15.4, clip +-0 / +3.5 ms; removed.)
// ------------------------------------------------------------------------------------------
template <int VPL>
__global__ __launch_bounds__(256) void ln_kernel(const LayerNormP p) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const int nvec = p.C / 8;
  float x[VPL][8];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int v = lane + k * 64;
    if (v < nvec) {
      const f16x8 h = *(const f16x8*)(p.X + row * p.C + v * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[k][e] = (float)h[e];
      if (p.addvec) {
        const f16x8 a = *(const f16x8*)(p.addvec + ((row + p.row0) / p.rows_per_vec) * p.C + v * 8);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) { o[e] = (f16)(x[k][e] + (float)a[e]); x[k][e] = (float)o[e]; }
        if (p.Xout) *(f16x8*)(p.Xout + row * p.C + v * 8) = o;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += x[k][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[k][e] = 0.f;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / p.C;
  float var = 0.f;
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int v = lane + k * 64;
    if (v < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = x[k][e] - mean; var += d * d; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
  const float rstd = rsqrtf(var / p.C + p.eps);
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int v = lane + k * 64;
    if (v < nvec) {
      const f16x8 g = *(const f16x8*)(p.gamma + v * 8);
      const f16x8 b = *(const f16x8*)(p.beta + v * 8);
      f16x8 y;
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = (f16)((x[k][e] - mean) * rstd * (float)g[e] + (float)b[e]);
      if (!p.Y8) { *(f16x8*)(p.Y + row * p.C + v * 8) = y; continue; }
      // MX-fp8 output (kernels/mx8.hip semantics on the fp16-rounded values): the 4 lanes 4j..4j+3 hold one 32-element block
      float amax = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf((float)y[e]));
      amax = fmaxf(amax, __shfl_xor(amax, 1));
      amax = fmaxf(amax, __shfl_xor(amax, 2));
      int ex = 0;
      if (amax > 0.f) { (void)frexpf(amax, &ex); ex = ex - 1 - 8; }
      ex = min(max(ex, -127), 127);
      const float inv = ldexpf(1.0f, -ex);
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = fminf(fmaxf((float)y[e] * inv, -448.f), 448.f);
      int p0 = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], 0, false); p0 = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], p0, true);
      int p1 = __builtin_amdgcn_cvt_pk_fp8_f32(t[4], t[5], 0, false); p1 = __builtin_amdgcn_cvt_pk_fp8_f32(t[6], t[7], p1, true);
      *(uint2*)(p.Y8 + row * p.C + v * 8) = make_uint2((unsigned)p0, (unsigned)p1);
      if ((lane & 3) == 0)   // one byte per block: dword (K step = v / 16, row), byte (v % 16) / 4
        ((unsigned char*)(p.S8 + (long)(v >> 4) * p.ld_s8 + row))[(v & 15) >> 2] = (unsigned char)(ex + 127);
    }
  }
}

// Round 6: rows of 320 / 640 / 1280 channels (every LayerNorm of the clip) on L = C / 40 lanes each - 8 / 4 / 2 rows per wave, five 16-byte vectors per lane, all
// 64 lanes loading (ln_kernel's one-wave-per-row form has 40 of 64 lanes active at C = 320 and a half-empty second / third load at 640 / 1280), the two
// reductions over log2(L) butterfly steps inside the row's lane group.  Same two-pass statistics and the same output expression as ln_kernel; the summation
// order inside a row differs (results agree to fp32 rounding of mean / variance).  Carries ln_kernel's MX-fp8 output form (C % 128 == 0).
template <int L>
__global__ __launch_bounds__(256) void ln40_kernel(const LayerNormP p) {
  constexpr int RPW = 64 / L, NV = 5;
  const int lane = threadIdx.x & 63, sub = lane / L, l = lane % L;
  const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + sub;
  const bool ok = row < p.M;
  const int C = 40 * L;
  const f16* xr = p.X + (ok ? row : 0) * C + l * 8;
  f16x8 h[NV], gm[NV], bt[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) h[k] = *(const f16x8*)(xr + k * L * 8);
  // gamma / beta with the row loads, not behind the two reductions: one memory round trip per wave instead of two
#pragma unroll
  for (int k = 0; k < NV; ++k) { gm[k] = *(const f16x8*)(p.gamma + (l + k * L) * 8); bt[k] = *(const f16x8*)(p.beta + (l + k * L) * 8); }
  if (p.addvec) {
    const f16* ar = p.addvec + (((ok ? row : 0) + p.row0) / p.rows_per_vec) * C + l * 8;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const f16x8 a = *(const f16x8*)(ar + k * L * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) h[k][e] = (f16)((float)h[k][e] + (float)a[e]);
      if (p.Xout && ok) *(f16x8*)(p.Xout + row * C + (l + k * L) * 8) = h[k];
    }
  }
  float x[NV][8];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) { x[k][e] = (float)h[k][e]; sum += x[k][e]; }
#pragma unroll
  for (int o = L / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum * (1.0f / (40 * L));
  float var = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) { x[k][e] -= mean; var += x[k][e] * x[k][e]; }
#pragma unroll
  for (int o = L / 2; o > 0; o >>= 1) var += __shfl_xor(var, o);
  const float rstd = rsqrtf(var * (1.0f / (40 * L)) + p.eps);
  // (a row's lanes leave together: the 4-lane block-maximum exchange of the fp8 form never crosses rows)
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (!ok) continue;
    const f16x8 g = gm[k], b = bt[k];
    f16x8 y;
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = (f16)(x[k][e] * rstd * (float)g[e] + (float)b[e]);
    if (!p.Y8) { *(f16x8*)(p.Y + row * C + (l + k * L) * 8) = y; continue; }
    // MX-fp8 output (ln_kernel's, kernels/mx8.hip semantics on the fp16-rounded values): vector v = l + k L; the 4 lanes l = 4j .. 4j + 3 hold one 32-element block
    const int v = l + k * L;
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf((float)y[e]));
    amax = fmaxf(amax, __shfl_xor(amax, 1));
    amax = fmaxf(amax, __shfl_xor(amax, 2));
    int ex = 0;
    if (amax > 0.f) { (void)frexpf(amax, &ex); ex = ex - 1 - 8; }
    ex = min(max(ex, -127), 127);
    const float inv = ldexpf(1.0f, -ex);
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = fminf(fmaxf((float)y[e] * inv, -448.f), 448.f);
    int p0 = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], 0, false); p0 = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], p0, true);
    int p1 = __builtin_amdgcn_cvt_pk_fp8_f32(t[4], t[5], 0, false); p1 = __builtin_amdgcn_cvt_pk_fp8_f32(t[6], t[7], p1, true);
    *(uint2*)(p.Y8 + row * C + v * 8) = make_uint2((unsigned)p0, (unsigned)p1);
    if ((l & 3) == 0) ((unsigned char*)(p.S8 + (long)(v >> 4) * p.ld_s8 + row))[(v & 15) >> 2] = (unsigned char)(ex + 127);
  }
}

void launch_layernorm(const LayerNormP& p, hipStream_t s) {
  UG_REQUIRE(p.C % 8 == 0, "LayerNorm C must be a multiple of 8");
  const int vpl = cdiv(p.C / 8, 64);
  UG_REQUIRE(vpl <= 4, "LayerNorm C too large");
  if (p.Y8) UG_REQUIRE(p.C % 128 == 0 && p.S8 && p.ld_s8 >= p.M, "LayerNorm MX-fp8 output needs C % 128 == 0 and a scale buffer");
  static const bool no40 = getenv("UG_LN_NO40") != nullptr;   // A/B aid
  if (!no40 && (p.C == 320 || p.C == 640 || p.C == 1280)) {
    const int L = p.C / 40, rows_per_block = 4 * (64 / L);
    dim3 grid(cdiv(p.M, rows_per_block)), block(256);
    if (L == 8) hipLaunchKernelGGL(ln40_kernel<8>, grid, block, 0, s, p);
    else if (L == 16) hipLaunchKernelGGL(ln40_kernel<16>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(ln40_kernel<32>, grid, block, 0, s, p);
    UG_CHECK(hipGetLastError());
    return;
  }
  dim3 grid(cdiv(p.M, 4)), block(256);
  switch (vpl) {
    case 1: hipLaunchKernelGGL(ln_kernel<1>, grid, block, 0, s, p); break;
    case 2: hipLaunchKernelGGL(ln_kernel<2>, grid, block, 0, s, p); break;
    case 3: hipLaunchKernelGGL(ln_kernel<3>, grid, block, 0, s, p); break;
    default: hipLaunchKernelGGL(ln_kernel<4>, grid, block, 0, s, p); break;
  }
  UG_CHECK(hipGetLastError());
}

