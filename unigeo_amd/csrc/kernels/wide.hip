// fp32-grade arithmetic on the fp16 matrix cores, for the one place the reference leaves fp16: the VAE *encoder*
// (diffusers' force_upcast: the pipeline casts the VAE to float32 around encode; call site
// /root/reference/model/depthcrafter.py:24-29,80-90).
//
// The checkpoint's VAE weights are fp16 values (variant="fp16"), merely widened to fp32 by the up-cast, so a layer
// y = W x with fp32 x is reproduced exactly (to fp32 rounding of the sum) by splitting only the ACTIVATION:
//     x = hi + lo,  hi = fp16(x),  lo = fp16(x - hi)            (22 significant bits; |lo| floor 2^-24)
//     y = W hi + W lo                                           (fp16 x fp16 products are exact in the fp32 accumulator)
// i.e. the same implicit-GEMM kernels run on a K-doubled problem: A' = [hi | lo] per tap, W' = [W | W].  Everything
// between the GEMMs - residual stream, GroupNorm statistics and affine, SiLU, softmax - lives here in fp32.
// Activation x activation products (the mid-block attention's QK^T and PV) need three terms: a b ~ ah bh + al bh + ah bl.
//
// All kernels are HBM-bound elementwise / row passes over fp32 tensors: 16-byte loads, 8-byte fp16 stores.
#include "../common.h"
#include <algorithm>

typedef _Float16 h4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split4(const f32x4 x, h4& hi, h4& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f16 h = (f16)x[e];
    hi[e] = h;
    lo[e] = (f16)(x[e] - (float)h);
  }
}

// ---------------------------------------------------------------------------------------------------- elementwise
__global__ __launch_bounds__(256) void k_split_pair(const float* x, f16* y, long M, int C) {
  const int nv = C / 4;
  const long n = M * nv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long m = i / nv; const int c = (int)(i - m * nv) * 4;
    h4 hi, lo;
    split4(*(const f32x4*)(x + m * C + c), hi, lo);
    *(h4*)(y + m * 2 * C + c) = hi;
    *(h4*)(y + m * 2 * C + C + c) = lo;
  }
}
void launch_split_pair(const float* x, f16* y, long M, int C, hipStream_t s) {
  UG_REQUIRE(C % 4 == 0, "split_pair: C % 4");
  const long n = M * (C / 4);
  hipLaunchKernelGGL(k_split_pair, dim3((unsigned)std::min<long>((n + 255) / 256, 16384)), dim3(256), 0, s, x, y, M, C);
  UG_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_add_f32(const float* a, const float* b, float* y, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 u = ((const f32x4*)a)[i], v = ((const f32x4*)b)[i];
    ((f32x4*)y)[i] = u + v;
  }
}
void launch_add_f32(const float* a, const float* b, float* y, long n, hipStream_t s) {
  UG_REQUIRE(n % 4 == 0, "add_f32: n % 4");
  hipLaunchKernelGGL(k_add_f32, dim3((unsigned)std::min<long>((n / 4 + 255) / 256, 16384)), dim3(256), 0, s, a, b, y, n / 4);
  UG_CHECK(hipGetLastError());
}

// y[m, 0:C] = x[m*ldx + 0:C] (fp32 -> fp16): the posterior mean half of the moments
__global__ __launch_bounds__(256) void k_take_cols_f16(const float* x, long ldx, f16* y, long M, int C) {
  const long n = M * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long m = i / C; const int c = (int)(i - m * C);
    y[i] = (f16)x[m * ldx + c];
  }
}
void launch_take_cols_f16(const float* x, long ldx, f16* y, long M, int C, hipStream_t s) {
  hipLaunchKernelGGL(k_take_cols_f16, dim3((unsigned)std::min<long>((M * C + 255) / 256, 4096)), dim3(256), 0, s, x, ldx, y, M, C);
  UG_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------- GroupNorm
// fp32 in -> (x - mean) * rstd * gamma + beta (-> SiLU) -> hi/lo pair out.  Statistics per (frame, group): per-thread
// fp32 partial sums over <= a few hundred values, everything above that in fp64, fixed order (deterministic).
struct Gn32P { const float* X; f16* Y; int T, HW, C, G; float eps; int silu; const f16* gamma; const f16* beta; double* part; float* mr; };

// Geometry: nv = C/8 eight-channel vectors per row (two 16-byte loads, one 16-byte store per plane), 256/nv rows per iteration.
__global__ __launch_bounds__(256) void k_gn32_stats(const Gn32P p, int nchunk, int rpc) {
  __shared__ double rs[256], rq[256];
  const int nv = p.C / 8, rpi = 256 / nv, tid = threadIdx.x;
  const int rsub = tid / nv, v = tid - rsub * nv;
  const int t = blockIdx.y, r0 = blockIdx.x * rpc, r1 = min(r0 + rpc, p.HW);
  float s = 0.f, q = 0.f;
  if (rsub < rpi) {
    const float* base = p.X + (long)t * p.HW * p.C + v * 8;
    int r = r0 + rsub;
    for (; r + rpi < r1; r += 2 * rpi) {          // two rows = four 16-byte loads in flight
      f32x4 x[4];
#pragma unroll
      for (int u = 0; u < 2; ++u) { x[2 * u] = *(const f32x4*)(base + (long)(r + u * rpi) * p.C); x[2 * u + 1] = *(const f32x4*)(base + (long)(r + u * rpi) * p.C + 4); }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s += x[u][e]; q += x[u][e] * x[u][e]; }
    }
    for (; r < r1; r += rpi) {
      const f32x4 a = *(const f32x4*)(base + (long)r * p.C), b = *(const f32x4*)(base + (long)r * p.C + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s += a[e] + b[e]; q += a[e] * a[e] + b[e] * b[e]; }
    }
  }
  rs[tid] = (rsub < rpi) ? (double)s : 0.0; rq[tid] = (rsub < rpi) ? (double)q : 0.0;
  __syncthreads();
  // group g owns channels [g*cpg, (g+1)*cpg): cpg is a multiple of 4; an 8-channel vector may straddle two groups only when cpg == 4
  const int cpg = p.C / p.G;
  if (tid < p.G) {
    double a = 0.0, b = 0.0;
    if (cpg % 8 == 0) {
      const int vpg = cpg / 8;
      for (int rr = 0; rr < rpi; ++rr)
        for (int k = 0; k < vpg; ++k) { a += rs[rr * nv + tid * vpg + k]; b += rq[rr * nv + tid * vpg + k]; }
    }
    double* d = p.part + (((long)t * nchunk + blockIdx.x) * p.G + tid) * 2;
    d[0] = a; d[1] = b;
  }
}
// cpg == 4: the 8-channel vectors straddle two groups - the statistics pass uses 4-channel vectors instead
__global__ __launch_bounds__(256) void k_gn32_stats4(const Gn32P p, int nchunk, int rpc) {
  __shared__ double rs[256], rq[256];
  const int nv = p.C / 4, rpi = 256 / nv, tid = threadIdx.x;
  const int rsub = tid / nv, v = tid - rsub * nv;
  const int t = blockIdx.y, r0 = blockIdx.x * rpc, r1 = min(r0 + rpc, p.HW);
  float s = 0.f, q = 0.f;
  if (rsub < rpi) {
    const float* base = p.X + (long)t * p.HW * p.C + v * 4;
    int r = r0 + rsub;
    for (; r + 3 * rpi < r1; r += 4 * rpi) {
      f32x4 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = *(const f32x4*)(base + (long)(r + u * rpi) * p.C);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s += x[u][e]; q += x[u][e] * x[u][e]; }
    }
    for (; r < r1; r += rpi) {
      const f32x4 x = *(const f32x4*)(base + (long)r * p.C);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s += x[e]; q += x[e] * x[e]; }
    }
  }
  rs[tid] = (rsub < rpi) ? (double)s : 0.0; rq[tid] = (rsub < rpi) ? (double)q : 0.0;
  __syncthreads();
  const int vpg = (p.C / p.G) / 4;
  if (tid < p.G) {
    double a = 0.0, b = 0.0;
    for (int rr = 0; rr < rpi; ++rr)
      for (int k = 0; k < vpg; ++k) { a += rs[rr * nv + tid * vpg + k]; b += rq[rr * nv + tid * vpg + k]; }
    double* d = p.part + (((long)t * nchunk + blockIdx.x) * p.G + tid) * 2;
    d[0] = a; d[1] = b;
  }
}

__global__ __launch_bounds__(256) void k_gn32_finalize(const Gn32P p, int nchunk) {
  // grid (T); 256 / G threads share one group's chunk partials (fixed order: deterministic), then a short LDS combine
  __shared__ double sa[256], sb[256];
  const int t = blockIdx.x, tid = threadIdx.x, G = p.G;
  const int SUB = 256 / G, g = tid % G, sub = tid / G;
  double a = 0.0, b = 0.0;
  if (sub < SUB)
    for (int ch = sub; ch < nchunk; ch += SUB) {
      const double* d = p.part + (((long)t * nchunk + ch) * G + g) * 2;
      a += d[0]; b += d[1];
    }
  sa[tid] = a; sb[tid] = b;
  __syncthreads();
  if (tid < G) {
    for (int s2 = 1; s2 < SUB; ++s2) { a += sa[tid + s2 * G]; b += sb[tid + s2 * G]; }
    const double n = (double)(p.C / G) * p.HW;
    const double mean = a / n;
    double var = b / n - mean * mean;
    if (var < 0.0) var = 0.0;
    p.mr[((long)t * G + tid) * 2 + 0] = (float)mean;
    p.mr[((long)t * G + tid) * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
  }
}

__global__ __launch_bounds__(256) void k_gn32_apply(const Gn32P p, int rpc) {
  const int nv = p.C / 8, rpi = 256 / nv, tid = threadIdx.x;
  const int rsub = tid / nv, v = tid - rsub * nv;
  if (rsub >= rpi) return;
  const int t = blockIdx.y, r0 = blockIdx.x * rpc, r1 = min(r0 + rpc, p.HW);
  const int c = v * 8, cpg = p.C / p.G;
  float mean[8], ga[8], be[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int g = (c + e) / cpg;
    mean[e] = p.mr[((long)t * p.G + g) * 2];
    ga[e] = p.mr[((long)t * p.G + g) * 2 + 1] * (float)p.gamma[c + e]; be[e] = (float)p.beta[c + e];
  }
  auto emit = [&](long m, const f32x4 xa, const f32x4 xb) {
    f16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float f = ((e < 4 ? xa[e] : xb[e - 4]) - mean[e]) * ga[e] + be[e];
      if (p.silu) f = f * __builtin_amdgcn_rcpf(1.0f + __expf(-f));
      const f16 h = (f16)f;
      hi[e] = h; lo[e] = (f16)(f - (float)h);
    }
    *(f16x8*)(p.Y + m * 2 * p.C + c) = hi;
    *(f16x8*)(p.Y + m * 2 * p.C + p.C + c) = lo;
  };
  int r = r0 + rsub;
  for (; r + rpi < r1; r += 2 * rpi) {              // two rows = four 16-byte loads in flight
    f32x4 x[4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float* src = p.X + ((long)t * p.HW + r + u * rpi) * p.C + c;
      x[2 * u] = *(const f32x4*)src; x[2 * u + 1] = *(const f32x4*)(src + 4);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) emit((long)t * p.HW + r + u * rpi, x[2 * u], x[2 * u + 1]);
  }
  for (; r < r1; r += rpi) {
    const float* src = p.X + ((long)t * p.HW + r) * p.C + c;
    emit((long)t * p.HW + r, *(const f32x4*)src, *(const f32x4*)(src + 4));
  }
}

static inline void gn32_chunks(int T, int HW, int C, int& nchunk, int& rpc) {
  const int rpi = 256 / (C / 8);
  rpc = std::max((HW + cdiv(2048, T) - 1) / cdiv(2048, T), 4 * rpi);
  nchunk = cdiv(HW, rpc);
}
size_t gn32_ws_bytes(int T, int HW, int C, int G) {
  int nchunk, rpc;
  gn32_chunks(T, HW, C, nchunk, rpc);
  return (size_t)T * nchunk * G * 2 * sizeof(double) + (size_t)T * G * 2 * sizeof(float) + 256;
}
// X fp32 [T*HW, C] -> Y fp16 pair [T*HW, 2C]; ws from gn32_ws_bytes
void launch_gn32_pair(const float* X, f16* Y, int T, int HW, int C, int G, float eps, int silu, const f16* gamma, const f16* beta,
                      void* ws, hipStream_t s) {
  UG_REQUIRE(C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0, "gn32: C/8 must divide 256");
  UG_REQUIRE(C % G == 0 && (C / G) % 4 == 0 && G <= 256, "gn32: channels per group must be a multiple of 4");
  int nchunk, rpc;
  gn32_chunks(T, HW, C, nchunk, rpc);
  Gn32P p; p.X = X; p.Y = Y; p.T = T; p.HW = HW; p.C = C; p.G = G; p.eps = eps; p.silu = silu; p.gamma = gamma; p.beta = beta;
  p.part = (double*)ws; p.mr = (float*)((char*)ws + (size_t)T * nchunk * G * 2 * sizeof(double));
  if ((C / G) % 8 == 0) hipLaunchKernelGGL(k_gn32_stats, dim3(nchunk, T), dim3(256), 0, s, p, nchunk, rpc);
  else hipLaunchKernelGGL(k_gn32_stats4, dim3(nchunk, T), dim3(256), 0, s, p, nchunk, rpc);
  hipLaunchKernelGGL(k_gn32_finalize, dim3(T), dim3(256), 0, s, p, nchunk);
  hipLaunchKernelGGL(k_gn32_apply, dim3(nchunk, T), dim3(256), 0, s, p, rpc);
  UG_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------- attention terms
// qkv fp32 [M, 3C] -> Aq [M, 3C] = [qh | ql | qh], Bk [M, 3C] = [kh | kh | kl]   (QK^T ~ qh kh + ql kh + qh kl)
__global__ __launch_bounds__(256) void k_qk_terms(const float* qkv, f16* Aq, f16* Bk, long M, int C) {
  const int nv = C / 4;
  const long n = M * nv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long m = i / nv; const int c = (int)(i - m * nv) * 4;
    h4 qh, ql, kh, kl;
    split4(*(const f32x4*)(qkv + m * 3 * C + c), qh, ql);
    split4(*(const f32x4*)(qkv + m * 3 * C + C + c), kh, kl);
    f16* a = Aq + m * 3 * C + c; f16* b = Bk + m * 3 * C + c;
    *(h4*)a = qh; *(h4*)(a + C) = ql; *(h4*)(a + 2 * C) = qh;
    *(h4*)b = kh; *(h4*)(b + C) = kh; *(h4*)(b + 2 * C) = kl;
  }
}
void launch_qk_terms(const float* qkv, f16* Aq, f16* Bk, long M, int C, hipStream_t s) {
  UG_REQUIRE(C % 4 == 0, "qk_terms: C % 4");
  hipLaunchKernelGGL(k_qk_terms, dim3((unsigned)std::min<long>((M * (C / 4) + 255) / 256, 16384)), dim3(256), 0, s, qkv, Aq, Bk, M, C);
  UG_CHECK(hipGetLastError());
}

// V of qkv fp32 [B*S, 3C] -> Vt [B][C][3*Spad] = [vh^T | vh^T | vl^T], zero tails   (PV ~ ph vh + pl vh + ph vl)
__global__ void k_vt_terms(const float* qkv, f16* Vt, int S, int Spad, int C) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, s0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int s = s0 + i, c = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (s < S && c < C) ? qkv[((long)b * S + s) * 3 * C + 2 * C + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int c = c0 + i, s = s0 + threadIdx.x;
    if (c < C && s < Spad) {
      const float x = tile[threadIdx.x][i];
      const f16 h = (f16)x, l = (f16)(x - (float)h);
      f16* d = Vt + ((long)b * C + c) * 3 * Spad + s;
      d[0] = h; d[Spad] = h; d[2 * Spad] = l;
    }
  }
}
void launch_vt_terms(const float* qkv, f16* Vt, int B, int S, int Spad, int C, hipStream_t s) {
  hipLaunchKernelGGL(k_vt_terms, dim3(cdiv(C, 32), cdiv(Spad, 32), B), dim3(32, 8), 0, s, qkv, Vt, S, Spad, C);
  UG_CHECK(hipGetLastError());
}

// row softmax of fp32 scores [rows, ld_in] (first S valid) -> P [rows, 3*Spad] = [ph | pl | ph], zero tails; one wave per row
__global__ __launch_bounds__(256) void k_softmax_pair(const float* in, long ld_in, f16* out, int Spad, long rows, int S) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* x = in + row * ld_in;
  float mx = -3.0e38f;
  for (int i = lane; i < S; i += 64) mx = fmaxf(mx, x[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int i = lane; i < S; i += 64) sum += expf(x[i] - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float inv = 1.0f / sum;
  f16* y = out + row * 3 * Spad;
  for (int i = lane; i < Spad; i += 64) {
    const float pr = (i < S) ? expf(x[i] - mx) * inv : 0.f;
    const f16 h = (f16)pr, l = (f16)(pr - (float)h);
    y[i] = h; y[Spad + i] = l; y[2 * Spad + i] = h;
  }
}
void launch_softmax_pair(const float* in, long ld_in, f16* out, int Spad, long rows, int S, hipStream_t s) {
  hipLaunchKernelGGL(k_softmax_pair, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, s, in, ld_in, out, Spad, rows, S);
  UG_CHECK(hipGetLastError());
}
