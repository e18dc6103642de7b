// fp16 MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X, CDNA4).
//
//   Out[M,N] = epilogue( A[M,K] . W[N,K]^T )
//
// * 128 x BN x 64 block tile (BN = 128 or 64), 256 threads = 4 wave64 in a 2x2 grid, each wave
//   owning a 64 x (BN/2) accumulator made of 16x16x32 f16 MFMA tiles (fp32 accumulate).
// * Operand tiles are staged global -> LDS with the direct-to-LDS 16-byte loads
//   (global_load_lds_dwordx4): no VGPR round trip.  The LDS image is lane-linear, so the
//   XOR swizzle that makes the ds_read_b128 fragment reads bank-conflict free is applied on
//   the per-lane *source* address (chunk ^= (row>>1)&7) and mirrored on the read side.
// * Double-buffered LDS, one barrier per K-tile: the prefetch of tile t+1 is issued right
//   after the barrier and lands while tile t is multiplied.
// * The A operand is either a dense row-major matrix or the im2col view of channels-last
//   tensors generated on the fly by the address computation (taps over t/y/x, stride,
//   nearest-2x upsample, concat of two sources).  Padding / out-of-range rows read a zero page.
// * MFMA operands are swapped (W fragment as "A", activation fragment as "B") and the W rows of
//   a tile are permuted when they are staged, so every lane ends up with 16 (BN=128) or 8
//   (BN=64) *contiguous* output columns of one output row: bias / residual / GEGLU / store are
//   all 16-byte vector operations.
#include "../common.h"

#define BM 128
#define BK 64

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

template <int BN, bool CONV, bool UNI>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmP p) {
  constexpr int NT = BN / 32;   // 16-wide MFMA column tiles per wave
  constexpr int WID = 4 * NT;   // contiguous output columns per lane
  extern __shared__ __attribute__((aligned(16))) f16 smem[];
  f16* As = smem;                    // [2][BM*BK]
  f16* Bs = smem + 2 * BM * BK;      // [2][BN*BK]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (p.N + BN - 1) / BN;
  const int tn = blockIdx.x % ntn, tm = blockIdx.x / ntn;
  const int m0 = tm * BM, n0 = tn * BN;

  const int bz = blockIdx.z;
  const int bo = bz / p.nb_inner, bi = bz - bo * p.nb_inner;
  const f16* A0 = p.A0 + bo * p.sA_o + bi * p.sA_i;
  const f16* Wb = p.W + bo * p.sW_o + bi * p.sW_i;
  const long out_off = bo * p.sO_o + bi * p.sO_i;

  const int pc = lane & 7;   // physical 16-byte chunk inside the 128-byte LDS row
  const int Cin = p.C0 + p.C1;

  // ---- per-thread A rows ----
  int a_lc[4];               // logical chunk (after un-swizzling)
  bool a_ok[4];
  const f16* a_ptr[4];       // dense: row base pointer
  int a_t[4], a_y[4], a_x[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (wave * 4 + j) * 8 + (lane >> 3);
    a_lc[j] = pc ^ ((r >> 1) & 7);
    const int m = m0 + r;
    a_ok[j] = m < p.M;
    if (CONV) {
      const int hw = p.Ho * p.Wo;
      const int t = m / hw, rem = m - t * hw;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_t[j] = t; a_y[j] = oy * p.stride - p.pad_t; a_x[j] = ox * p.stride - p.pad_l;
      a_ptr[j] = nullptr;
    } else {
      a_ptr[j] = A0 + (long)m * p.C0;
      a_t[j] = a_y[j] = a_x[j] = 0;
    }
  }
  // ---- per-thread W rows (permuted: LDS row lr holds W row n0 + perm(lr)) ----
  int b_lc[NT];
  const f16* b_ptr[NT];
  bool b_ok[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int lr = (wave * NT + j) * 8 + (lane >> 3);
    b_lc[j] = pc ^ ((lr >> 1) & 7);
    const int half = lr / (BN / 2), rem = lr % (BN / 2);
    const int jj = rem >> 4, i = rem & 15;
    const int nloc = half * (BN / 2) + (i >> 2) * WID + jj * 4 + (i & 3);
    const int n = n0 + nloc;
    b_ok[j] = n < p.N;
    b_ptr[j] = Wb + (long)n * p.ldw;
  }

  auto stage = [&](int kt0, int buf) {
    // A tile
    int it = 0, iy = 0, ix = 0, cb = 0;
    if (CONV && UNI) {  // whole 64-wide K tile sits inside one tap
      const int tap = kt0 / Cin;
      cb = kt0 - tap * Cin;
      it = tap / (p.ky * p.kx);
      const int r2 = tap - it * (p.ky * p.kx);
      iy = r2 / p.kx; ix = r2 - iy * p.kx;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f16* src = p.zero;
      const int k = kt0 + a_lc[j] * 8;
      if (a_ok[j] && k < p.K) {
        if (!CONV) {
          src = a_ptr[j] + k;
        } else {
          int c;
          if (UNI) {
            c = cb + a_lc[j] * 8;
          } else {
            const int tap = k / Cin;
            c = k - tap * Cin;
            it = tap / (p.ky * p.kx);
            const int r2 = tap - it * (p.ky * p.kx);
            iy = r2 / p.kx; ix = r2 - iy * p.kx;
          }
          const int tt = a_t[j] + it - (p.kt >> 1);
          const int y = a_y[j] + iy, x = a_x[j] + ix;
          if (tt >= 0 && tt < p.T && y >= 0 && y < p.Hi * p.ups && x >= 0 && x < p.Wi * p.ups) {
            const long pix = ((long)tt * p.Hi + (y / p.ups)) * p.Wi + (x / p.ups);
            src = (c < p.C0) ? (A0 + pix * p.C0 + c) : (p.A1 + pix * p.C1 + (c - p.C0));
          }
        }
      }
      __builtin_amdgcn_global_load_lds((gptr_t)src,
                                       (lptr_t)(As + buf * (BM * BK) + (wave * 4 + j) * 8 * BK), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int k = kt0 + b_lc[j] * 8;
      const f16* src = (b_ok[j] && k < p.K) ? (b_ptr[j] + k) : p.zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src,
                                       (lptr_t)(Bs + buf * (BN * BK) + (wave * NT + j) * 8 * BK), 16, 0, 0);
    }
  };

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;
  const int l15 = lane & 15, g = lane >> 4;
  const int sw = l15 >> 1;  // read-side swizzle term: rows are (multiple of 16) + l15

  stage(0, 0);
  int buf = 0;
  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < nk) stage((t + 1) * BK, buf ^ 1);
    const f16* Ab = As + buf * (BM * BK) + (wm * 64 + l15) * BK;
    const f16* Bb = Bs + buf * (BN * BK) + (wn * (BN / 2) + l15) * BK;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = ((kk * 4 + g) ^ sw) * 8;
      f16x8 af[4], bf[NT];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *(const f16x8*)(Ab + i * 16 * BK + ch);
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j] = *(const f16x8*)(Bb + j * 16 * BK + ch);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    buf ^= 1;
  }

  // ---- epilogue: lane holds WID contiguous columns of row m ----
  const int nb = n0 + wn * (BN / 2) + g * WID;
  const bool geglu = (p.flags & UG_F_GEGLU) != 0;
  const bool of32 = (p.flags & UG_F_OUT_F32) != 0;
  const int Nout = geglu ? p.N / 2 : p.N;
  const int ob = geglu ? nb / 2 : nb;          // first output column of this lane
  const int OW = geglu ? WID / 2 : WID;        // output columns of this lane
  const bool full = (nb + WID <= p.N);
  const bool vec = full && ((p.ldo & 7) == 0) && (!p.R1 || (p.ldr1 & 7) == 0) &&
                   (!p.R2 || (p.ldr2 & 7) == 0) && (OW % 8 == 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + l15;
    if (m >= p.M) continue;
    float v[WID];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[j * 4 + r] = acc[i][j][r];
    if (full) {
      if (p.bias) {
#pragma unroll
        for (int e = 0; e < WID; e += 8) {
          const f16x8 b = *(const f16x8*)(p.bias + nb + e);
#pragma unroll
          for (int q = 0; q < 8; ++q) v[e + q] += (float)b[q];
        }
      }
      if (p.bias2) {
#pragma unroll
        for (int e = 0; e < WID; e += 8) {
          const f16x8 b = *(const f16x8*)(p.bias2 + nb + e);
#pragma unroll
          for (int q = 0; q < 8; ++q) v[e + q] += (float)b[q];
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < WID; ++e)
        if (nb + e < p.N) {
          if (p.bias) v[e] += (float)p.bias[nb + e];
          if (p.bias2) v[e] += (float)p.bias2[nb + e];
        }
    }
    if (geglu) {
      if (WID == 16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * gelu_f(v[8 + e]);
      }
    }
    if (vec) {
#pragma unroll
      for (int e = 0; e < OW; e += 8) {
        float o[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = p.c0 * v[e + q];
        if (p.R1) {
          const f16x8 r = *(const f16x8*)(p.R1 + (long)m * p.ldr1 + ob + e);
#pragma unroll
          for (int q = 0; q < 8; ++q) o[q] += p.c1 * (float)r[q];
        }
        if (p.R2) {
          const f16x8 r = *(const f16x8*)(p.R2 + (long)m * p.ldr2 + ob + e);
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
        if (of32) {
          float* O = (float*)p.Out + out_off + (long)m * p.ldo + ob + e;
          *(f32x4*)O = (f32x4){o[0], o[1], o[2], o[3]};
          *(f32x4*)(O + 4) = (f32x4){o[4], o[5], o[6], o[7]};
        } else {
          f16x8 h;
#pragma unroll
          for (int q = 0; q < 8; ++q) h[q] = (f16)o[q];
          *(f16x8*)((f16*)p.Out + out_off + (long)m * p.ldo + ob + e) = h;
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < OW; ++e) {
        const int n = ob + e;
        if (n < Nout) {
          float o = p.c0 * v[e];
          if (p.R1) o += p.c1 * (float)p.R1[(long)m * p.ldr1 + n];
          if (p.R2) o += p.c2 * (float)p.R2[(long)m * p.ldr2 + n];
          if (p.act == UG_ACT_SILU) o = silu_f(o);
          else if (p.act == UG_ACT_GELU) o = gelu_f(o);
          if (of32) ((float*)p.Out)[out_off + (long)m * p.ldo + n] = o;
          else ((f16*)p.Out)[out_off + (long)m * p.ldo + n] = (f16)o;
        }
      }
    }
  }
}

template <int BN, bool CONV, bool UNI>
static void launch_t(const GemmP& p, int batch, hipStream_t s) {
  const int ntm = cdiv(p.M, BM), ntn = cdiv(p.N, BN);
  const size_t lds = (size_t)2 * (BM * BK + BN * BK) * sizeof(f16);
  dim3 grid(ntm * ntn, 1, batch);
  hipLaunchKernelGGL((gemm_kernel<BN, CONV, UNI>), grid, dim3(256), lds, s, p);
}

void launch_gemm(const GemmP& p, int batch, hipStream_t s) {
  UG_REQUIRE(p.K % 8 == 0, "GEMM K must be a multiple of 8");
  UG_REQUIRE(p.ldw % 8 == 0, "GEMM ldw must be a multiple of 8");
  UG_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "empty GEMM");
  UG_REQUIRE(p.zero != nullptr, "zero page missing");
  UG_REQUIRE(p.nb_inner >= 1, "nb_inner");
  const bool geglu = p.flags & UG_F_GEGLU;
  if (geglu) UG_REQUIRE(p.N % 128 == 0, "GEGLU GEMM needs N % 128 == 0");
  // BN=64 when N is not a multiple of 128 but wastes less at 64 (e.g. 320, 4, 8)
  const bool bn64 = !geglu && (cdiv(p.N, 64) * 64 < cdiv(p.N, 128) * 128);
  if (p.conv) {
    const int Cin = p.C0 + p.C1;
    UG_REQUIRE(p.C0 % 8 == 0 && p.C1 % 8 == 0, "conv channel counts must be multiples of 8");
    UG_REQUIRE(p.K == Cin * p.kt * p.ky * p.kx, "conv K mismatch");
    UG_REQUIRE(p.M == p.T * p.Ho * p.Wo, "conv M mismatch");
    UG_REQUIRE(p.ups == 1 || p.ups == 2, "ups");
    const bool uni = (Cin % BK) == 0;
    if (bn64) { if (uni) launch_t<64, true, true>(p, batch, s); else launch_t<64, true, false>(p, batch, s); }
    else      { if (uni) launch_t<128, true, true>(p, batch, s); else launch_t<128, true, false>(p, batch, s); }
  } else {
    UG_REQUIRE(p.C0 % 8 == 0, "dense lda must be a multiple of 8");
    if (bn64) launch_t<64, false, false>(p, batch, s); else launch_t<128, false, false>(p, batch, s);
  }
  UG_CHECK(hipGetLastError());
}
