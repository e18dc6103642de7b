// MX-fp8 quantisation (OCP MX: 32-element blocks along K share one e8m0 power-of-two scale; elements are OCP e4m3) for the fp8
// linears of BASELINE configs[4] (north_star: "fp8 MFMA ... (CDNA4 fp8)").  The reference has no fp8 path - this is the build's
// reduced-precision option, reported with its measured error next to the fp16 path.
//
//   scale  e = floor(log2(amax_block)) - 8   (e4m3's largest binade is 2^8: 448 = 1.75 * 2^8), stored biased by 127
//   q      = e4m3_rne(x * 2^-e), saturated to +-448
// Layout of the scales: one dword per (row, 128-wide K step) holding the four block exponents, K-step-major
// (scales[kstep * ld + row]) - the GEMM fetches the dwords of a tile's rows with one 1 KiB direct-to-LDS load per K step.
// Eight lanes share a 128-element K step of one row: 32-byte loads, 16-byte stores, two shuffles for the block maximum.
#include "../common.h"
#include <algorithm>

__global__ __launch_bounds__(256) void k_quant_mx8(const f16* x, long ldx, long M, int K, unsigned char* q, unsigned* scales, long ld_s) {
  const int ksteps = K / 128;
  const long nitem = M * ksteps * 8;                       // one item = 16 consecutive elements
  for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < ((nitem + 63) / 64) * 64; it += (long)gridDim.x * 256) {
    const bool ok = it < nitem;
    const long unit = ok ? it : nitem - 1;
    const long rk = unit >> 3; const int sub = (int)(unit & 7);          // (row, K step), 16-element slice inside the step
    const long m = rk / ksteps; const int ks = (int)(rk - m * ksteps);
    const f16* src = x + m * ldx + ks * 128 + sub * 16;
    const f16x8 a = *(const f16x8*)src, b = *(const f16x8*)(src + 8);
    float v[16], amax = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] = (float)a[e]; v[8 + e] = (float)b[e]; }
#pragma unroll
    for (int e = 0; e < 16; ++e) amax = fmaxf(amax, fabsf(v[e]));
    amax = fmaxf(amax, __shfl_xor(amax, 1));              // lanes 2b, 2b+1 share a 32-element block
    int ex = 0;
    if (amax > 0.f) { (void)frexpf(amax, &ex); ex = ex - 1 - 8; }   // floor(log2 amax) - 8
    ex = min(max(ex, -127), 127);
    const float inv = ldexpf(1.0f, -ex);
    unsigned w[4];
#pragma unroll
    for (int e = 0; e < 16; e += 4) {
      float t[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) t[r] = fminf(fmaxf(v[e + r] * inv, -448.f), 448.f);
      int pk = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], 0, false);
      pk = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], pk, true);
      w[e >> 2] = (unsigned)pk;
    }
    // the four block exponents of the K step -> one dword (blocks are lane pairs: sub = 0,1 | 2,3 | 4,5 | 6,7)
    const unsigned be = (unsigned)(ex + 127);
    const int lane = threadIdx.x & 63, base = lane & ~7;
    const unsigned e0 = __shfl(be, base), e1 = __shfl(be, base + 2), e2 = __shfl(be, base + 4), e3 = __shfl(be, base + 6);
    if (ok) {
      *(uint4*)(q + m * (long)K + ks * 128 + sub * 16) = make_uint4(w[0], w[1], w[2], w[3]);
      if (sub == 0) scales[(long)ks * ld_s + m] = e0 | (e1 << 8) | (e2 << 16) | (e3 << 24);
    }
  }
}
void launch_quant_mx8(const f16* x, long ldx, long M, int K, unsigned char* q, unsigned* scales, long ld_s, hipStream_t s) {
  UG_REQUIRE(K % 128 == 0 && ldx % 8 == 0 && ld_s >= M, "quant_mx8: K % 128 == 0, 16-byte aligned rows");
  const long nitem = M * (K / 128) * 8;
  hipLaunchKernelGGL(k_quant_mx8, dim3((unsigned)std::min<long>((nitem + 255) / 256, 16384)), dim3(256), 0, s, x, ldx, M, K, q, scales, ld_s);
  UG_CHECK(hipGetLastError());
}
