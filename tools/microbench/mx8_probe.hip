// Layout probe for v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3, e8m0 block scales): which K elements and which scale
// does lane l hold?  Tries the hypotheses below against a host reference and prints the max error of each (0 = the layout).
//   build: hipcc --offload-arch=gfx950 -O2 mx8_probe.hip -o mx8_probe      run on the GPU box: ./mx8_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

static float e4m3(unsigned char b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  const float v = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -v : v;
}
// hyp: 0 = lane l holds K [32*(l>>4), +32) contiguous; 1 = lane l holds K [16*(l>>4), +16) and [64 + 16*(l>>4), +16)
__global__ void k(const unsigned char* A, const unsigned char* B, const unsigned char* SA, const unsigned char* SB, float* C, int hyp) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  unsigned char a[32], b[32];
  for (int i = 0; i < 32; ++i) {
    const int kk = hyp == 0 ? 32 * g + i : (i < 16 ? 16 * g + i : 64 + 16 * g + (i - 16));
    a[i] = A[r * 128 + kk]; b[i] = B[r * 128 + kk];
  }
  v8i av, bv;
  for (int w = 0; w < 8; ++w) {
    av[w] = a[4 * w] | (a[4 * w + 1] << 8) | (a[4 * w + 2] << 16) | (a[4 * w + 3] << 24);
    bv[w] = b[4 * w] | (b[4 * w + 1] << 8) | (b[4 * w + 2] << 16) | (b[4 * w + 3] << 24);
  }
  const int sa = SA[r * 4 + g], sb = SB[r * 4 + g];     // byte 0 of the scale VGPR: block scale of (row r, K block g)
  v4f acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 0, sa, 0, sb);
  // standard 16x16 C/D map: col = lane & 15, row = (lane >> 4) * 4 + reg
  for (int q = 0; q < 4; ++q) C[(g * 4 + q) * 16 + r] = acc[q];
}
int main() {
  std::vector<unsigned char> A(16 * 128), B(16 * 128), SA(64), SB(64);
  srand(1);
  for (auto& x : A) { do x = rand() & 255; while ((x & 0x7f) == 0x7f); }
  for (auto& x : B) { do x = rand() & 255; while ((x & 0x7f) == 0x7f); }
  for (auto& x : SA) x = 127 + (rand() % 5) - 2;
  for (auto& x : SB) x = 127 + (rand() % 5) - 2;
  unsigned char *dA, *dB, *dSA, *dSB; float* dC;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dSA, 64); hipMalloc(&dSB, 64); hipMalloc(&dC, 1024);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  hipMemcpy(dSA, SA.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dSB, SB.data(), 64, hipMemcpyHostToDevice);
  for (int pass = 0; pass < 2; ++pass) {
  if (pass == 1) {   // unit scales: what is left is the instruction's own accumulation error
    for (auto& x : SA) x = 127; for (auto& x : SB) x = 127;
    hipMemcpy(dSA, SA.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dSB, SB.data(), 64, hipMemcpyHostToDevice);
    printf("-- all block scales = 2^0 --\n");
  }
  for (int hyp = 0; hyp < 2; ++hyp) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dC, hyp);
    float C[256]; hipMemcpy(C, dC, 1024, hipMemcpyDeviceToHost);
    // which operand is "rows" of C?  try C[i][j] = sum_k A[i][k] B[j][k] and the transpose, with per-(row, 32-block) scales
    for (int tr = 0; tr < 2; ++tr) {
      double worst = 0, mag = 0;
      for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0;
        for (int kk = 0; kk < 128; ++kk)
          s += (double)e4m3(A[i * 128 + kk]) * ldexp(1.0, SA[i * 4 + kk / 32] - 127) * e4m3(B[j * 128 + kk]) * ldexp(1.0, SB[j * 4 + kk / 32] - 127);
        const double got = tr ? C[j * 16 + i] : C[i * 16 + j];
        worst = fmax(worst, fabs(got - s)); mag = fmax(mag, fabs(s));
      }
      printf("hyp %d (K %s), C %s: max err %.4g (max |C| %.4g)\n", hyp, hyp ? "16+16 split" : "32 contiguous", tr ? "= (A B^T)^T" : "= A B^T", worst, mag);
    }
  }
  }
  return 0;
}
