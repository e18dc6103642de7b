// On-device evaluation metrics - the reference computes them on the host:
//   depth : /root/reference/metrics/eval_depth.py:6-246 as called by eval.py:49 (mask 0 < gt < 80, least-squares
//           scale/shift of metrics/alignment.py:150-167, AbsRel / SqRel / RMSE / LogRMSE / delta thresholds on the
//           custom mask)
//   normal: /root/reference/metrics/eval_normal.py:4-34 (angular error in degrees; mean / median / rmse / % under
//           5, 7.5, 11.25, 22.5, 30 degrees)
// All reductions are two-stage with a fixed order (per-block fp64 partials, summed by the host in block order), the
// median is an exact two-pass histogram selection with integer atomics: results are deterministic.
#include "../common.h"

#define MB 256   // threads per block
#define MAXB 1024

__device__ __forceinline__ double block_sum(double v, double* sh) {
  const int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int o = MB / 2; o > 0; o >>= 1) {
    if (tid < o) sh[tid] += sh[tid + o];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

// pass 1: normal-equation sums over mask1 = (0 < gt < max_depth): n, sum p, sum p^2, sum g, sum p*g
__global__ __launch_bounds__(MB) void k_depth_fit(const float* pred, const float* gt, long n, float max_depth, double* part) {
  __shared__ double sh[MB];
  double a[5] = {0, 0, 0, 0, 0};
  for (long i = (long)blockIdx.x * MB + threadIdx.x; i < n; i += (long)gridDim.x * MB) {
    const float g = gt[i];
    if (g > 0.f && g < max_depth) {
      const double p = pred[i];
      a[0] += 1.0; a[1] += p; a[2] += p * p; a[3] += g; a[4] += p * (double)g;
    }
  }
  for (int k = 0; k < 5; ++k) {
    const double r = block_sum(a[k], sh);
    if (threadIdx.x == 0) part[(long)blockIdx.x * 5 + k] = r;
  }
}

// pass 2: metrics of p' = s*p + t on mask1 & custom mask
__global__ __launch_bounds__(MB) void k_depth_metrics(const float* pred, const float* gt, const unsigned char* cmask, long n,
                                                      float max_depth, float s, float t, double* part) {
  __shared__ double sh[MB];
  double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // n, absrel, sqrel, sq, logsq, d1, d125, d125^2, d125^3
  for (long i = (long)blockIdx.x * MB + threadIdx.x; i < n; i += (long)gridDim.x * MB) {
    const float g = gt[i];
    if (g > 0.f && g < max_depth && (!cmask || cmask[i])) {
      const float p = s * pred[i] + t;
      const float d = p - g;
      a[0] += 1.0; a[1] += fabsf(d) / g; a[2] += d * d / g; a[3] += d * d;
      const float pc = fmaxf(p, 1e-5f);
      const float l = logf(pc) - logf(g);
      a[4] += l * l;
      const float r = fmaxf(pc / g, g / pc);
      a[5] += r < 1.0f; a[6] += r < 1.25f; a[7] += r < 1.5625f; a[8] += r < 1.953125f;
    }
  }
  for (int k = 0; k < 9; ++k) {
    const double r = block_sum(a[k], sh);
    if (threadIdx.x == 0) part[(long)blockIdx.x * 9 + k] = r;
  }
}

// angular error per pixel (degrees), -1 for masked pixels; partial sums n, sum e, sum e^2, counts under 5 thresholds;
// coarse histogram (4096 bins over [0,180]) for the median
__global__ __launch_bounds__(MB) void k_normal_err(const float* pn, const float* gn, const unsigned char* mask, long n, float* err,
                                                   double* part, unsigned* hist) {
  __shared__ double sh[MB];
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long i = (long)blockIdx.x * MB + threadIdx.x; i < n; i += (long)gridDim.x * MB) {
    float e = -1.f;
    if (!mask || mask[i]) {
      const float px = pn[i * 3], py = pn[i * 3 + 1], pz = pn[i * 3 + 2];
      const float gx = gn[i * 3], gy = gn[i * 3 + 1], gz = gn[i * 3 + 2];
      const float dot = px * gx + py * gy + pz * gz;
      const float na = sqrtf(px * px + py * py + pz * pz), nb = sqrtf(gx * gx + gy * gy + gz * gz);
      float c = dot / (na * nb + 1e-6f);
      c = fminf(fmaxf(c, -1.f), 1.f);
      e = acosf(c) * 57.29577951308232f;
      a[0] += 1.0; a[1] += e; a[2] += (double)e * e;
      a[3] += e < 5.f; a[4] += e < 7.5f; a[5] += e < 11.25f; a[6] += e < 22.5f; a[7] += e < 30.f;
      int bin = (int)(e * (4096.0f / 180.0f)); bin = bin < 0 ? 0 : (bin > 4095 ? 4095 : bin);
      atomicAdd(&hist[bin], 1u);
    }
    err[i] = e;
  }
  for (int k = 0; k < 8; ++k) {
    const double r = block_sum(a[k], sh);
    if (threadIdx.x == 0) part[(long)blockIdx.x * 8 + k] = r;
  }
}

// collect the errors that fall into one histogram bin (the bin holding the median)
__global__ void k_collect_bin(const float* err, long n, int bin, float* out, unsigned* count, unsigned cap) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float e = err[i];
    if (e < 0.f) continue;
    int b = (int)(e * (4096.0f / 180.0f)); b = b < 0 ? 0 : (b > 4095 ? 4095 : b);
    if (b == bin) { const unsigned k = atomicAdd(count, 1u); if (k < cap) out[k] = e; }
  }
}

static int nblocks(long n) { long b = (n + MB - 1) / MB; return (int)(b > MAXB ? MAXB : (b < 1 ? 1 : b)); }

void launch_depth_fit(const float* pred, const float* gt, long n, float max_depth, double* part, int* nb, hipStream_t s) {
  *nb = nblocks(n);
  hipLaunchKernelGGL(k_depth_fit, dim3(*nb), dim3(MB), 0, s, pred, gt, n, max_depth, part);
}
void launch_depth_metrics(const float* pred, const float* gt, const unsigned char* cmask, long n, float max_depth, float sc,
                          float sh, double* part, int* nb, hipStream_t s) {
  *nb = nblocks(n);
  hipLaunchKernelGGL(k_depth_metrics, dim3(*nb), dim3(MB), 0, s, pred, gt, cmask, n, max_depth, sc, sh, part);
}
void launch_normal_err(const float* pn, const float* gn, const unsigned char* mask, long n, float* err, double* part,
                       unsigned* hist, int* nb, hipStream_t s) {
  *nb = nblocks(n);
  hipLaunchKernelGGL(k_normal_err, dim3(*nb), dim3(MB), 0, s, pn, gn, mask, n, err, part, hist);
}
void launch_collect_bin(const float* err, long n, int bin, float* out, unsigned* count, unsigned cap, hipStream_t s) {
  hipLaunchKernelGGL(k_collect_bin, dim3(nblocks(n)), dim3(MB), 0, s, err, n, bin, out, count, cap);
}
