// Small memory-bound kernels: casts, weight re-layouts, the DepthCrafter pipeline glue
// (input preparation, CLIP pre-processing, Euler scheduler step, output post-processing) and
// the depth -> surface-normal kernel.  All grid-stride / one element group per thread.
#include "../common.h"

#define GS_LOOP(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)
static inline dim3 gs_grid(long n, int block = 256) {
  long g = (n + block - 1) / block;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return dim3((unsigned)g);
}

__global__ void k_cast_f32_f16(const float* in, f16* out, long n) { GS_LOOP(i, n) out[i] = (f16)in[i]; }
__global__ void k_cast_f16_f32(const f16* in, float* out, long n) { GS_LOOP(i, n) out[i] = (float)in[i]; }
void launch_cast_f32_f16(const float* in, f16* out, long n, hipStream_t s) {
  hipLaunchKernelGGL(k_cast_f32_f16, gs_grid(n), dim3(256), 0, s, in, out, n);
}
void launch_cast_f16_f32(const f16* in, float* out, long n, hipStream_t s) {
  hipLaunchKernelGGL(k_cast_f16_f32, gs_grid(n), dim3(256), 0, s, in, out, n);
}

// [O][I][taps] -> [Opad][taps][Ipad]
__global__ void k_permute_conv_w(const f16* in, f16* out, int O, int I, int taps, int Ipad, int Opad) {
  const long n = (long)Opad * taps * Ipad;
  GS_LOOP(idx, n) {
    const int i = idx % Ipad;
    const int tp = (idx / Ipad) % taps;
    const int o = idx / ((long)Ipad * taps);
    out[idx] = (o < O && i < I) ? in[((long)o * I + i) * taps + tp] : (f16)0.f;
  }
}
void launch_permute_conv_w(const f16* in, f16* out, int O, int I, int taps, int Ipad, int Opad, hipStream_t s) {
  hipLaunchKernelGGL(k_permute_conv_w, gs_grid((long)Opad * taps * Ipad), dim3(256), 0, s, in, out, O, I, taps, Ipad, Opad);
}

// [O][taps][Ipad] -> [O][Ipad/64][taps][64]: the chunk-major K order of the implicit-GEMM kernels (GemmP::kchunk)
__global__ void k_rechunk_conv_w(const f16* in, f16* out, long O, int taps, int Ipad) {
  const long n = O * taps * Ipad;
  GS_LOOP(idx, n) {
    const int c = idx % 64;
    const int tp = (idx / 64) % taps;
    const int chk = (idx / (64L * taps)) % (Ipad / 64);
    const long o = idx / ((long)Ipad * taps);
    out[idx] = in[(o * taps + tp) * Ipad + chk * 64 + c];
  }
}
void launch_rechunk_conv_w(const f16* in, f16* out, long O, int taps, int Ipad, hipStream_t s) {
  hipLaunchKernelGGL(k_rechunk_conv_w, gs_grid(O * taps * Ipad), dim3(256), 0, s, in, out, O, taps, Ipad);
}

// nearest-2x upsample followed by a zero-padded 3x3 conv == four 2x2 convs on the source grid, one per output parity
// (a,b): rows 2y-1,2y,2y+1 of the upsampled image are source rows y-1,y,y (a=0) or y,y,y+1 (a=1), so the 3 taps
// collapse to 2 with weights {w0, w1+w2} or {w0+w1, w2}; same along x.  Sums are formed in fp32 and rounded once.
__global__ void k_upsample_phase_w(const f16* w9, f16* w4, int O, int I, int Ipad) {
  const long n = (long)4 * O * 4 * Ipad;
  GS_LOOP(idx, n) {
    const int i = idx % Ipad;
    const int tap = (idx / Ipad) % 4;
    const int o = (idx / ((long)Ipad * 4)) % O;
    const int ph = idx / ((long)Ipad * 4 * O);
    const int a = ph >> 1, b = ph & 1, pp = tap >> 1, q = tap & 1;
    float acc = 0.f;
    if (i < I)
      for (int iy = 0; iy < 3; ++iy)
        for (int ix = 0; ix < 3; ++ix) {
          const int sy = a == 0 ? (iy == 0 ? 0 : 1) : (iy == 2 ? 1 : 0);
          const int sx = b == 0 ? (ix == 0 ? 0 : 1) : (ix == 2 ? 1 : 0);
          if (sy == pp && sx == q) acc += (float)w9[(((long)o * I + i) * 3 + iy) * 3 + ix];
        }
    w4[idx] = (f16)acc;
  }
}
void launch_upsample_phase_w(const f16* w9, f16* w4, int O, int I, int Ipad, hipStream_t s) {
  hipLaunchKernelGGL(k_upsample_phase_w, gs_grid((long)16 * O * Ipad), dim3(256), 0, s, w9, w4, O, I, Ipad);
}

__global__ void k_gather_rows(const f16* in, f16* out, const int* rowmap, int rows, int cols) {
  const long n = (long)rows * cols;
  GS_LOOP(idx, n) {
    const int r = idx / cols, c = idx - (long)r * cols;
    out[idx] = in[(long)rowmap[r] * cols + c];
  }
}
void launch_gather_rows(const f16* in, f16* out, const int* rowmap, int rows, int cols, hipStream_t s) {
  hipLaunchKernelGGL(k_gather_rows, gs_grid((long)rows * cols), dim3(256), 0, s, in, out, rowmap, rows, cols);
}

__global__ void k_copy2d(const f16* in, long ldi, f16* out, long ldo, long rows, int cols) {
  const long n = rows * cols;
  GS_LOOP(idx, n) {
    const long r = idx / cols; const int c = idx - r * cols;
    out[r * ldo + c] = in[r * ldi + c];
  }
}
void launch_copy2d(const f16* in, long ldi, f16* out, long ldo, long rows, int cols, hipStream_t s) {
  hipLaunchKernelGGL(k_copy2d, gs_grid(rows * cols), dim3(256), 0, s, in, ldi, out, ldo, rows, cols);
}

__global__ void k_fill(f16* p, float v, long n) { GS_LOOP(i, n) p[i] = (f16)v; }
void launch_fill_f16(f16* p, float v, long n, hipStream_t s) {
  hipLaunchKernelGGL(k_fill, gs_grid(n), dim3(256), 0, s, p, v, n);
}

__global__ void k_fill_random(f16* p, long n, unsigned seed) {
  GS_LOOP(i, n) {
    unsigned h = (unsigned)i * 2654435761u ^ (seed * 0x9E3779B9u);
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    p[i] = (f16)((float)(h & 0xffff) / 32768.0f - 1.0f);
  }
}
void launch_fill_random(f16* p, long n, unsigned seed, hipStream_t s) {
  hipLaunchKernelGGL(k_fill_random, gs_grid(n), dim3(256), 0, s, p, n, seed);
}

__global__ void k_add_rowvec(const f16* x, const f16* vec, f16* y, long M, int C, int rpv) {
  const long n = M * (C / 8);
  const int nv = C / 8;
  GS_LOOP(idx, n) {
    const long m = idx / nv; const int v = idx - m * nv;
    const f16x8 a = *(const f16x8*)(x + m * C + v * 8);
    const f16x8 b = *(const f16x8*)(vec + (m / rpv) * C + v * 8);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)((float)a[e] + (float)b[e]);
    *(f16x8*)(y + m * C + v * 8) = o;
  }
}
void launch_add_rowvec(const f16* x, const f16* vec, f16* y, long M, int C, int rows_per_vec, hipStream_t s) {
  hipLaunchKernelGGL(k_add_rowvec, gs_grid(M * (C / 8)), dim3(256), 0, s, x, vec, y, M, C, rows_per_vec);
}

__global__ void k_axpby(const f16* a, const f16* b, f16* y, float ca, float cb, long n) {
  GS_LOOP(i, n) y[i] = (f16)(ca * (float)a[i] + (b ? cb * (float)b[i] : 0.f));
}
void launch_axpby(const f16* a, const f16* b, f16* y, float ca, float cb, long n, hipStream_t s) {
  hipLaunchKernelGGL(k_axpby, gs_grid(n), dim3(256), 0, s, a, b, y, ca, cb, n);
}

// ------------------------------------------------------------------------------------------
// DepthCrafter pipeline glue.  The fp16 round-offs of the reference pipeline (it runs in
// torch.float16) are reproduced step by step: x.half(); x*2-1; noise_aug*noise; sum.
// ------------------------------------------------------------------------------------------
// frames f32 [T,H,W,3] in [0,1]; noise f32 [T,3,H,W] -> clip_src f16 [T,H,W,3] in [-1,1],
// vae_in f16 [T,H,W,8] (channels 3..7 zero).
__global__ void k_prep_video(const float* frames, const float* noise, f16* clip_src, f16* vae_in, int T, long HW,
                             float aug) {
  const long n = (long)T * HW;
  GS_LOOP(pix, n) {
    const long t = pix / HW, r = pix - t * HW;
    f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const f16 x = (f16)frames[pix * 3 + c];
      const f16 v = (f16)((float)(f16)((float)x * 2.0f) - 1.0f);
      clip_src[pix * 3 + c] = v;
      const f16 nz = (f16)noise[(t * 3 + c) * HW + r];
      const f16 an = (f16)(aug * (float)nz);
      o[c] = (f16)((float)v + (float)an);
    }
    *(f16x8*)(vae_in + pix * 8) = o;
  }
}
void launch_prep_video(const float* frames, const float* noise, f16* clip_src, f16* vae_in, int T, int H, int W,
                       float noise_aug, hipStream_t s) {
  hipLaunchKernelGGL(k_prep_video, gs_grid((long)T * H * W), dim3(256), 0, s, frames, noise, clip_src, vae_in, T,
                     (long)H * W, noise_aug);
}

// CLIP pre-processing fused: separable gaussian pre-blur (reflect) -> bicubic (A=-0.75,
// align_corners) to S224 x S224 -> (v+1)/2 -> CLIP mean/std -> fp16 patch matrix
// [T * (S224/P)^2, Kpad], k = c*P*P + py*P + px.
struct BlurTaps { int ky, kx; float gy[9], gx[9]; };
__device__ __forceinline__ int reflect_idx(int i, int n) { if (i < 0) i = -i; if (i >= n) i = 2 * n - 2 - i; return i; }
__device__ __forceinline__ void cubic_w(float t, float (&w)[4]) {
  const float A = -0.75f;
  float x = t + 1.f; w[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
  x = t;             w[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 1.f - t;       w[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 2.f - t;       w[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}
struct Norm3 { float mean[3], stdv[3]; };
__global__ void k_clip_patchify(const f16* src, f16* patches, int T, int H, int W, int S224, int P, int Kpad,
                                BlurTaps bt, Norm3 nm) {
  const long n = (long)T * S224 * S224;
  const int np = S224 / P;
  const float* mean = nm.mean; const float* stdv = nm.stdv;
  GS_LOOP(idx, n) {
    const int ox = idx % S224, oy = (idx / S224) % S224;
    const long t = idx / ((long)S224 * S224);
    // torch: scale = (in-1)/(out-1) (align_corners); src = scale * dst
    const float sy = (S224 > 1) ? ((float)(H - 1) / (float)(S224 - 1)) * oy : 0.f;
    const float sx = (S224 > 1) ? ((float)(W - 1) / (float)(S224 - 1)) * ox : 0.f;
    const int iy = (int)floorf(sy), ix = (int)floorf(sx);
    float wy[4], wx[4];
    cubic_w(sy - iy, wy); cubic_w(sx - ix, wx);
    float acc[3] = {0.f, 0.f, 0.f};
    for (int a = 0; a < 4; ++a) {
      const int by = min(max(iy - 1 + a, 0), H - 1);       // bicubic border: clamp
      for (int b = 0; b < 4; ++b) {
        const int bx = min(max(ix - 1 + b, 0), W - 1);
        float bl[3] = {0.f, 0.f, 0.f};                      // blurred value at (by,bx)
        for (int u = 0; u < bt.ky; ++u) {
          const int yy = reflect_idx(by + u - bt.ky / 2, H);
          float row[3] = {0.f, 0.f, 0.f};
          for (int v = 0; v < bt.kx; ++v) {
            const int xx = reflect_idx(bx + v - bt.kx / 2, W);
            const f16* px = src + ((t * H + yy) * W + xx) * 3;
            row[0] += bt.gx[v] * (float)px[0]; row[1] += bt.gx[v] * (float)px[1]; row[2] += bt.gx[v] * (float)px[2];
          }
          bl[0] += bt.gy[u] * row[0]; bl[1] += bt.gy[u] * row[1]; bl[2] += bt.gy[u] * row[2];
        }
        const float w = wy[a] * wx[b];
        acc[0] += w * bl[0]; acc[1] += w * bl[1]; acc[2] += w * bl[2];
      }
    }
    const int pr = oy / P, pcx = ox / P, py = oy % P, px = ox % P;
    f16* dst = patches + ((t * np + pr) * np + pcx) * (long)Kpad;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = ((acc[c] + 1.0f) * 0.5f - mean[c]) / stdv[c];
      dst[c * P * P + py * P + px] = (f16)v;
    }
  }
}
static void gauss_taps(int src, int dst, int& ks, float* g) {
  double sigma = ((double)src / dst - 1.0) / 2.0;
  if (sigma < 0.001) sigma = 0.001;
  ks = (int)(2.0 * 2 * sigma); if (ks < 3) ks = 3;
  if (ks % 2 == 0) ks += 1;
  UG_REQUIRE(ks <= 9, "antialias kernel too large (downscale factor > ~5)");
  float sigf = (float)sigma, sum = 0.f;
  for (int i = 0; i < ks; ++i) { float x = (float)(i - ks / 2); g[i] = expf(-(x * x) / (2.f * sigf * sigf)); sum += g[i]; }
  for (int i = 0; i < ks; ++i) g[i] /= sum;
}
void launch_clip_patchify(const f16* video_m11, f16* patches, int T, int H, int W, int S224, int P, int Kpad,
                          hipStream_t s, int imagenet_norm) {
  const Norm3 nm = imagenet_norm ? Norm3{{0.485f, 0.456f, 0.406f}, {0.229f, 0.224f, 0.225f}}
                                 : Norm3{{0.48145466f, 0.4578275f, 0.40821073f}, {0.26862954f, 0.26130258f, 0.27577711f}};
  BlurTaps bt;
  gauss_taps(H, S224, bt.ky, bt.gy);
  gauss_taps(W, S224, bt.kx, bt.gx);
  hipLaunchKernelGGL(k_clip_patchify, gs_grid((long)T * S224 * S224), dim3(256), 0, s, video_m11, patches, T, H, W,
                     S224, P, Kpad, bt, nm);
}

// ---- StableNormal glue ----
__global__ void k_scale_rows(f16* w, const f16* gamma, long n, int K) {
  GS_LOOP(i, n) { w[i] = (f16)((float)w[i] * (float)gamma[i / K]); }
}
void launch_scale_rows(f16* w, const f16* gamma, int N, int K, hipStream_t s) {
  hipLaunchKernelGGL(k_scale_rows, gs_grid((long)N * K), dim3(256), 0, s, w, gamma, (long)N * K, K);
}
__global__ void k_add_grid_nearest(f16* x, const f16* grid, int B, int h, int w, int g, int C) {
  const long n = (long)B * h * w * (C / 8);
  GS_LOOP(i, n) {
    const int cv = i % (C / 8); const long pix = i / (C / 8);
    const int xx = pix % w, yy = (pix / w) % h; const long b = pix / ((long)w * h);
    const f16x8 a = *(const f16x8*)(x + pix * C + cv * 8);
    const f16x8 gvec = *(const f16x8*)(grid + ((b * g + (yy * g) / h) * g + (xx * g) / w) * C + cv * 8);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)((float)a[e] + (float)gvec[e]);
    *(f16x8*)(x + pix * C + cv * 8) = o;
  }
}
void launch_add_grid_nearest(f16* x, const f16* grid, int B, int h, int w, int g, int C, hipStream_t s) {
  hipLaunchKernelGGL(k_add_grid_nearest, gs_grid((long)B * h * w * (C / 8)), dim3(256), 0, s, x, grid, B, h, w, g, C);
}
__global__ void k_sn_normals_out(const f16* dec, int ldd, float* out, long pixels) {
  GS_LOOP(p, pixels) {
    float v[3]; float n2 = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) { v[c] = fminf(fmaxf((float)dec[p * ldd + c], -1.f), 1.f); n2 += v[c] * v[c]; }
    const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-6f);
#pragma unroll
    for (int c = 0; c < 3; ++c) out[p * 3 + c] = v[c] * inv;
  }
}
// Antialiased bilinear resize of a channels-last f32 image batch [B,Hi,Wi,C] -> [B,Ho,Wo,C]: torch's F.interpolate(mode="bilinear",
// align_corners=False, antialias=True) restated - per axis a triangle filter centred at scale * (o + 0.5) with support max(scale, 1), taps
// [int(centre - support + 0.5), int(centre + support + 0.5)) clipped to the image, weights normalised (up-sampling: plain bilinear).
// normalise = 1: L2-normalise the C channels of every output pixel (resizing unit normals back to the input size).
__global__ void k_resize_bilinear_aa(const float* in, float* out, int B, int Hi, int Wi, int Ho, int Wo, int C, int normalise) {
  const float sy = (float)Hi / Ho, sx = (float)Wi / Wo;
  const float supy = sy >= 1.f ? sy : 1.f, supx = sx >= 1.f ? sx : 1.f;
  const float ivy = sy >= 1.f ? 1.f / sy : 1.f, ivx = sx >= 1.f ? 1.f / sx : 1.f;
  const long n = (long)B * Ho * Wo;
  GS_LOOP(idx, n) {
    const int ox = (int)(idx % Wo); const int oy = (int)((idx / Wo) % Ho); const int b = (int)(idx / ((long)Wo * Ho));
    const float cy = sy * (oy + 0.5f), cx = sx * (ox + 0.5f);
    const int y0 = max(0, (int)(cy - supy + 0.5f)), y1 = min(Hi, (int)(cy + supy + 0.5f));
    const int x0 = max(0, (int)(cx - supx + 0.5f)), x1 = min(Wi, (int)(cx + supx + 0.5f));
    float wys = 0.f, wxs = 0.f;
    for (int y = y0; y < y1; ++y) wys += fmaxf(0.f, 1.f - fabsf((y - cy + 0.5f) * ivy));
    for (int x = x0; x < x1; ++x) wxs += fmaxf(0.f, 1.f - fabsf((x - cx + 0.5f) * ivx));
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int y = y0; y < y1; ++y) {
      const float wy = fmaxf(0.f, 1.f - fabsf((y - cy + 0.5f) * ivy)) / wys;
      float row[4] = {0.f, 0.f, 0.f, 0.f};
      for (int x = x0; x < x1; ++x) {
        const float wx = fmaxf(0.f, 1.f - fabsf((x - cx + 0.5f) * ivx)) / wxs;
        const float* px = in + (((long)b * Hi + y) * Wi + x) * C;
        for (int c = 0; c < C; ++c) row[c] += wx * px[c];
      }
      for (int c = 0; c < C; ++c) acc[c] += wy * row[c];
    }
    if (normalise) {
      float q = 0.f;
      for (int c = 0; c < C; ++c) q += acc[c] * acc[c];
      const float r = rsqrtf(fmaxf(q, 1e-24f));
      for (int c = 0; c < C; ++c) acc[c] *= r;
    }
    for (int c = 0; c < C; ++c) out[idx * C + c] = acc[c];
  }
}
void launch_resize_bilinear_aa(const float* in, float* out, int B, int Hi, int Wi, int Ho, int Wo, int C, int normalise, hipStream_t s) {
  UG_REQUIRE(C >= 1 && C <= 4, "resize: 1..4 channels");
  hipLaunchKernelGGL(k_resize_bilinear_aa, gs_grid((long)B * Ho * Wo), dim3(256), 0, s, in, out, B, Hi, Wi, Ho, Wo, C, normalise);
}

void launch_sn_normals_out(const f16* dec, int ldd, float* out, long pixels, hipStream_t s) {
  hipLaunchKernelGGL(k_sn_normals_out, gs_grid(pixels), dim3(256), 0, s, dec, ldd, out, pixels);
}

// noise f32 [T,4,h,w] (NCHW, as torch.randn would lay it out) -> latents f16 [T,h,w,4] * sigma0
__global__ void k_init_latents(const float* noise, f16* lat, float sigma0, int T, long hw) {
  const long n = (long)T * hw * 4;
  GS_LOOP(idx, n) {
    const int c = idx & 3; const long pix = idx >> 2;
    const long t = pix / hw, r = pix - t * hw;
    lat[idx] = (f16)((float)(f16)noise[(t * 4 + c) * hw + r] * sigma0);
  }
}
void launch_init_latents2(const float* noise, f16* lat, float sigma0, int T, long hw, hipStream_t s) {
  hipLaunchKernelGGL(k_init_latents, gs_grid((long)T * hw * 4), dim3(256), 0, s, noise, lat, sigma0, T, hw);
}

__global__ void k_make_unet_input(const f16* lat, const f16* cond, f16* x, long pixels, float inv_scale_den) {
  GS_LOOP(p, pixels) {
    const f16x4 l = *(const f16x4*)(lat + p * 4);
    const f16x4 c = *(const f16x4*)(cond + p * 4);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = (f16)((float)l[e] / inv_scale_den); o[4 + e] = c[e]; }
    *(f16x8*)(x + p * 8) = o;
  }
}
void launch_make_unet_input(const f16* lat, const f16* cond, f16* x, long pixels, float den, hipStream_t s) {
  hipLaunchKernelGGL(k_make_unet_input, gs_grid(pixels), dim3(256), 0, s, lat, cond, x, pixels, den);
}

// Euler step, v-prediction (diffusers EulerDiscreteScheduler.step with s_churn = 0):
//   x0 = v * (-sigma / sqrt(sigma^2+1)) + x / (sigma^2+1);  x <- x + (x - x0)/sigma * (sigma_next - sigma)
// the product v*c is rounded to fp16 first (v stays fp16 in the reference while x is upcast).
__global__ void k_euler_step(const f16* v, f16* lat, long n, float sigma, float sigma_next) {
  const float c = -sigma / sqrtf(sigma * sigma + 1.f);
  const float d2 = sigma * sigma + 1.f;
  const float dt = sigma_next - sigma;
  GS_LOOP(i, n) {
    const float x = (float)lat[i];
    const float vc = (float)(f16)((float)v[i] * c);
    const float x0 = vc + x / d2;
    const float der = (x - x0) / sigma;
    lat[i] = (f16)(x + der * dt);
  }
}
void launch_euler_step(const f16* v, f16* lat, long n, float sigma, float sigma_next, hipStream_t s) {
  hipLaunchKernelGGL(k_euler_step, gs_grid(n), dim3(256), 0, s, v, lat, n, sigma, sigma_next);
}

__global__ void k_silu(const f16* in, f16* out, long n) {
  GS_LOOP(i, n) { const float x = (float)in[i]; out[i] = (f16)(x * __builtin_amdgcn_rcpf(1.0f + __expf(-x))); }
}
void launch_silu_f16(const f16* in, f16* out, long n, hipStream_t s) {
  hipLaunchKernelGGL(k_silu, gs_grid(n), dim3(256), 0, s, in, out, n);
}

__global__ void k_clip_assemble(const f16* patches, const f16* cls, const f16* pos, f16* tok, int T, int np, int d) {
  const long n = (long)T * (np + 1) * d;
  GS_LOOP(idx, n) {
    const int c = idx % d; const long r = idx / d;
    const int s = r % (np + 1); const long t = r / (np + 1);
    const float base = s == 0 ? (float)cls[c] : (float)patches[(t * np + (s - 1)) * d + c];
    tok[idx] = (f16)(base + (float)pos[(long)s * d + c]);
  }
}
void launch_clip_assemble(const f16* patches, const f16* cls, const f16* pos, f16* tok, int T, int np, int d,
                          hipStream_t s) {
  hipLaunchKernelGGL(k_clip_assemble, gs_grid((long)T * (np + 1) * d), dim3(256), 0, s, patches, cls, pos, tok, T, np, d);
}

__global__ void k_scale(const f16* in, f16* out, float sc, long n) { GS_LOOP(i, n) out[i] = (f16)((float)in[i] * sc); }
void launch_scale_f16(const f16* in, f16* out, float sc, long n, hipStream_t s) {
  hipLaunchKernelGGL(k_scale, gs_grid(n), dim3(256), 0, s, in, out, sc, n);
}

// out = a * x + b * y (fp16 storage, fp32 arithmetic); latent sliding windows: re-noising of the overlap frames
__global__ void k_axpby(const f16* x, float a, const f16* y, float b, f16* out, long n) {
  GS_LOOP(i, n) out[i] = (f16)(a * (float)x[i] + b * (float)y[i]);
}
void launch_axpby_f16(const f16* x, float a, const f16* y, float b, f16* out, long n, hipStream_t s) {
  hipLaunchKernelGGL(k_axpby, gs_grid(n), dim3(256), 0, s, x, a, y, b, out, n);
}
// linear cross-fade over `overlap` frames: all[f] = cur[f] * w_f + all[f] * (1 - w_f), w_f = f / (overlap - 1)
__global__ void k_crossfade(const f16* cur, f16* all, long frame_elems, int overlap) {
  const long n = frame_elems * overlap;
  GS_LOOP(i, n) {
    const int f = (int)(i / frame_elems);
    const float w = overlap > 1 ? (float)f / (float)(overlap - 1) : 0.f;
    all[i] = (f16)((float)cur[i] * w + (float)all[i] * (1.f - w));
  }
}
void launch_crossfade_f16(const f16* cur, f16* all, long frame_elems, int overlap, hipStream_t s) {
  hipLaunchKernelGGL(k_crossfade, gs_grid(frame_elems * overlap), dim3(256), 0, s, cur, all, frame_elems, overlap);
}

__global__ void k_pad_channels(const f16* in, int Cin, f16* out, int Cout, long pixels) {
  const long n = pixels * Cout;
  GS_LOOP(idx, n) {
    const long p = idx / Cout; const int c = idx - p * Cout;
    out[idx] = c < Cin ? in[p * Cin + c] : (f16)0.f;
  }
}
void launch_pad_channels(const f16* in, int Cin, f16* out, int Cout, long pixels, hipStream_t s) {
  hipLaunchKernelGGL(k_pad_channels, gs_grid(pixels * Cout), dim3(256), 0, s, in, Cin, out, Cout, pixels);
}

// decoder tail: Conv3d(3,3,(3,1,1)) over frames (zero padded in t) -> fp16 -> (x/2+0.5).clamp(0,1) f32
// x [T,HW,Cs] (first 3 channels valid), w [3 out][3 in][3 kt], b [3]; out f32 [T,HW,3]
__global__ void k_time_conv_out(const f16* x, const f16* w, const f16* b, float* out, int T, long HW, int Cs) {
  const long n = (long)T * HW;
  float wf[27], bf[3];
#pragma unroll
  for (int i = 0; i < 27; ++i) wf[i] = (float)w[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) bf[i] = (float)b[i];
  GS_LOOP(idx, n) {
    const long t = idx / HW, r = idx - t * HW;
    float acc[3] = {bf[0], bf[1], bf[2]};
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const long tt = t + kt - 1;
      if (tt < 0 || tt >= T) continue;
      const f16* px = x + (tt * HW + r) * Cs;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {
        const float v = (float)px[ci];
#pragma unroll
        for (int co = 0; co < 3; ++co) acc[co] += wf[(co * 3 + ci) * 3 + kt] * v;
      }
    }
#pragma unroll
    for (int co = 0; co < 3; ++co) {
      const float h = (float)(f16)acc[co];
      out[idx * 3 + co] = fminf(fmaxf(h * 0.5f + 0.5f, 0.f), 1.f);
    }
  }
}
void launch_time_conv_out(const f16* x, const f16* w, const f16* b, float* frames_out, int T, long HW, int Cs,
                          hipStream_t s) {
  hipLaunchKernelGGL(k_time_conv_out, gs_grid((long)T * HW), dim3(256), 0, s, x, w, b, frames_out, T, HW, Cs);
}

// wrapper post-processing (reference model/depthcrafter.py:92-97)
__device__ __forceinline__ unsigned f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }
__global__ void k_depth_minmax(const float* frames, unsigned* mm, long pixels) {
  float lo = 3e38f, hi = -3e38f;
  GS_LOOP(p, pixels) {
    const float d = (frames[p * 3] + frames[p * 3 + 1] + frames[p * 3 + 2]) / 3.0f;
    lo = fminf(lo, d); hi = fmaxf(hi, d);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
  if ((threadIdx.x & 63) == 0) { atomicMin(&mm[0], f2ord(lo)); atomicMax(&mm[1], f2ord(hi)); }
}
__global__ void k_depth_apply(const float* frames, float* depth, const unsigned* mm, long pixels) {
  const float lo = ord2f(mm[0]), hi = ord2f(mm[1]);
  GS_LOOP(p, pixels) {
    const float d = (frames[p * 3] + frames[p * 3 + 1] + frames[p * 3 + 2]) / 3.0f;
    const float x = (d - lo) / (hi - lo);
    depth[p] = 1.0f / (x + 0.1f);
  }
}
__global__ void k_mm_init(unsigned* mm) { mm[0] = 0xffffffffu; mm[1] = 0u; }
void launch_depth_post(const float* frames, float* depth, float* minmax_ws, long pixels, hipStream_t s) {
  unsigned* mm = (unsigned*)minmax_ws;
  hipLaunchKernelGGL(k_mm_init, dim3(1), dim3(1), 0, s, mm);
  hipLaunchKernelGGL(k_depth_minmax, gs_grid(pixels), dim3(256), 0, s, frames, mm, pixels);
  hipLaunchKernelGGL(k_depth_apply, gs_grid(pixels), dim3(256), 0, s, frames, depth, (const unsigned*)mm, pixels);
}

// ------------------------------------------------------------------------------------------
// depth -> OpenGL-frame surface normals (reference model/depthcrafter.py:48-59 +
// utils/geometry_utils.py:9-70,246-253).  Per pixel: back-project, 5x5 zero-padded box sums of
// the 9 moment images, solve (A^T A + 1e-6 I) n = A^T 1, normalise, orient towards the camera,
// negate y,z.  The 3x3 symmetric solve is done in fp64 (adjugate); the moments are summed in
// fp64 as well, which is *more* accurate than the reference's fp32 conv2d + lstsq - parity is
// asserted on the angular error (tests/test_normals.py), not on raw floats.
// ------------------------------------------------------------------------------------------
__global__ void k_normals(const float* depth, const float* K33, float* normals, int T, int H, int W) {
  const long n = (long)T * H * W;
  GS_LOOP(idx, n) {
    const int x = idx % W, y = (idx / W) % H;
    const long t = idx / ((long)W * H);
    const float* K = K33 + t * 9;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double sxx = 0, syy = 0, szz = 0, sxy = 0, sxz = 0, syz = 0, sx = 0, sy = 0, sz = 0;
    for (int dy = -2; dy <= 2; ++dy) {
      const int yy = y + dy;
      if (yy < 0 || yy >= H) continue;
      for (int dx = -2; dx <= 2; ++dx) {
        const int xx = x + dx;
        if (xx < 0 || xx >= W) continue;
        const double z = depth[(t * H + yy) * W + xx];
        // reference: numpy promotes to f64, then .float()
        const double px = (double)(float)(((double)xx - cx) * z / fx);
        const double py = (double)(float)(((double)yy - cy) * z / fy);
        const double pz = (double)(float)z;
        sxx += px * px; syy += py * py; szz += pz * pz;
        sxy += px * py; sxz += px * pz; syz += py * pz;
        sx += px; sy += py; sz += pz;
      }
    }
    const double a = sxx + 1e-6, b = sxy, c = sxz, d = syy + 1e-6, e = syz, f = szz + 1e-6;
    // adjugate of the symmetric matrix [[a,b,c],[b,d,e],[c,e,f]]
    const double A00 = d * f - e * e, A01 = c * e - b * f, A02 = b * e - c * d;
    const double A11 = a * f - c * c, A12 = b * c - a * e, A22 = a * d - b * b;
    double nx = A00 * sx + A01 * sy + A02 * sz;
    double ny = A01 * sx + A11 * sy + A12 * sz;
    double nz = A02 * sx + A12 * sy + A22 * sz;
    const double det = a * A00 + b * A01 + c * A02;
    if (det < 0) { nx = -nx; ny = -ny; nz = -nz; }   // direction of A^-1 rhs = adj/det
    const double nn = sqrt(nx * nx + ny * ny + nz * nz);
    nx /= nn; ny /= nn; nz /= nn;
    const double z0 = depth[idx];
    const double p0x = (double)(float)(((double)x - cx) * z0 / fx), p0y = (double)(float)(((double)y - cy) * z0 / fy);
    if (nx * p0x + ny * p0y + nz * (double)(float)z0 > 0) { nx = -nx; ny = -ny; nz = -nz; }
    normals[idx * 3 + 0] = (float)nx;
    normals[idx * 3 + 1] = (float)(-ny);
    normals[idx * 3 + 2] = (float)(-nz);
  }
}
void launch_normals(const float* depth, const float* K33, float* normals, int T, int H, int W, hipStream_t s) {
  hipLaunchKernelGGL(k_normals, gs_grid((long)T * H * W), dim3(256), 0, s, depth, K33, normals, T, H, W);
}

