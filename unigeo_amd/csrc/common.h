// Shared declarations for the gfx950 (MI355X / CDNA4) DepthCrafter engine.
// Layout convention everywhere: activations are channels-last fp16, i.e. a [T,H,W,C] video
// tensor is the row-major matrix [M = T*H*W, C]; weights are [N][K] row-major ("B^T").
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdexcept>
#include <string>

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define UG_CHECK(expr)                                                                        \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      throw std::runtime_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) +    \
                               " at " + __FILE__ + ":" + std::to_string(__LINE__));           \
  } while (0)

#define UG_REQUIRE(cond, msg)                                                                 \
  do {                                                                                        \
    if (!(cond))                                                                              \
      throw std::runtime_error(std::string("requirement failed: ") + #cond + " : " + (msg) + \
                               " at " + __FILE__ + ":" + std::to_string(__LINE__));           \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
// hipFuncSetAttribute is per device: the "already raised the dynamic-LDS limit" caches are indexed by the current device
static inline int ug_dev_slot() { int d = 0; (void)hipGetDevice(&d); return d & 31; }

// ---------------------------------------------------------------------------------------
// GEMM / implicit-GEMM convolution  (kernels/gemm.hip)
//   Out[m,n] = act( c0 * f(acc[m,n] + bias[n] + bias2[n]) + c1*R1[m,n] + c2*R2[m,n] )
//   acc = sum_k A(m,k) * W[n,k];  A is either a dense row-major matrix or the implicit
//   im2col view of one/two channels-last tensors (3x3 / 1x1 / (3,1,1) taps, stride 1|2,
//   optional nearest-2x upsample of the source, optional channel concat of two sources).
// ---------------------------------------------------------------------------------------
enum { UG_ACT_NONE = 0, UG_ACT_SILU = 1, UG_ACT_GELU = 2 };
enum { UG_F_GEGLU = 1, UG_F_OUT_F32 = 2, UG_F_NOXCD = 4, UG_F_PRIO = 8 /* tuning knob: static priority for the younger half of an 8-wave workgroup */,
       UG_F_XCDROUND = 32 /* tuning knob 128: the round-strided XCD tile walk of rounds 1 - 4 (kernels/gemm_common.h: tile_walk) */,
       UG_F_R1_F32 = 16 /* R1 points at float32 (ldr1 in floats): the residual stream of the float32-grade VAE encoder, added in the epilogue instead of by a separate pass */ };

struct GemmP {
  const f16* A0; const f16* A1;
  int C0, C1;            // conv: channels of the two sources; dense: C0 = row stride (lda), C1 = 0
  int M, N, K;
  int conv;              // 0 dense, 1 implicit conv
  int T, Hi, Wi, Ho, Wo; // conv: frames, source dims (pre-upsample), output dims
  int ups, stride, pad_t, pad_l;
  int kt, ky, kx;        // taps along time / y / x
  const f16* W; long ldw;
  const f16* bias; const f16* bias2;
  const f16* R1; long ldr1; float c1;
  const f16* R2; long ldr2; float c2;
  float c0;
  void* Out; long ldo;
  int act, flags;
  const f16* zero;       // >=16 B of zeros in global memory (source for padded / OOB loads)
  int nb_inner;          // batch = gridDim.z = nb_outer * nb_inner
  long sA_o, sA_i, sW_o, sW_i, sO_o, sO_i;
  int cfg_p1;            // 0 = pick the tile config automatically, else tile config id + 1
  int splitk;            // 0 = automatic, 1 = off, >1 = K slices (needs `partial`)
  float* partial;        // split-K scratch: splitk * M * N floats
  void* trace;           // UG_GEMM_TRACE builds only: cycle-stamp dump (tools/gemm_trace.py)
  int up_phase;          // 0 = off; 1 + (a*2+b): conv output row (t,y,x) is stored at row (t, 2y+a, 2x+b) of a 2Ho x 2Wo grid
  // MX-fp8 dense path (launch_gemm_mx8): A0 / W point at OCP e4m3 bytes ([M][K] / [N][K], K % 128 == 0); sa / sw hold the e8m0 block
  // scales, four per dword (K blocks of 32 inside one 128-wide K step), K-step-major: sa[kstep * ld_sa + m], sw[kstep * ld_sw + n]
  const unsigned* sa; const unsigned* sw; long ld_sa, ld_sw;
  int kchunk;            // im2col K order: 0 = [tap][channel] (weights [N][taps][C]); 1 = [64-channel chunk][tap][64] - all taps of a chunk are
                         // consecutive K steps, so the activation rows a tile re-reads per tap are still in the XCD's L2 (needs (C0+C1) % 64 == 0)
  int tm_T, tm_nb;       // temporal conv (kt > 1): walk the M tiles frame-fastest - walk index i -> tile (i % tm_T) * tm_nb + i / tm_T, so the tiles of one
                         // pixel block in neighbouring frames (which read each other's rows as taps) run together; 0 = off (needs Ho*Wo % BM == 0)
  int m_off;             // im2col only, producer / consumer kernel only: this launch covers output rows [m_off, m_off + M) of the convolution
                         // (Out / R1 / R2 already point at row m_off) - row-split launches, see launch_gemm
  int group_m;           // tile walk: 0 / 1 = row-major, g > 1 = g M-tiles x all N tiles column by column (set by launch_gemm; see tile_coord)
  int halo_tw, halo_lg;  // set by launch_conv_halo only (kernels/conv_halo.hip): the tile is (256 / halo_tw) x halo_tw output pixels of one frame, halo_tw = 1 << halo_lg;
                         // the epilogue maps tile row r to launch row m0 + (r >> halo_lg) * Wo + (r & (halo_tw - 1))
  int tune_cfg_p1, tune_split_p1, tune_knobs;   // GemmTune of the launching context, + 1 so that a zeroed GemmP means "no override"
  // GroupNorm statistics of the OUTPUT from the epilogue (round 4): per block of `rb` consecutive output rows (rb = the wave tile's rows, reported by
  // launch_gemm) and per column the sum / sum of squares of the fp16 values stored, stat_part[blk * N + n]; stat_hw = output rows per frame (blocks
  // must not straddle frames).  launch_gemm declines (reports rb = 0, writes nothing) when the chosen kernel cannot: split-K, ragged tiles, ...
  float2* stat_part; int stat_hw;
};
// tuning overrides (A/B tools and the tile-config tests; ug_tune_force sets them on ONE context, the engine copies them into every GemmP):
// cfg / split -1 = planner's choice; knobs = bit mask documented in kernels/gemm.hip
struct GemmTune { int cfg = -1, split = -1, knobs = 0; };
static inline void gemm_apply_tune(GemmP& p, const GemmTune& t) { p.tune_cfg_p1 = t.cfg + 1; p.tune_split_p1 = t.split + 1; p.tune_knobs = t.knobs; }
void launch_gemm_mx8(const GemmP& p, hipStream_t s);   // dense only; C0 = lda and ldw in BYTES (= elements)
// fp16 [M, K] (row stride ldx) -> e4m3 bytes [M, K] + e8m0 scales (layout above, ld_s >= round_up(M, 256)); K % 128 == 0
void launch_quant_mx8(const f16* x, long ldx, long M, int K, unsigned char* q, unsigned* scales, long ld_s, hipStream_t s);
void launch_gemm(const GemmP& p, int batch, hipStream_t s, int* stat_rb = nullptr);   // stat_rb: rows per statistics block written (0 = none), see GemmP::stat_part
// weight-stationary streaming GEMM for K = 320, N in {320, 640, 960} (kernels/gemm_stream.hip; tile config 80): dense, fp16 out, bias + one residual
bool gemm_stream_supported(const GemmP& p, int batch);
void launch_gemm_stream(const GemmP& p, hipStream_t s);
// halo-staged 3x3 convolution (kernels/conv_halo.hip; tile configs 70 / 71 = 256 x 160 / 256 x 128, 72 / 73 = 192 x 128 / 192 x 160 output pixels x columns): stride 1, pad 1, chunk-major weights, whole 256-pixel tiles
bool conv_halo_supported(const GemmP& p, int batch, int bm, int bn);
void launch_conv_halo(const GemmP& p, int bm, int bn, hipStream_t s);
void gemm_plan(const GemmP& p, int batch, int* cfg_out, int* split_out);   // heuristic used when cfg/splitk are 0

// Fused GEGLU feed-forward (kernels/ff_fused.hip): Out = c0 * (GEGLU(X W1^T + b1) W2^T + b2) + c1 R1 + c2 R2, all [M, C] row-major (ld = C);
// W1 [8C][C] / b1 [8C] in the bound GEGLU row order (blocks of 16 rows = [8 value | 8 gate]), W2 [C][4C], C in {64,...,320}
struct FFusedP {
  const f16* X; const f16* W1; const f16* b1; const f16* W2; const f16* b2;
  const f16* R1; const f16* R2; float c0, c1, c2;
  f16* Out; int M, C; const f16* zero;
  // optional pre-norm, done on the X tile in LDS (what launch_layernorm would have written, same roundings):
  //   x' = fp16(X + addvec[row / rows_per_vec]) (addvec optional), X_used = fp16(LayerNorm(x') * ln_g + ln_b);
  // with addvec the R1 residual is taken as fp16(R1 + addvec[...]) as well (R1 == X: the block's residual stream)
  const f16* ln_g; const f16* ln_b; float ln_eps; const f16* addvec; int rows_per_vec;
  int variant;            // 0 = default (cross-tile prefetch), 1 = without it (A/B; Ctx::ff_variant, ug_tune_ff)
};
bool ff_fused_supported(int C);
void launch_ff_fused(const FFusedP& p, hipStream_t s);

// ---------------------------------------------------------------------------------------
// Normalisation (kernels/norm.hip)
// ---------------------------------------------------------------------------------------
// GroupNorm over a (virtual concat of two) channels-last tensor(s); statistics per frame
// (temporal == 0) or pooled over all T frames (temporal == 1); optional fused SiLU.
struct GroupNormP {
  const f16* X0; const f16* X1; int C0, C1;
  int T, HW, G; float eps; int temporal; int silu;
  const f16* gamma; const f16* beta;
  f16* Y;                 // [T*HW, C0+C1]
  float* ws;              // >= T * G * 2 * nchunk floats (+ T*G*2 for mean/rstd)
  int mode;               // 0 = automatic; 1 / 2 / 4 / 6 force a launch scheme (launch_groupnorm)
  // statistics already reduced per block of part_rb rows and per channel by the producing GEMM's epilogue (GemmP::stat_part; single source, HW % part_rb == 0):
  // gn_finalize_cols + gn_apply, no pass over X for the statistics
  const float2* part; int part_rb;
};
bool launch_groupnorm(const GroupNormP& p, hipStream_t s);                 // true = the statistics came from p.part (no pass over X for them)
bool groupnorm_uses_part(const GroupNormP& p, int part_rb);                // the launcher's rule for that, for the callers that decide whether a producer writes partial sums
size_t groupnorm_ws_floats(int T, int HW, int C, int G);

// LayerNorm over rows of [M, C]; optional pre-add of a per-frame broadcast vector
// (x += vec[(m / rows_per_vec) * C + c]) whose sum is also written back to Xout.
struct LayerNormP {
  const f16* X; f16* Y; int M, C; float eps;
  const f16* gamma; const f16* beta;
  const f16* addvec; int rows_per_vec; f16* Xout;   // optional
  long row0;                                        // index of row 0 in the addvec numbering (a row sub-range of a larger tensor)
  // optional MX-fp8 output INSTEAD of Y (fp8 linear path): e4m3 bytes [M][C] + e8m0 block scales, layout of launch_quant_mx8 (C % 128 == 0)
  unsigned char* Y8; unsigned* S8; long ld_s8;
};
void launch_layernorm(const LayerNormP& p, hipStream_t s);

// ---------------------------------------------------------------------------------------
// Attention (kernels/attn.hip)
// ---------------------------------------------------------------------------------------
// Flash-style self-attention, head_dim 64.  Q/K/V are column slices of row-major matrices
// (row stride ld*), batch b = frame, rows [b*S, (b+1)*S), head h at columns [h*64, h*64+64).
// Cross-attention: Sk != 0 gives the key/value sequence length (queries keep S); kv_shared = 1 makes every batch read the
// same Sk key/value rows (a text context computed once), otherwise batch b's keys are rows [b*Sk, (b+1)*Sk).
struct FlashP {
  const f16* Q; const f16* K; const f16* V; long ldq, ldk, ldv;
  f16* O; long ldo;
  int B, H, S; float scale;
  int Sk = 0; int kv_shared = 0;
  int variant = -1;   // A/B aid (tools/bench_flash.py; Ctx::flash_variant, ug_tune_flash): -1 = the default form (23), else a bit mask - 1 = one softmax step per 64 keys, 2 = XCD-grouped
                      // workgroup order, 4 = 2-slot K/V ring + 4 workgroups per CU
};
void launch_flash_attn64(const FlashP& p, hipStream_t s);
bool flash_attn_dh_supported(int d);
void launch_flash_attn_dh(const FlashP& p, int d, hipStream_t s);   // self-attention with head dim d in {32, 48, 80, 96, 112, 128} (CLIP ViT-H/14: 80)

// Temporal self-attention: for every pixel p and head h, sequence over the T frames
// (row of frame t = t*HW + p), head_dim 64, T <= 128 (BASELINE config 5 uses 50-frame clips; upstream DepthCrafter's default window is 110 frames).
struct TemporalAttnP {
  const f16* Q; const f16* K; const f16* V; long ld;
  f16* O; long ldo;
  int T, HW, H; float scale;
};
void launch_temporal_attn64(const TemporalAttnP& p, hipStream_t s);

// Row softmax fp32 [rows, ld_in] (first S cols valid) -> fp16 [rows, ld_out], zero tail.
void launch_softmax_rows(const float* in, long ld_in, f16* out, long ld_out, long rows, int S,
                         hipStream_t s);
// Batched transpose: in [B][R][C] (row stride ldi, batch stride sbi) -> out [B][C][Rpad]
// (row stride ldo >= R, zero tail, batch stride sbo).
void launch_transpose(const f16* in, long ldi, long sbi, f16* out, long ldo, long sbo, int B, int R,
                      int C, int nb_inner, long sbi_inner, long sbo_inner, hipStream_t s);

// ---------------------------------------------------------------------------------------
// Small / memory-bound kernels (kernels/misc.hip)
// ---------------------------------------------------------------------------------------
void launch_cast_f32_f16(const float* in, f16* out, long n, hipStream_t s);
void launch_cast_f16_f32(const f16* in, float* out, long n, hipStream_t s);
// weight re-layouts (one-off at bind time)
void launch_permute_conv_w(const f16* in, f16* out, int O, int I, int taps, int Ipad, int Opad,
                           hipStream_t s);   // [O][I][taps] -> [Opad][taps][Ipad] (zero padded)
void launch_upsample_phase_w(const f16* w9, f16* w4, int O, int I, int Ipad, hipStream_t s);   // [O][I][3][3] -> 4 x [O][2*2][Ipad]
void launch_rechunk_conv_w(const f16* in, f16* out, long O, int taps, int Ipad, hipStream_t s);   // [O][taps][Ipad] -> [O][Ipad/64][taps][64]
void launch_gather_rows(const f16* in, f16* out, const int* rowmap, int rows, int cols, hipStream_t s);
void launch_copy2d(const f16* in, long ldi, f16* out, long ldo, long rows, int cols, hipStream_t s);
void launch_fill_f16(f16* p, float v, long n, hipStream_t s);
void launch_fill_random(f16* p, long n, unsigned seed, hipStream_t s);   // uniform [-1,1) hash noise
void launch_add_rowvec(const f16* x, const f16* vec, f16* y, long M, int C, int rows_per_vec,
                       hipStream_t s);      // y[m,c] = x[m,c] + vec[(m/rows_per_vec)*C + c]
void launch_axpby(const f16* a, const f16* b, f16* y, float ca, float cb, long n, hipStream_t s);

// DepthCrafter pipeline glue
void launch_prep_video(const float* frames, const float* noise, f16* clip_src, f16* vae_in,
                       int T, int H, int W, float noise_aug, hipStream_t s);
void launch_clip_patchify(const f16* video_m11, f16* patches, int T, int H, int W, int S224,
                          int P, int Kpad, hipStream_t s, int imagenet_norm = 0);   // 0: CLIP mean/std, 1: ImageNet mean/std (DINOv2)
void launch_scale_rows(f16* w, const f16* gamma, int N, int K, hipStream_t s);     // w[n][:] *= gamma[n] (LayerScale folded into a projection)
void launch_add_grid_nearest(f16* x, const f16* grid, int B, int h, int w, int g, int C, hipStream_t s);   // x[b,y,x,:] += grid[b, y*g/h, x*g/w, :]
void launch_sn_normals_out(const f16* dec, int ldd, float* out, long pixels, hipStream_t s);   // clip to [-1,1], L2-normalise -> f32 [pixels,3]
void launch_resize_bilinear_aa(const float* in, float* out, int B, int Hi, int Wi, int Ho, int Wo, int C, int normalise, hipStream_t s);   // torch antialias bilinear; normalise: re-normalise the channel vector
void launch_init_latents2(const float* noise, f16* lat, float sigma0, int T, long hw, hipStream_t s);
void launch_silu_f16(const f16* in, f16* out, long n, hipStream_t s);
void launch_clip_assemble(const f16* patches, const f16* cls, const f16* pos, f16* tok, int T, int np, int d,
                          hipStream_t s);   // tok[t][0]=cls+pos[0]; tok[t][1+i]=patches[t*np+i]+pos[1+i]
void launch_make_unet_input(const f16* lat, const f16* cond, f16* x, long pixels, float inv_scale,
                            hipStream_t s);   // x[p, 0:4] = lat/sqrt(s^2+1), x[p,4:8] = cond
void launch_euler_step(const f16* v, f16* lat, long n, float sigma, float sigma_next, hipStream_t s);
void launch_scale_f16(const f16* in, f16* out, float sc, long n, hipStream_t s);
void launch_axpby_f16(const f16* x, float a, const f16* y, float b, f16* out, long n, hipStream_t s);
void launch_crossfade_f16(const f16* cur, f16* all, long frame_elems, int overlap, hipStream_t s);
void launch_pad_channels(const f16* in, int Cin, f16* out, int Cout, long pixels, hipStream_t s);
void launch_time_conv_out(const f16* x, const f16* w, const f16* b, float* frames_out, int T,
                          long HW, int Cs, hipStream_t s);  // + (x/2+0.5).clamp(0,1) -> f32 [T,HW,3]
void launch_depth_post(const float* frames, float* depth, float* minmax_ws, long pixels,
                       hipStream_t s);        // channel mean, clip-global min-max, 1/(x+0.1)
void launch_normals(const float* depth, const float* K33, float* normals, int T, int H, int W,
                    hipStream_t s);

// ---------------------------------------------------------------------------------------
// fp32-grade path over fp16 hi/lo pairs (kernels/wide.hip): the VAE encoder, which the reference runs in float32
// ---------------------------------------------------------------------------------------
void launch_split_pair(const float* x, f16* y, long M, int C, hipStream_t s);          // [M,C] f32 -> [M,2C] = [hi | lo]
void launch_add_f32(const float* a, const float* b, float* y, long n, hipStream_t s);
void launch_take_cols_f16(const float* x, long ldx, f16* y, long M, int C, hipStream_t s);
size_t gn32_ws_bytes(int T, int HW, int C, int G);
void launch_gn32_pair(const float* X, f16* Y, int T, int HW, int C, int G, float eps, int silu, const f16* gamma, const f16* beta,
                      void* ws, hipStream_t s);                                          // GroupNorm(+SiLU) f32 -> pair
void launch_qk_terms(const float* qkv, f16* Aq, f16* Bk, long M, int C, hipStream_t s);
void launch_vt_terms(const float* qkv, f16* Vt, int B, int S, int Spad, int C, hipStream_t s);
void launch_softmax_pair(const float* in, long ld_in, f16* out, int Spad, long rows, int S, hipStream_t s);

// calibration probe (kernels/probe.hip): chip-wide fp16 MFMA rate, operands in registers; scratch >= 256 * 512 floats
float bench_mfma_peak(float* scratch, int iters, hipStream_t s);

// ---------------------------------------------------------------------------------------
// Evaluation metrics on device (kernels/metrics.hip)
// ---------------------------------------------------------------------------------------
void launch_depth_fit(const float* pred, const float* gt, long n, float max_depth, double* part, int* nb, hipStream_t s);
void launch_depth_metrics(const float* pred, const float* gt, const unsigned char* cmask, long n, float max_depth, float sc,
                          float sh, double* part, int* nb, hipStream_t s);
void launch_normal_err(const float* pn, const float* gn, const unsigned char* mask, long n, float* err, double* part,
                       unsigned* hist, int* nb, hipStream_t s);
void launch_collect_bin(const float* err, long n, int bin, float* out, unsigned* count, unsigned cap, hipStream_t s);
