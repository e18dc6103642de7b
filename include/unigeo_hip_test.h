/* libunigeo_hip.so - TEST / TUNING entry points (not part of the drop-in boundary).
 *
 * include/unigeo_hip.h is the product header: what INTEGRATION.md section 2 maps to the reference's model/ plugin surface.  The entry points
 * here exist for the parity tests (stage- and op-level calls, host in / host out), the A/B tools under tools/ and the profiling scripts;
 * a maintainer integrating the library does not need them.  Same conventions as unigeo_hip.h (int return, ug_last_error).
 */
#ifndef UNIGEO_HIP_TEST_H
#define UNIGEO_HIP_TEST_H
#include "unigeo_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Tuning / parity aids for the fused GEGLU feed-forward kernel of the narrow transformer blocks (kernels/ff_fused.hip; the reference's
 * FeedForward module inside the un-vendored UNet): ug_set_ff_fused(0) falls back to two GEMM launches; ug_op_ff evaluates
 * c0 * (GEGLU(X W1^T + b1) W2^T + b2) + c1 * R1 on [M, C] with either implementation (W1 [8C][C], b1 [8C], W2 [C][4C] in diffusers order). */
int ug_set_ff_fused(ug_ctx* ctx, int on);   /* bit 0: fused feed-forward kernel, bit 1: the block's LayerNorm inside it; default 3 */
/* The same block with its pre-norm (reference: BasicTransformerBlock.norm3 -> ff, TemporalBasicTransformerBlock.norm_in -> ff_in, inside
 * the un-vendored UNet): out = c0 * FF(LayerNorm(x') * gamma + beta) + c1 * x', x' = fp16(X + addvec[row / rows_per_vec]) (addvec NULL: x' = X).
 * mode 0: LayerNorm launch + two GEMMs, 1: LayerNorm launch + fused feed-forward, 2: all inside the fused kernel (product path at C <= 320). */
int ug_op_ln_ff(ug_ctx* ctx, const float* X, int M, int C, const float* gamma, const float* beta, float eps, const float* addvec, int rows_per_vec,
                const float* W1, const float* b1, const float* W2, const float* b2, float c0, float c1, int mode, float* out);
int ug_bench_flash(ug_ctx* ctx, int B, int H, int S, int variant, int iters, float* us_out);   /* flash-attention A/B on device-resident random data */
int ug_bench_ff(ug_ctx* ctx, int M, int C, int fused, int iters, float* us_out);
int ug_op_ff(ug_ctx* ctx, const float* X, int M, int C, const float* W1, const float* b1, const float* W2, const float* b2, const float* R1,
             float c0, float c1, int fused, float* out);
/* Parity instrumentation (no reference counterpart; the reference would use the pipeline's callback_on_step_end): while
 * host_latents != NULL, ug_dc_run copies the latents after each of the first `steps` Euler steps to
 * host_latents[step][T][h][w][4] (float32, channels-last).  NULL switches it off.  Costs one host sync per step. */
int ug_dc_set_trace(ug_ctx* ctx, float* host_latents, int steps);
/* stage-level (parity tests): which = 0 YOSO pair / 1 refinement pair; use_ctrl: run the matching ControlNet on zimg (and DINO tokens) first */
int ug_sn_unet_forward(ug_ctx* ctx, int which, const float* sample_bchw, const float* zimg_bchw, int B, int h, int w, float t_unet,
                       float t_ctrl, const float* prompt_embeds, const float* dino_tokens, int use_ctrl, float* out_bchw);
int ug_sn_dino(ug_ctx* ctx, const float* images_bhwc, int B, int H, int W, float* tokens_out /*[B, g*g, D]*/);
int ug_sn_vae_decode(ug_ctx* ctx, const float* z_bchw, int B, int h, int w, float* out_bhwc /*[B,8h,8w,3] raw decoder output*/);
int ug_sn_vae_encode(ug_ctx* ctx, const float* img_m11_bhwc, int B, int H, int W, float* lat_out /*[B,4,H/8,W/8] posterior mode, unscaled*/);
/* Stage-level entry points (host in / host out) - what the parity tests drive.  Each replaces
 * the corresponding diffusers module call inside the pipeline (un-vendored). */
int ug_clip_embed(ug_ctx* ctx, const float* frames_thwc, int T, int H, int W, float* emb_out /*[T,proj]*/);
int ug_vae_encode(ug_ctx* ctx, const float* video_m11_thwc, int T, int H, int W, float* lat_out /*[T,4,H/8,W/8]*/);
int ug_vae_decode(ug_ctx* ctx, const float* z_tchw, int T, int h, int w, float* frames_out /*[T,8h,8w,3] in [0,1]*/);
int ug_unet_forward(ug_ctx* ctx, const float* sample_tchw /*[T,Cin,h,w]*/, int T, int h, int w, float timestep,
                    const float* clip_emb /*[T,cross]*/, float* out_tchw /*[T,Cout,h,w]*/);
/* Op-level entry points for kernel parity tests (row-major host matrices, fp32 in/out, computed in fp16). */
int ug_op_linear(ug_ctx* ctx, const float* A, int M, int K, const float* W, int N, const float* bias,
                 const float* R1, float c0, float c1, int act, int geglu, float* out);
/* MX-fp8 linear: A [M,K], W [N,K] are quantised on device (kernels/mx8.hip) and multiplied on the fp8 matrix cores; optional outputs:
 * the quantised A bytes [M,K] and its scale dwords [K/128][round_up(M,256)] (4 e8m0 per dword) for bit-level checks. */
int ug_op_linear_mx8(ug_ctx* ctx, const float* A, int M, int K, const float* W, int N, const float* bias, int geglu, float* out,
                     unsigned char* a8_out, unsigned* scales_out);
int ug_op_conv(ug_ctx* ctx, const float* x0_thwc, int C0, const float* x1_thwc, int C1, int T, int H, int W,
               const float* weight /*[O][I][kt][ky][kx]*/, const float* bias, int O, int kt, int k, int stride,
               int pad_t, int pad_l, int ups, float* out_thwc);
/* convolution (k x k pad k/2, or (kt,1,1)) + residual, then GroupNorm + SiLU of its output with the statistics (a) from a pass over the stored tensor,
 * (b) from the convolution's epilogue (GemmP::stat_part); *rb_out = rows per statistics block (0: the planner's kernel has no statistics epilogue) */
int ug_op_conv_gn(ug_ctx* ctx, const float* x_thwc, int C0, int T, int H, int W, const float* weight, const float* bias, const float* res, int O, int kt, int k,
                  int G, float eps, int temporal, const float* gamma, const float* beta, float* conv_out, float* y_pass, float* y_epi, int* rb_out);
int ug_op_groupnorm(ug_ctx* ctx, const float* x0, int C0, const float* x1, int C1, int T, int HW, int G, float eps,
                    int temporal, int silu, const float* gamma, const float* beta, float* out);
int ug_op_layernorm(ug_ctx* ctx, const float* x, int M, int C, float eps, const float* gamma, const float* beta,
                    const float* addvec, int rows_per_vec, float* out, float* xout);
int ug_op_flash_attn(ug_ctx* ctx, const float* qkv /*[B*S,3*H*64]*/, int B, int H, int S, float* out /*[B*S,H*64]*/);
int ug_op_temporal_attn(ug_ctx* ctx, const float* qkv /*[T*HW,3*H*64]*/, int T, int HW, int H, float* out);
int ug_op_attention_generic(ug_ctx* ctx, const float* qkv /*[B*S,3*H*d]*/, int B, int S, int H, int d, float* out);
int ug_op_flash_attn_dh(ug_ctx* ctx, const float* qkv /*[B*S,3*H*d]*/, int B, int S, int H, int d, float* out);   /* fused self-attention, head dim d in {32,48,80,96,112,128}: the CLIP tower's 16 x 80 heads */
int ug_op_euler_step(ug_ctx* ctx, const float* v, float* latents_inout, long n, float sigma, float sigma_next);
/* Tuning aids (not on the product path): GEMM / implicit-conv microbenchmark on device-resident random data,
 * and an override of the tile/split-K heuristic (-1 = heuristic). ms_out: [ms per launch, cfg, split, M, K]. */
/* GroupNorm launch-scheme A/B (mode: launch_groupnorm in kernels/norm.hip); tuning aid, no reference counterpart. */
int ug_bench_groupnorm(ug_ctx* ctx, int C0, int C1, int T, int HW, int temporal, int mode, int iters, float* us_out);
int ug_bench_gemm(ug_ctx* ctx, int M, int N, int K, int conv, int T, int Hi, int Wi, int C0, int C1, int kt, int k,
                  int stride, int ups, int cfg, int split, int iters, float* ms_out);
/* calibration: chip-wide fp16 MFMA rate with operands in registers (16x16x32 f16, 256 workgroups x 8 waves), TFLOP/s - the matrix-pipe
 * ceiling at the clock the power management allows under a pure matrix load (bench.py reports it next to the 2.5 PFLOP/s spec peak) */
int ug_bench_mfma_peak(ug_ctx* ctx, int iters, float* tflops_out);
int ug_tune_force(ug_ctx* ctx, int cfg, int split);   /* test / A-B aid, per context: force a GEMM tile config (cfg >= 0) and split-K factor for this context's launches, (-1, -1) = planner; cfg = -100 - mask sets the knob mask (kernels/gemm.hip) */
int ug_tune_ff(ug_ctx* ctx, int variant);      /* A/B aid, per context: fused feed-forward kernel variant, 0 = cross-tile prefetch (default), 1 = without it; bit-identical outputs */
int ug_tune_flash(ug_ctx* ctx, int variant);   /* test aid, per context: flash-attention variant mask of this context's launches (bit 0: one softmax step per 64 keys, bit 1: XCD-grouped workgroup order, bit 2: 2-slot ring + 4 workgroups per CU, bit 4: lazy rescale + dot2 row sums, bit 6: 8-wave ping-pong kernel; -1 = default 23) */
int ug_profile_begin_shapes(ug_ctx* ctx);   /* same, keyed by kernel family AND problem shape */

#ifdef __cplusplus
}
#endif
#endif
