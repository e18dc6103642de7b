/* libunigeo_hip.so - C ABI of the MI355X-native (gfx950) DepthCrafter inference path.
 *
 * This is the drop-in boundary behind UniGeo's model/ plugin surface.  Each entry point
 * names the reference interface it replaces (paths relative to the reference checkout).
 * Plain C types only: no torch / numpy types cross this boundary.
 *
 * Conventions
 *   - every function returning int: 0 = OK, non-zero = error; the message is available from
 *     ug_last_error(ctx) (the reference raises Python exceptions instead - the ctypes shim in
 *     unigeo_amd/_lib.py turns a non-zero code back into RuntimeError).
 *   - the caller owns all host buffers; the library owns all device memory.
 *   - one context per GPU; calls on a context must be serialised by the caller; the library
 *     runs on its own HIP stream and synchronises before returning.
 *   - "video" tensors are channels-last: frames [T,H,W,3] float32 in [0,1] exactly as
 *     DepthCrafter.prepare_input produces them (model/depthcrafter.py:39-45).
 *   - noise is an INPUT (the reference draws it from the global CUDA RNG without a generator,
 *     model/depthcrafter.py:80-90): noise_latents [T,4,H/8,W/8], noise_aug [T,3,H,W], float32,
 *     laid out as torch.randn would produce them inside the pipeline (NCHW).
 */
#ifndef UNIGEO_HIP_H
#define UNIGEO_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ug_ctx ug_ctx;

enum { UG_DTYPE_F16 = 0, UG_DTYPE_F32 = 1 };

/* Architecture hyper-parameters.  Defaults (ug_*_config_default) are SVD-XT / DepthCrafter:
 * replaces the config.json files read by from_pretrained (model/depthcrafter.py:18-29). */
typedef struct {
  int in_channels, out_channels, num_levels;
  int block_out_channels[8];
  int num_attention_heads[8];
  int down_has_attn[8];
  int layers_per_block, cross_attention_dim, addition_time_embed_dim, projection_class_embeddings_input_dim;
  int norm_groups;
  float eps_cross_attn_blocks, eps_plain_down_block, eps_mid_block, eps_up_blocks;
} ug_unet_config;

typedef struct {
  int in_channels, out_channels, latent_channels, num_levels;
  int block_out_channels[8];
  int layers_per_block, norm_groups;
  float scaling_factor;
} ug_vae_config;

typedef struct {
  int hidden_size, intermediate_size, num_hidden_layers, num_attention_heads, image_size, patch_size, projection_dim;
  float layer_norm_eps;
} ug_clip_config;

void ug_unet_config_default(ug_unet_config* c);
void ug_vae_config_default(ug_vae_config* c);
void ug_clip_config_default(ug_clip_config* c);

/* Context: replaces `self.device = cuda:0` + pipeline.to(device) (model/depthcrafter.py:11,31).
 * workspace_bytes: transient activation arena; persist_bytes: weights in kernel-ready layout. */
ug_ctx* ug_create(int device_id, size_t workspace_bytes, size_t persist_bytes);
void ug_destroy(ug_ctx* ctx);
const char* ug_last_error(ug_ctx* ctx);   /* ctx may be NULL: returns the creation error */
size_t ug_workspace_peak(ug_ctx* ctx);

/* Weights: replaces DiffusersUNet...from_pretrained / DepthCrafterPipeline.from_pretrained
 * (model/depthcrafter.py:18-29).  Tensors are uploaded under their diffusers / transformers
 * state-dict names with a component prefix ("unet.", "vae.", "clip."), then bound; binding
 * hard-fails on any missing, mis-shaped or unexpected tensor. */
int ug_load_tensor(ug_ctx* ctx, const char* name, int dtype, int ndim, const int64_t* shape, const void* host_data);
int ug_bind_unet(ug_ctx* ctx, const ug_unet_config* cfg);
int ug_bind_vae(ug_ctx* ctx, const ug_vae_config* cfg);
int ug_bind_clip(ug_ctx* ctx, const ug_clip_config* cfg);

/* The pipeline call: replaces `self.pipeline(frames, height, width, output_type="np",
 * guidance_scale=1.0, num_inference_steps, window_size=len(frames), overlap, ...).frames[0]`
 * (model/depthcrafter.py:80-90) plus the wrapper post-processing at :92-97 (depth) and
 * prepare_output at :48-59 (normals, OpenGL frame).
 *   ug_dc_set_inputs : host -> HBM (frames, noise, per-frame 3x3 intrinsics or NULL)
 *   ug_dc_run        : CLIP + VAE-encode + `steps` x (scale, concat, UNet, Euler) + VAE temporal
 *                      decode (chunks of decode_chunk frames) + depth (+ normals); all on device
 *   ug_dc_get_outputs: HBM -> host; any pointer may be NULL
 *                      frames_out [T,H,W,3], depth_out [T,H,W], normals_out [T,H,W,3] float32 */
int ug_dc_set_inputs(ug_ctx* ctx, const float* frames_thwc, int T, int H, int W, const float* noise_latents,
                     const float* noise_aug, const float* intrinsics_t33);
int ug_dc_run(ug_ctx* ctx, int steps, int decode_chunk, int with_normals);
/* Long-video mode of the pipeline call (`window_size` / `overlap` of model/depthcrafter.py:87-88, which the reference pins to
 * len(frames) / 25, i.e. OFF): latent sliding windows of `window` (<= 128) frames with `overlap` re-noised + cross-faded frames,
 * restated from upstream DepthCrafter's published pipeline (UNPINNED).  window == 0 or >= T is ug_dc_run.  The first `window`
 * frames of the noise passed to ug_dc_set_inputs are the window noise (rotated by `overlap` frames per window, as upstream). */
int ug_dc_run_windows(ug_ctx* ctx, int steps, int decode_chunk, int with_normals, int window, int overlap);
int ug_dc_get_outputs(ug_ctx* ctx, float* frames_out, float* depth_out, float* normals_out);
/* Arithmetic of the VAE *encoder*.  The reference pipeline up-casts the VAE to float32 around encode (diffusers force_upcast;
 * pipeline built at model/depthcrafter.py:24-29) and runs everything else in fp16.  on = 1 (default): float32-grade encoder -
 * fp32 residual stream / GroupNorm / softmax, GEMMs on fp16 hi/lo activation pairs against the (fp16-valued) weights, which is
 * exact to fp32 rounding.  on = 0: fp16 storage with fp32 accumulation, like the decoder (faster, ~1e-3 off the fp32 result). */
int ug_set_vae_encode_fp32(ug_ctx* ctx, int on);
/* GroupNorm launch scheme of the UNet-sized tensors (A/B, parity tests): on = 1 one launch - a workgroup keeps its rows in registers while the
 * per-frame statistics are handed over through write-through partials and a ticket (kernels/norm.hip: gn_fused); on = 0 (default) the three launches
 * (statistics / finalise / apply).  Same arithmetic up to the order of the fp32 chunk sums.  Measured: the hand-off (a chain of ~8 uncached round
 * trips) costs more than the second read of the tensor it saves - 45 vs 40 us at level 0, 22 vs 16 us on the smallest tensors. */
int ug_set_gn_fused(ug_ctx* ctx, int on);
/* Independent sub-graphs of one pipeline call in flight at a time (default 1 = strictly one kernel after another; 2 measured -0.6 % on the headline clip).  The reference's
 * pipeline encodes / decodes the clip in chunks of `decode_chunk_size` frames one after the other and computes the CLIP embeddings before
 * them (the calls inside pipeline(...) at model/depthcrafter.py:80-90); those chunks do not depend on each other, so the engine issues them on
 * separate HIP streams - one chunk's HBM-bound passes overlap another's MFMA-bound ones.  Same kernels, same launch parameters:
 * outputs are bit-identical for every setting. */
int ug_set_concurrency(ug_ctx* ctx, int lanes);
/* Tuning / parity aids for the fused GEGLU feed-forward kernel of the narrow transformer blocks (kernels/ff_fused.hip; the reference's
 * FeedForward module inside the un-vendored UNet): ug_set_ff_fused(0) falls back to two GEMM launches; ug_op_ff evaluates
 * c0 * (GEGLU(X W1^T + b1) W2^T + b2) + c1 * R1 on [M, C] with either implementation (W1 [8C][C], b1 [8C], W2 [C][4C] in diffusers order). */
int ug_set_ff_fused(ug_ctx* ctx, int on);   /* bit 0: fused feed-forward kernel, bit 1: the block's LayerNorm inside it, bit 2: fused LayerNorm -> Q|K|V projection (measured slower, off); default 3 */
/* The same block with its pre-norm (reference: BasicTransformerBlock.norm3 -> ff, TemporalBasicTransformerBlock.norm_in -> ff_in, inside
 * the un-vendored UNet): out = c0 * FF(LayerNorm(x') * gamma + beta) + c1 * x', x' = fp16(X + addvec[row / rows_per_vec]) (addvec NULL: x' = X).
 * mode 0: LayerNorm launch + two GEMMs, 1: LayerNorm launch + fused feed-forward, 2: all inside the fused kernel (product path at C <= 320). */
int ug_op_ln_ff(ug_ctx* ctx, const float* X, int M, int C, const float* gamma, const float* beta, float eps, const float* addvec, int rows_per_vec,
                const float* W1, const float* b1, const float* W2, const float* b2, float c0, float c1, int mode, float* out);
/* LayerNorm -> linear as one kernel (the Q|K|V projections of the narrow transformer blocks; reference: BasicTransformerBlock.norm1 -> attn1.to_q/k/v
 * and TemporalBasicTransformerBlock.norm1 -> attn1 inside the un-vendored UNet).  out [M, N] = LN(X [M, C]) . W [N, C]^T + bias; fused = 0 runs the
 * LayerNorm launch + GEMM it replaces; iters > 0 also times the call (us_out). */
int ug_op_ln_linear(ug_ctx* ctx, const float* X, int M, int C, const float* gamma, const float* beta, float eps, const float* W, int N, const float* bias,
                    int fused, int iters, float* out, float* us_out);
int ug_bench_flash(ug_ctx* ctx, int B, int H, int S, int variant, int iters, float* us_out);   /* flash-attention A/B on device-resident random data */
int ug_bench_ff(ug_ctx* ctx, int M, int C, int fused, int iters, float* us_out);
int ug_op_ff(ug_ctx* ctx, const float* X, int M, int C, const float* W1, const float* b1, const float* W2, const float* b2, const float* R1,
             float c0, float c1, int fused, float* out);
/* BASELINE configs[4] (north_star: "fp8 MFMA ... (CDNA4 fp8)"): on = 1 runs the UNet transformers' linear layers whose K is a multiple
 * of 128 on MX-fp8 matrix instructions (v_mfma_scale_f32_16x16x128_f8f6f4: OCP e4m3 elements, one e8m0 power-of-two scale per 32
 * K elements; activations are quantised on the fly, weights once at bind time).  Reduced precision - the reference has no fp8 path;
 * the measured error against the fp16 path and the oracle is reported by tests/test_fp8_gpu.py.  Default 0. */
int ug_set_fp8_linears(ug_ctx* ctx, int on);
/* Parity instrumentation (no reference counterpart; the reference would use the pipeline's callback_on_step_end): while
 * host_latents != NULL, ug_dc_run copies the latents after each of the first `steps` Euler steps to
 * host_latents[step][T][h][w][4] (float32, channels-last).  NULL switches it off.  Costs one host sync per step. */
int ug_dc_set_trace(ug_ctx* ctx, float* host_latents, int steps);
/* Device addresses of the resident outputs (valid until the next ug_dc_set_inputs): lets the caller hand
 * them to RCCL (torch.distributed) for the cross-GPU gather without a host round trip. */
int ug_dc_device_ptrs(ug_ctx* ctx, void** frames_dev, void** depth_dev, void** normals_dev);

/* StableNormal: replaces `self.predictor = torch.hub.load("Stable-X/StableNormal", "StableNormal")` (model/stablenormal.py:16) and
 * `self.predictor(image)` (:39).  The hub predictor's code is un-vendored; what runs here is the restatement documented in
 * oracle/stablenormal.py / DESIGN.md (UNPINNED): SD AutoencoderKL, two SD-2.1-class UNet2DConditionModels (one-step "YOSO" estimate
 * + DDIM refinement), two ControlNet trunks (image latent; image latent + DINOv2 tokens) and a DINOv2 ViT-L/14 tower.
 *   ug_bind_stablenormal : tensors uploaded with ug_load_tensor under "sn.vae.", "sn.unet_yoso.", "sn.controlnet_yoso.", "sn.unet.",
 *                          "sn.controlnet_dino.", "sn.dino."; sd = UNet2DConditionModel config (ug_unet_config; the SVD-only fields are
 *                          ignored), dino = ViT config (ug_clip_config; projection_dim ignored)
 *   ug_sn_run            : images [B,H,W,3] float32 in [0,1] (H, W multiples of 64; the frames of a clip are a batch - the reference
 *                          loops over them), prompt_embeds [77, cross_attention_dim] float32 (text-encoder output for the fixed
 *                          prompt, computed once by the caller), YOSO timestep, and the refinement schedule as data: nsteps DDIM
 *                          timesteps with the per-step update x <- ca[i]*x + cb[i]*unet(x, t_i)  ->  unit normals [B,H,W,3] in [-1,1] */
int ug_bind_stablenormal(ug_ctx* ctx, const ug_unet_config* sd, const ug_vae_config* vae, const ug_clip_config* dino);
int ug_sn_run(ug_ctx* ctx, const float* images_bhwc, int B, int H, int W, const float* prompt_embeds, float yoso_timestep, int nsteps,
              const float* timesteps, const float* ca, const float* cb, float* normals_out);
/* stage-level (parity tests): which = 0 YOSO pair / 1 refinement pair; use_ctrl: run the matching ControlNet on zimg (and DINO tokens) first */
int ug_sn_unet_forward(ug_ctx* ctx, int which, const float* sample_bchw, const float* zimg_bchw, int B, int h, int w, float t_unet,
                       float t_ctrl, const float* prompt_embeds, const float* dino_tokens, int use_ctrl, float* out_bchw);
int ug_sn_dino(ug_ctx* ctx, const float* images_bhwc, int B, int H, int W, float* tokens_out /*[B, g*g, D]*/);
int ug_sn_vae_decode(ug_ctx* ctx, const float* z_bchw, int B, int h, int w, float* out_bhwc /*[B,8h,8w,3] raw decoder output*/);
/* Antialiased bilinear resize [B,Hi,Wi,C] -> [B,Ho,Wo,C] (float32, C <= 4) on the device = torch F.interpolate(mode="bilinear",
 * align_corners=False, antialias=True); normalise = 1 re-normalises the channel vector of every output pixel (unit normals).  Used by the
 * StableNormal predictor's optional processing resolution (the hub predictor behind model/stablenormal.py:16,39 resizes its input to a fixed
 * processing resolution and the prediction back - DESIGN.md section 9, S1). */
int ug_resize_bilinear(ug_ctx* ctx, const float* in_bhwc, int B, int Hi, int Wi, int C, int Ho, int Wo, int normalise, float* out_bhwc);
int ug_sn_vae_encode(ug_ctx* ctx, const float* img_m11_bhwc, int B, int H, int W, float* lat_out /*[B,4,H/8,W/8] posterior mode, unscaled*/);

/* Stage-level entry points (host in / host out) - what the parity tests drive.  Each replaces
 * the corresponding diffusers module call inside the pipeline (un-vendored). */
int ug_clip_embed(ug_ctx* ctx, const float* frames_thwc, int T, int H, int W, float* emb_out /*[T,proj]*/);
int ug_vae_encode(ug_ctx* ctx, const float* video_m11_thwc, int T, int H, int W, float* lat_out /*[T,4,H/8,W/8]*/);
int ug_vae_decode(ug_ctx* ctx, const float* z_tchw, int T, int h, int w, float* frames_out /*[T,8h,8w,3] in [0,1]*/);
int ug_unet_forward(ug_ctx* ctx, const float* sample_tchw /*[T,Cin,h,w]*/, int T, int h, int w, float timestep,
                    const float* clip_emb /*[T,cross]*/, float* out_tchw /*[T,Cout,h,w]*/);
int ug_normals_from_depth(ug_ctx* ctx, const float* depth_thw, const float* intrinsics_t33, int T, int H, int W,
                          float* normals_out);

/* Evaluation metrics on device.  pred == NULL uses the resident output of the last ug_dc_run
 * (depth [T,H,W] / normals [T,H,W,3]); gt / mask are host arrays (mask: 1 byte per pixel, may be NULL).
 *   ug_eval_depth  : replaces depth_evaluation(..., custom_mask, align_with_lstsq=True) (metrics/eval_depth.py:6-246,
 *                    metrics/alignment.py:150-167) -> out[11] = AbsRel, SqRel, RMSE, LogRMSE, d<1, d<1.25, d<1.25^2,
 *                    d<1.25^3, valid_pixels, scale, shift
 *   ug_eval_normal : replaces normal_evaluation (metrics/eval_normal.py:4-72) -> out[8] = mean, median, rmse,
 *                    %<5, %<7.5, %<11.25, %<22.5, %<30 */
int ug_eval_depth(ug_ctx* ctx, const float* pred_depth, const float* gt_depth, const unsigned char* custom_mask, long n,
                  float max_depth, double* out11);
int ug_eval_normal(ug_ctx* ctx, const float* pred_normals, const float* gt_normals, const unsigned char* mask, long n,
                   double* out8);

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
int ug_tune_force(ug_ctx* ctx, int cfg, int split);   /* test / A-B aid, per context: force a GEMM tile config (cfg >= 0) and split-K factor for this context's launches, (-1, -1) = planner; cfg = -100 - mask sets the knob mask (kernels/gemm.hip) */
int ug_tune_ff(int variant);      /* A/B aid: fused feed-forward kernel variant, 1 = GEGLU of chunk j software-pipelined into the MFMAs of chunk j + 1 (default), 0 = the round-2 kernel; bit-identical outputs */
int ug_tune_flash(int variant);   /* test aid: process default of the flash-attention variant mask (bit 0: one softmax step per 64 keys, bit 1: XCD-grouped workgroup order, bit 2: 2-slot ring + 4 workgroups per CU, bit 4: lazy rescale + dot2 row sums, bit 5: software-pipelined kernel; default 23).  ug_bench_flash passes its variant with the launch and leaves this alone. */

/* HIP-event profiling of everything launched between begin and end; end returns a JSON
 * object {kernel_family: {ms, calls, flops, bytes}} valid until the next call on ctx. */
int ug_profile_begin(ug_ctx* ctx);
int ug_profile_begin_shapes(ug_ctx* ctx);   /* same, keyed by kernel family AND problem shape */
const char* ug_profile_end(ug_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
