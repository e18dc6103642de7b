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
/* Independent sub-graphs of one pipeline call in flight at a time (default 1 = strictly one kernel after another; 2 measured -0.6 % on the headline clip).  The reference's
 * pipeline encodes / decodes the clip in chunks of `decode_chunk_size` frames one after the other and computes the CLIP embeddings before
 * them (the calls inside pipeline(...) at model/depthcrafter.py:80-90); those chunks do not depend on each other, so the engine issues them on
 * separate HIP streams - one chunk's HBM-bound passes overlap another's MFMA-bound ones.  Same kernels, same launch parameters:
 * outputs are bit-identical for every setting. */
int ug_set_concurrency(ug_ctx* ctx, int lanes);
/* Two (or more) contexts on ONE GPU, each running its own clip (round 5): the reference's evaluation loop handles one independent clip after the other
 * (eval.py:33-56: `for data_idx ...: output = model.forward(data)`), so a second plugin instance on the same GPU can process the next clip meanwhile - its
 * kernels fill the CUs that one clip's tile tails and under-filled launches leave idle (+10 % aggregate frames/s).  on = 1 tells a context that it shares the
 * GPU: heuristics that pay extra launches / work to fill the last round of ONE kernel are dropped (the fused feed-forward takes all rows, the tile planner
 * ignores the last-round fill).  Same arithmetic per layer up to the tile choice (all tiles are bit-identical, tests/test_ops_gpu.py).  Default 0. */
int ug_set_coscheduled(ug_ctx* ctx, int on);
/* BASELINE configs[4] (north_star: "fp8 MFMA ... (CDNA4 fp8)"): on = 1 runs the UNet transformers' linear layers whose K is a multiple
 * of 128 on MX-fp8 matrix instructions (v_mfma_scale_f32_16x16x128_f8f6f4: OCP e4m3 elements, one e8m0 power-of-two scale per 32
 * K elements; activations are quantised on the fly, weights once at bind time).  Reduced precision - the reference has no fp8 path;
 * the measured error against the fp16 path and the oracle is reported by tests/test_fp8_gpu.py.  Default 0. */
int ug_set_fp8_linears(ug_ctx* ctx, int on);
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
/* Antialiased bilinear resize [B,Hi,Wi,C] -> [B,Ho,Wo,C] (float32, C <= 4) on the device = torch F.interpolate(mode="bilinear",
 * align_corners=False, antialias=True); normalise = 1 re-normalises the channel vector of every output pixel (unit normals).  Used by the
 * StableNormal predictor's optional processing resolution (the hub predictor behind model/stablenormal.py:16,39 resizes its input to a fixed
 * processing resolution and the prediction back - DESIGN.md section 9, S1). */
int ug_resize_bilinear(ug_ctx* ctx, const float* in_bhwc, int B, int Hi, int Wi, int C, int Ho, int Wo, int normalise, float* out_bhwc);

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

/* HIP-event profiling of everything launched between begin and end; end returns a JSON
 * object {kernel_family: {ms, calls, flops, bytes}} valid until the next call on ctx. */
int ug_profile_begin(ug_ctx* ctx);
const char* ug_profile_end(ug_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif

