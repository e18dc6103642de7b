// MI355X-native DepthCrafter engine: context, device arena, weight registry, model graphs.
// Host C++ only sequences kernels on one HIP stream; every FLOP of the hot path runs in the
// hand-written kernels under kernels/.  No PyTorch, no BLAS/MIOpen, no fallback.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace ug {

struct RawTensor { void* dev = nullptr; int dtype = 0; std::vector<long> shape; long numel = 0; bool used = false; };

class Arena {
 public:
  void init(size_t bytes);
  void destroy();
  void* alloc(size_t bytes);
  template <class T> T* get(long n) { return (T*)alloc((size_t)n * sizeof(T)); }
  size_t mark() const { return off_; }
  void release(size_t m) { off_ = m; }
  size_t peak() const { return peak_; }
  size_t capacity() const { return cap_; }
  // a sub-arena over memory carved from another arena (lanes, see run_lanes): never destroy()ed, the owner releases its mark instead
  void view(void* base, size_t bytes) { base_ = (char*)base; cap_ = bytes; off_ = 0; peak_ = 0; hi_ = 0; }
  // high-water mark since the last reset_high(): how much a graph function needed on top of the offset it started from
  void reset_high() { hi_ = off_; }
  size_t high() const { return hi_; }
  void note_peak(size_t p) { if (p > peak_) peak_ = p; }
 private:
  char* base_ = nullptr; size_t cap_ = 0, off_ = 0, peak_ = 0, hi_ = 0;
};

struct Lin { const f16* w = nullptr; const f16* b = nullptr; int in = 0, out = 0;
             // optional MX-fp8 copy of the weight (kernels/mx8.hip): e4m3 bytes [out][in] + e8m0 block scales [in/128][ld_sw8] dwords
             const unsigned char* w8 = nullptr; const unsigned* sw8 = nullptr; long ld_sw8 = 0; };
struct Conv { const f16* w = nullptr; const f16* b = nullptr; int cin = 0, cinp = 0, cout = 0, kt = 1, ky = 1, kx = 1;
              const f16* wphase = nullptr;     // wphase: 4 x [cout][2*2][cinp] sub-pixel weights of a nearest-2x upsample conv
              int kchunk = 0; };               // 1: w / wphase are in the chunk-major K order [cout][cinp/64][taps][64] (GemmP::kchunk)
struct Norm { const f16* g = nullptr; const f16* b = nullptr; int c = 0; float eps = 1e-5f; };

struct Res2D { Norm n1, n2; Conv c1, c2, sc; Lin temb; bool has_sc = false, has_temb = false; int tidx = -1; };
struct ResT { Norm n1, n2; Conv c1, c2; Lin temb; bool has_temb = false; int tidx = -1; };
struct STRes { Res2D s; ResT t; float alpha = 0.5f; int cin = 0, cout = 0; };

struct Transformer {
  int C = 0, heads = 0;
  Norm gn; Lin proj_in, proj_out;
  Norm ln1, ln3; Lin qkv1, o1, v2, o2, ff1, ff2;
  Norm ln_in, tln1, tln3; Lin ffin1, ffin2, tqkv, to1, tv2, to2, tff1, tff2;
  Lin tpe1, tpe2;
  float alpha = 0.5f;
  // per-clip caches (device, persistent)
  f16* frame_emb = nullptr; int frame_emb_T = 0;   // [T][C]
  f16* cross_sp = nullptr;                          // [T][C]
  f16* cross_tm = nullptr;                          // [1][C]
};

struct UNetCfg {
  int in_ch = 8, out_ch = 4, nlev = 4; int boc[8] = {320, 640, 1280, 1280}; int heads[8] = {5, 10, 20, 20};
  int has_attn[8] = {1, 1, 1, 0}; int layers = 2, cross_dim = 1024, add_dim = 256, proj_in_dim = 768, groups = 32;
  float eps_xattn = 1e-6f, eps_down = 1e-5f, eps_mid = 1e-5f, eps_up = 1e-6f;
};
struct VAECfg { int in_ch = 3, out_ch = 3, lat = 4, nlev = 4; int boc[8] = {128, 256, 512, 512}; int layers = 2, groups = 32; float scaling = 0.18215f; };
struct CLIPCfg { int hidden = 1280, inter = 5120, layers = 32, heads = 16, image = 224, patch = 14, proj = 1024; float eps = 1e-5f; };

struct UNet {
  UNetCfg cfg; bool bound = false;
  Conv conv_in, conv_out; Norm norm_out;
  Lin te1, te2, ae1, ae2;
  struct Down { std::vector<STRes> res; std::vector<Transformer> attn; Conv down; bool has_down = false; };
  struct Up { std::vector<STRes> res; std::vector<Transformer> attn; Conv up; bool has_up = false; };
  std::vector<Down> down; std::vector<Up> up;
  STRes mid0, mid1; Transformer mid_attn;
  std::vector<Res2D*> temb_s; std::vector<ResT*> temb_t;   // res-blocks with time_emb_proj, in order
  // per-run caches
  f16* tproj = nullptr; long tproj_stride = 0; int tproj_steps = 0; std::vector<long> tproj_off_s, tproj_off_t;
};

struct VAttn { Norm gn; Lin qkv, out; int C = 0; };
struct VAE {
  VAECfg cfg; bool bound = false;
  Conv e_in, e_out, quant; Norm e_norm;
  struct EDown { std::vector<Res2D> res; Conv down; bool has_down = false; };
  std::vector<EDown> edown; Res2D emid0, emid1; VAttn eattn;
  // the same encoder with K-doubled weights [W | W] for fp16 hi/lo-pair activations (kernels/wide.hip): the reference runs
  // the encoder in float32 (force_upcast); norms share gamma / beta with the fp16 structures above
  std::vector<EDown> edown_w; Res2D emid0_w, emid1_w; VAttn eattn_w; Conv e_out_w, quant_w; bool wide_bound = false;
  Conv d_in, d_out; Norm d_norm; const f16* tco_w = nullptr; const f16* tco_b = nullptr;
  std::vector<STRes> dmid; VAttn dattn;
  struct DUp { std::vector<STRes> res; Conv up; bool has_up = false; };
  std::vector<DUp> dup;
  // plain 2-D decoder of the SD AutoencoderKL (StableNormal): mid res blocks, per-level res blocks (upsamplers live in dup[i].up)
  bool dec2d = false; Conv post_quant; std::vector<Res2D> d2_mid; std::vector<std::vector<Res2D>> d2_up;
};

struct CLIPLayer { Norm ln1, ln2; Lin qkv, out, fc1, fc2; };
struct CLIP {
  CLIPCfg cfg; bool bound = false;
  const f16* patch_w = nullptr; int patch_k = 0;  // [hidden][Kpad]
  const f16* patch_b = nullptr;                   // DINOv2's patch embedding has a bias (CLIP's has none)
  const f16* cls = nullptr; const f16* pos = nullptr;
  Norm pre, post; std::vector<CLIPLayer> layers; Lin proj;
};

// ---- StableNormal (BASELINE configs[3]): SD-2.1-class UNet2DConditionModel / ControlNet trunk, DINOv2 tower (sn_graphs.inc) ----
struct SDTransformer {
  int C = 0, heads = 0;
  Norm gn; Lin proj_in, proj_out; Norm ln1, ln2, ln3; Lin qkv1, o1, q2, kv2, o2, ff1, ff2;
  f16* kv = nullptr;    // per-run cache: [77][2C] = to_k | to_v of the text context
};
struct TembSet { std::vector<Res2D*> res; f16* tproj = nullptr; std::vector<long> off; int steps = 0; };
struct SDTrunk {
  Conv conv_in; Lin te1, te2;
  struct Down { std::vector<Res2D> res; std::vector<SDTransformer> attn; Conv down; bool has_down = false; };
  std::vector<Down> down; Res2D mid0, mid1; SDTransformer mid_attn;
};
struct SDUNetM {
  SDTrunk tr; TembSet ts;
  struct Up { std::vector<Res2D> res; std::vector<SDTransformer> attn; Conv up; bool has_up = false; };
  std::vector<Up> up; Norm norm_out; Conv conv_out;
};
struct ControlNetM { SDTrunk tr; TembSet ts; std::vector<Conv> zero; Conv zero_mid; Lin dino_proj; bool has_dino = false; };
struct SN {
  UNetCfg cfg; bool bound = false;
  SDUNetM unet_y, unet_r; ControlNetM ctrl_y, ctrl_d; VAE vae; CLIP dino;
};

struct ProfRec { std::string name; double flops, bytes; hipEvent_t e0, e1; };

// A lane = one extra HIP stream + a private slice of the activation arena.  Independent sub-graphs of one clip (the 8-frame chunks of the VAE
// encoder / decoder, the CLIP tower) are issued on different lanes so that one chunk's HBM-bound passes (GroupNorm, fp32 adds, splits) and
// the thin last rounds of its persistent GEMMs overlap another chunk's MFMA-bound kernels (engine.hip: run_lanes).
struct Lane { hipStream_t stream = nullptr; hipEvent_t done = nullptr; };

struct Ctx {
  int device = 0; hipStream_t stream = nullptr;
  Arena ws;        // transient activations
  Arena persist;   // bound weights + per-clip caches
  f16* zero = nullptr;
  std::string err;
  std::unordered_map<std::string, RawTensor> raw;
  UNet unet; VAE vae; CLIP clip; SN sn;
  // profiling
  bool prof_on = false; bool prof_shapes = false; std::vector<ProfRec> prof; std::string prof_json;
  // resident pipeline I/O
  int T = 0, H = 0, W = 0;
  float* d_frames = nullptr; float* d_noise_lat = nullptr; float* d_noise_aug = nullptr; float* d_K = nullptr;
  float* d_out_frames = nullptr; float* d_depth = nullptr; float* d_normals = nullptr; float* d_mm = nullptr;
  f16* d_clip_emb = nullptr; f16* d_cond = nullptr; f16* d_lat = nullptr;
  size_t io_mark = 0; bool io_ready = false;
  // parity instrumentation: when set, dc_run copies the latents after every Euler step to this host buffer ([steps][T*h*w*4] f32)
  float* trace_host = nullptr; int trace_steps = 0;
  // page-locked staging for the host buffers the C ABI hands over (pageable numpy memory): a copy through it is one memcpy + one DMA at
  // link rate instead of the runtime's chunked staging of a pageable hipMemcpy (slot 0: inputs, 1: outputs); grown on demand
  void* pin[2] = {nullptr, nullptr}; size_t pin_sz[2] = {0, 0};
  int ff_fused = 3;          // bit 0: fused GEGLU feed-forward kernel for the narrow (C <= 320) transformer blocks (0 = two GEMM launches);
                             // bit 1: its LayerNorm (+ broadcast row added to the residual stream) applied inside that kernel (A/B runs)
  int fp8_linears = 0;       // 1 = run the UNet's eligible linear layers on MX-fp8 MFMAs (BASELINE configs[4]; reduced precision, off by default)
  int vae_encode_fp32 = 1;   // 1 = reference behaviour (float32-grade encoder), 0 = fp16 storage like the decoder
  // lanes (run_lanes): streams are created on first use; lane_need remembers the arena bytes a task kind needed when it first ran serially
  std::vector<Lane> lanes; hipEvent_t fork_ev = nullptr;
  int concurrency = 1;       // independent sub-graphs in flight (1 = everything on the one stream, in order; ug_set_concurrency); outputs are
                             // bit-identical.  Measured on the 25 x 384 x 512 clip: 2 lanes -0.6 %, 3 lanes +-0 - the persistent GEMMs fill the
                             // register file of every CU, so a second stream's kernels only get the tail rounds
  std::map<std::string, size_t> lane_need;
  int cur_lane = 0;          // 0 = main stream, l + 1 = lane l (run_lanes)
  int cosched = 0;           // this context shares the GPU with another clip in flight (ug_set_coscheduled): heuristics that trade extra launches / work for a fuller
                             // last round of ONE kernel are off - the fused feed-forward takes all rows (no two-GEMM tail), the tile planner ignores the last-round fill
  int ff_variant = 0, flash_variant = -1;   // ug_tune_ff / ug_tune_flash: per-context A/B overrides copied into FFusedP / FlashP (0 / -1 = the defaults)
  GemmTune tune;             // ug_tune_force: tile-config / split-K / knob overrides for THIS context's GEMM launches (tests, A/B tools)
};

void* pinned(Ctx& c, int slot, size_t bytes);   // page-locked staging buffer of at least `bytes`

// ---- binding ----
void upload_raw(Ctx& c, const std::string& name, int dtype, const std::vector<long>& shape, const void* host);
void bind_unet(Ctx& c, const UNetCfg& cfg, const std::string& prefix);
void bind_vae(Ctx& c, const VAECfg& cfg, const std::string& prefix);
void bind_clip(Ctx& c, const CLIPCfg& cfg, const std::string& prefix);
void finish_binding(Ctx& c, const std::string& prefix);   // fail on unused tensors, free raw

// ---- graphs (device pointers, stream-ordered, transient memory from c.ws) ----
// x [T,h,w,in_ch] f16; clip_emb [T,cross_dim] f16; tsteps host array of continuous timesteps
void unet_prepare(Ctx& c, int T, const f16* clip_emb, const float* timesteps, int nsteps);
f16* unet_forward(Ctx& c, const f16* x, int T, int h, int w, int step);   // -> [T,h,w,out_ch] (ws)
f16* vae_encode(Ctx& c, const f16* x8, int T, int H, int W);              // x8 [T,H,W,8] -> [T,H/8,W/8,4]; precision per c.vae_encode_fp32
void vae_decode(Ctx& c, const f16* z, int T, int h, int w, float* frames_out);  // z [T,h,w,4] (already /scaling) -> f32 [T,8h,8w,3]
f16* clip_embed(Ctx& c, const f16* video_m11, int T, int H, int W);       // [T,H,W,3] -> [T,proj]

f16* vae_encode_v(Ctx& c, VAE& v, const f16* x8, int T, int H, int W, bool fp32_grade);

// ---- StableNormal (sn_graphs.inc) ----
void bind_sn(Ctx& c, const UNetCfg& ucfg, const VAECfg& vcfg, const CLIPCfg& dcfg, const std::string& prefix);   // "<prefix>{vae,unet_yoso,controlnet_yoso,unet,controlnet_dino,dino}."
// images f32 [B,H,W,3] in [0,1] (host), prompt f32 [77,cross] (host) -> normals f32 [B,H,W,3] (host)
void sn_run(Ctx& c, const float* images, int B, int H, int W, const float* prompt, float yoso_t, int nsteps, const float* timesteps,
            const float* ca, const float* cb, float* normals_out);
// stage-level (parity tests): which = 0 YOSO pair, 1 refinement pair; device tensors, channels-last fp16
f16* sn_unet_eval(Ctx& c, int which, const f16* sample4, const f16* zimg4, int B, int h, int w, float t_unet, float t_ctrl, const f16* prompt,
                  const f16* dino_tok, int use_ctrl);
f16* sn_dino_tokens(Ctx& c, const f16* img_m11, int B, int H, int W);          // [B, g*g, D]
f16* sn_vae_decode(Ctx& c, const f16* z4, int B, int h, int w);                // [B, 8h, 8w, 8] (3 valid channels), raw decoder output

// ---- pipeline ----
void dc_set_inputs(Ctx& c, const float* frames, int T, int H, int W, const float* noise_lat, const float* noise_aug,
                   const float* K33);
void dc_run(Ctx& c, int steps, int chunk, int with_normals, int window = 0, int overlap = 0);
void dc_get_outputs(Ctx& c, float* frames, float* depth, float* normals);

// profiling helpers
void prof_begin(Ctx& c, bool shapes = false);
std::string prof_end(Ctx& c);

}  // namespace ug
