// Engine core: arena, weight binding (diffusers / transformers state-dict names -> kernel-ready
// fp16 layouts), op wrappers with optional HIP-event profiling.
#include "engine.h"
#include <chrono>

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <exception>
#include <sstream>

namespace ug {

// ------------------------------------------------------------------ arena
void Arena::init(size_t bytes) {
  UG_CHECK(hipMalloc((void**)&base_, bytes));
  cap_ = bytes; off_ = 0; peak_ = 0;
}
void Arena::destroy() { if (base_) (void)hipFree(base_); base_ = nullptr; cap_ = 0; }
void* Arena::alloc(size_t bytes) {
  const size_t a = (bytes + 255) & ~(size_t)255;
  if (off_ + a > cap_)
    throw std::runtime_error("device arena exhausted: need " + std::to_string(off_ + a) + " bytes, capacity " +
                             std::to_string(cap_) + " (raise workspace_bytes / persist_bytes in ug_create)");
  void* p = base_ + off_;
  off_ += a;
  if (off_ > peak_) peak_ = off_;
  if (off_ > hi_) hi_ = off_;
  return p;
}

// A pipeline call's hold on the activation arena: whatever way the call ends (a failed UG_REQUIRE, "device arena exhausted", a HIP error), the
// stream(s) are drained, the arena returns to where the call found it and the per-run caches that point into it are forgotten - a later call
// on the same context starts from a clean arena instead of failing with "arena exhausted" for ever.
struct RunGuard {
  Ctx& c; size_t mk;
  explicit RunGuard(Ctx& c_) : c(c_), mk(c_.ws.mark()) {}
  ~RunGuard() {
    (void)hipStreamSynchronize(c.stream);
    for (auto& l : c.lanes) (void)hipStreamSynchronize(l.stream);
    c.ws.release(mk);
    c.unet.tproj = nullptr; c.unet.tproj_steps = 0;
    auto forget = [](Transformer& t) { t.frame_emb = nullptr; t.frame_emb_T = 0; t.cross_sp = nullptr; t.cross_tm = nullptr; };
    for (auto& d : c.unet.down) for (auto& t : d.attn) forget(t);
    forget(c.unet.mid_attn);
    for (auto& d : c.unet.up) for (auto& t : d.attn) forget(t);
    auto forget_sd = [](SDTransformer& t) { t.kv = nullptr; };
    auto forget_trunk = [&](SDTrunk& tr) { for (auto& d : tr.down) for (auto& t : d.attn) forget_sd(t); forget_sd(tr.mid_attn); };
    for (SDUNetM* u : {&c.sn.unet_y, &c.sn.unet_r}) { forget_trunk(u->tr); for (auto& d : u->up) for (auto& t : d.attn) forget_sd(t); u->ts.tproj = nullptr; u->ts.steps = 0; }
    for (ControlNetM* m : {&c.sn.ctrl_y, &c.sn.ctrl_d}) { forget_trunk(m->tr); m->ts.tproj = nullptr; m->ts.steps = 0; }
  }
};

// ------------------------------------------------------------------ lanes
// Issue `ntask` mutually independent sub-graphs.  body(i) enqueues task i on c.stream with transient memory from c.ws and must write its
// results into buffers the caller allocated beforehand.  `kind(i)` names the task's shape (e.g. "dec:8x384x512"): the first time a kind runs
// it runs alone on the main stream and the arena bytes it needed are recorded; from then on tasks are spread over up to c.concurrency
// lanes, each with a private arena slice of the recorded size, forked from / joined to the main stream with events.  The kernels and
// their launch parameters are the same either way, so outputs are bit-identical (tests/test_pipeline_gpu.py).  Event-bracketed
// profiling passes (c.prof_on) stay serial so that every kernel is timed alone.
// Lanes that run the same kernel sequence start in lockstep and stay there (the dispatcher arbitrates the queues symmetrically), so all of them
// sit in their HBM-bound passes at the same time and in their MFMA-bound ones at the same time - measured: no gain at all.  A one-wave
// kernel that sleeps for ~us microseconds at the head of lane l breaks the symmetry.
__global__ void k_lane_stagger(int us) {
  const long long t0 = wall_clock64();                 // 100 MHz constant clock
  while (wall_clock64() - t0 < (long long)us * 100) __builtin_amdgcn_s_sleep(64);
}

template <class Kind, class Body>
static void run_lanes(Ctx& c, int ntask, Kind kind, Body body) {
  std::vector<size_t> need(ntask, 0);
  bool known = true;
  for (int i = 0; i < ntask; ++i) {
    auto it = c.lane_need.find(kind(i));
    if (it == c.lane_need.end()) known = false; else need[i] = it->second;
  }
  int nl = std::min(c.concurrency, ntask);
  size_t slice = 0;
  for (size_t n : need) slice = std::max(slice, n);
  slice = (slice + (1 << 20)) & ~(size_t)255;
  if (known) while (nl > 1 && c.ws.mark() + (size_t)nl * slice > c.ws.capacity()) --nl;    // not enough arena for nl slices: fewer lanes
  if (!known || nl <= 1 || c.prof_on) {
    for (int i = 0; i < ntask; ++i) {
      const size_t m0 = c.ws.mark();
      c.ws.reset_high();
      body(i);
      c.lane_need[kind(i)] = c.ws.high() - m0;
      UG_REQUIRE(c.ws.mark() == m0, "lane task must release its transient memory");
    }
    return;
  }
  while ((int)c.lanes.size() < nl) {
    Lane l;
    UG_CHECK(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
    UG_CHECK(hipEventCreateWithFlags(&l.done, hipEventDisableTiming));
    c.lanes.push_back(l);
  }
  if (!c.fork_ev) UG_CHECK(hipEventCreateWithFlags(&c.fork_ev, hipEventDisableTiming));
  const size_t mk = c.ws.mark();
  char* base = (char*)c.ws.alloc((size_t)nl * slice);
  hipStream_t main_stream = c.stream;
  Arena main_ws = c.ws;
  UG_CHECK(hipEventRecord(c.fork_ev, main_stream));
  static const int stagger_us = getenv("UG_LANE_STAGGER") ? atoi(getenv("UG_LANE_STAGGER")) : 1500;
  for (int l = 0; l < nl; ++l) {
    UG_CHECK(hipStreamWaitEvent(c.lanes[l].stream, c.fork_ev, 0));
    if (l > 0 && stagger_us > 0) hipLaunchKernelGGL(k_lane_stagger, dim3(1), dim3(64), 0, c.lanes[l].stream, l * stagger_us);
  }
  if (getenv("UG_LANE_DEBUG")) fprintf(stderr, "[ug] run_lanes: %d tasks on %d lanes, slice %.2f GB (%s ...)\n", ntask, nl, slice / 1073741824.0, kind(0).c_str());
  // longest task first, each to the lane with the least work so far (need is a fair proxy for a chunk's cost)
  std::vector<int> order(ntask); for (int i = 0; i < ntask; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return need[a] > need[b]; });
  std::vector<size_t> load(nl, 0);
  std::vector<std::vector<int>> plan(nl);
  for (int i : order) { const int l = (int)(std::min_element(load.begin(), load.end()) - load.begin()); plan[l].push_back(i); load[l] += need[i] + 1; }
  try {
    // round-robin over the lanes' queues so that the host feeds all streams evenly
    for (size_t k = 0;; ++k) {
      bool any = false;
      for (int l = 0; l < nl; ++l) {
        if (k >= plan[l].size()) continue;
        any = true;
        c.stream = c.lanes[l].stream;
        c.ws.view(base + (size_t)l * slice, slice);
        c.cur_lane = l + 1;
        body(plan[l][k]);
      }
      if (!any) break;
    }
  } catch (...) {
    c.stream = main_stream; c.ws = main_ws; c.cur_lane = 0;
    for (int l = 0; l < nl; ++l) (void)hipStreamSynchronize(c.lanes[l].stream);
    c.ws.release(mk);
    throw;
  }
  c.stream = main_stream; c.ws = main_ws; c.cur_lane = 0;
  for (int l = 0; l < nl; ++l) {
    UG_CHECK(hipEventRecord(c.lanes[l].done, c.lanes[l].stream));
    UG_CHECK(hipStreamWaitEvent(main_stream, c.lanes[l].done, 0));
  }
  c.ws.release(mk);    // stream-ordered: whatever reuses these bytes is enqueued behind the joins
}

// ------------------------------------------------------------------ profiling
struct ProfScope {
  Ctx& c; int idx = -1;
  ProfScope(Ctx& c_, const char* name, double flops, double bytes) : c(c_) {
    if (!c.prof_on) return;
    ProfRec r; r.name = name; r.flops = flops; r.bytes = bytes;
    UG_CHECK(hipEventCreate(&r.e0)); UG_CHECK(hipEventCreate(&r.e1));
    UG_CHECK(hipEventRecord(r.e0, c.stream));
    c.prof.push_back(r); idx = (int)c.prof.size() - 1;
  }
  ~ProfScope() { if (idx >= 0) (void)hipEventRecord(c.prof[idx].e1, c.stream); }
};
void prof_begin(Ctx& c, bool shapes) { c.prof.clear(); c.prof_on = true; c.prof_shapes = shapes; }
std::string prof_end(Ctx& c) {
  c.prof_on = false;
  UG_CHECK(hipStreamSynchronize(c.stream));
  struct Agg { double ms = 0, flops = 0, bytes = 0; long calls = 0; };
  std::map<std::string, Agg> agg;
  for (auto& r : c.prof) {
    float ms = 0.f;
    UG_CHECK(hipEventElapsedTime(&ms, r.e0, r.e1));
    Agg& a = agg[r.name]; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes; a.calls++;
    (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
  }
  std::ostringstream o;
  o << "{";
  bool first = true;
  if (c.prof_shapes) {   // launch order of the GEMM-family records (joined with rocprofv3's per-dispatch counters by tools/pmc_per_shape.py)
    o << "\"__order__\":[";
    bool f2 = true;
    for (auto& r : c.prof) {
      if (r.name.compare(0, 5, "gemm_") != 0) continue;
      if (!f2) o << ",";
      f2 = false;
      o << "[\"" << r.name << "\"," << r.bytes << "]";
    }
    o << "]";
    first = false;
  }
  c.prof.clear();
  for (auto& kv : agg) {
    if (!first) o << ",";
    first = false;
    o << "\"" << kv.first << "\":{\"ms\":" << kv.second.ms << ",\"calls\":" << kv.second.calls
      << ",\"flops\":" << kv.second.flops << ",\"bytes\":" << kv.second.bytes << "}";
  }
  o << "}";
  return o.str();
}

// ------------------------------------------------------------------ op wrappers
struct Epi {
  const f16* bias2 = nullptr; const f16* R1 = nullptr; long ldr1 = 0; float c1 = 1.f;
  const f16* R2 = nullptr; long ldr2 = 0; float c2 = 1.f; float c0 = 1.f; int act = 0; int flags = 0;
  const unsigned char* a8 = nullptr; const unsigned* sa8 = nullptr; long ld_sa8 = 0;   // A already in MX-fp8 (written by the producing LayerNorm)
  float alg = 1.f;   // algorithmic / executed FLOPs of this launch (0.5 for the K-doubled hi/lo-pair GEMMs of the float32-grade encoder)
  struct StatPart* so = nullptr; int stat_hw = 0;   // GroupNorm statistics of the output from the epilogue (stat_hw: rows per frame of a dense output)
};
// GroupNorm statistics handed from the producing GEMM's epilogue to the GroupNorm that follows (GemmP::stat_part): allocated by the caller next to the
// tensor, filled by conv() / linear() when the chosen kernel can (rb = rows per block, 0 = not written: groupnorm() then runs its statistics pass)
struct StatPart { float2* part = nullptr; int rb = 0; };
static StatPart stat_alloc(Ctx& c, long M, int N, int T, int hw, int G, int temporal) {
  StatPart s;
  // M >= 8192: below that GroupNorm keeps its one- / two-launch slab forms (4096 measured: +2 ms per clip); and only where launch_groupnorm would actually
  // consume the partial sums for the (T, hw, N, G, temporal) GroupNorm that follows - the launcher's own rule (groupnorm_uses_part)
  GroupNormP g; memset(&g, 0, sizeof(g));
  g.C0 = N; g.T = T; g.HW = hw; g.G = G; g.temporal = temporal; g.gamma = c.zero; g.beta = c.zero;
  if (!(c.tune.knobs & 131072) && M >= 8192 && N % G == 0 && groupnorm_uses_part(g, 16))
    s.part = (float2*)c.ws.get<float>(((M + 47) / 48) * (long)N * 2);   // 48 = the smallest wave tile
  return s;
}

static void run_gemm(Ctx& c, GemmP p, int batch, const char* tag, float alg = 1.f, StatPart* so = nullptr, int stat_hw = 0) {
  p.zero = c.zero;
  gemm_apply_tune(p, c.tune);
  if (p.nb_inner < 1) p.nb_inner = 1;
  char nm[128];
  if (c.prof_on && c.prof_shapes) snprintf(nm, sizeof(nm), "%s:%dx%dx%d%s", tag, p.M, p.N, p.K, batch > 1 ? "b" : "");
  else snprintf(nm, sizeof(nm), "%s", tag);
  int cfg, split;
  gemm_plan(p, batch, &cfg, &split);
  p.cfg_p1 = cfg + 1; p.splitk = split;
  const size_t mk = c.ws.mark();
  if (split > 1) p.partial = c.ws.get<float>((long)split * p.M * p.N);
  {
    // algorithmic FLOPs: a sub-pixel phase of an upsample conv stands for 9/4 of the MACs it executes
    // algorithmic bytes: every operand element once (the im2col view counts its source tensors, not the 9x gather), fp16
    const double nout = (p.flags & UG_F_GEGLU) ? p.N / 2 : p.N;
    const double a_el = p.conv ? (double)p.T * p.Hi * p.Wi * (p.C0 + p.C1) : (double)p.M * p.K;
    const double bytes = 2.0 * batch * (a_el + (double)p.N * p.K + (double)p.M * nout * ((p.flags & UG_F_OUT_F32) ? 2 : 1) +
                                        (p.R1 ? (double)p.M * nout : 0.0) + (p.R2 ? (double)p.M * nout : 0.0));
    ProfScope ps(c, nm, 2.0 * p.M * p.N * (double)p.K * batch * (p.up_phase ? 2.25 : 1.0) * alg, bytes);
    if (so && so->part) { p.stat_part = so->part; p.stat_hw = p.conv ? p.Ho * p.Wo : stat_hw; }
    launch_gemm(p, batch, c.stream, so && so->part ? &so->rb : nullptr);
  }
  c.ws.release(mk);
}

void* pinned(Ctx& c, int slot, size_t bytes) {
  if (c.pin_sz[slot] < bytes) {
    if (c.pin[slot]) UG_CHECK(hipHostFree(c.pin[slot]));
    c.pin[slot] = nullptr; c.pin_sz[slot] = 0;
    UG_CHECK(hipHostMalloc(&c.pin[slot], bytes, hipHostMallocDefault));
    c.pin_sz[slot] = bytes;
  }
  return c.pin[slot];
}

static inline long pad256(long x) { return (x + 255) / 256 * 256; }
// Out[M, lin.out] = A[M, lin.in] W^T (+bias) ...
static void linear(Ctx& c, const f16* A, long M, const Lin& l, f16* out, const Epi& e = Epi(), long lda = 0,
                   long ldo = 0) {
  GemmP p; memset(&p, 0, sizeof(p));
  if (c.fp8_linears && l.w8 && M >= 256 && !(e.flags & UG_F_OUT_F32)) {
    // MX-fp8 path: quantise the activation rows (32-element blocks, e8m0 scales), then the same persistent GEMM on e4m3 operands
    const size_t mk = c.ws.mark();
    const int K = l.in;
    const unsigned char* a8 = e.a8; const unsigned* sa = e.sa8; long ld_sa = e.ld_sa8;
    unsigned char* a8w = nullptr; unsigned* saw = nullptr;
    if (!a8) {
      a8w = (unsigned char*)c.ws.alloc((size_t)M * K);
      ld_sa = pad256(M);
      saw = (unsigned*)c.ws.alloc((size_t)(K / 128) * ld_sa * 4);
      a8 = a8w; sa = saw;
    }
    char nmq[96], nmg[96];
    if (c.prof_on && c.prof_shapes) { snprintf(nmq, sizeof(nmq), "quant_mx8:%ldx%d", M, K); snprintf(nmg, sizeof(nmg), "gemm_linear_mx8:%ldx%dx%d", M, l.out, K); }
    else { snprintf(nmq, sizeof(nmq), "quant_mx8"); snprintf(nmg, sizeof(nmg), "gemm_linear_mx8"); }
    if (a8w) {
      ProfScope ps(c, nmq, 0, (double)M * K * 3.0);
      launch_quant_mx8(A, lda ? lda : K, M, K, a8w, saw, ld_sa, c.stream);
    }
    p.A0 = (const f16*)a8; p.C0 = K; p.M = (int)M; p.N = l.out; p.K = K;
    p.W = (const f16*)l.w8; p.ldw = K; p.bias = l.b; p.bias2 = e.bias2;
    p.R1 = e.R1; p.ldr1 = e.ldr1; p.c1 = e.c1; p.R2 = e.R2; p.ldr2 = e.ldr2; p.c2 = e.c2; p.c0 = e.c0;
    p.act = e.act; p.flags = e.flags;
    const int nout = (e.flags & UG_F_GEGLU) ? l.out / 2 : l.out;
    p.Out = out; p.ldo = ldo ? ldo : nout;
    if (p.R1 && !p.ldr1) p.ldr1 = nout;
    if (p.R2 && !p.ldr2) p.ldr2 = nout;
    p.sa = sa; p.ld_sa = ld_sa; p.sw = l.sw8; p.ld_sw = l.ld_sw8; p.zero = c.zero; p.nb_inner = 1;
    gemm_apply_tune(p, c.tune);
    {
      ProfScope ps(c, nmg, 2.0 * M * (double)l.out * K, (double)M * K + (double)l.out * K + 2.0 * M * nout);
      launch_gemm_mx8(p, c.stream);
    }
    c.ws.release(mk);
    return;
  }
  p.A0 = A; p.C0 = (int)(lda ? lda : l.in); p.M = (int)M; p.N = l.out; p.K = l.in;
  p.W = l.w; p.ldw = l.in; p.bias = l.b; p.bias2 = e.bias2;
  p.R1 = e.R1; p.ldr1 = e.ldr1; p.c1 = e.c1; p.R2 = e.R2; p.ldr2 = e.ldr2; p.c2 = e.c2; p.c0 = e.c0;
  p.act = e.act; p.flags = e.flags;
  const int nout = (e.flags & UG_F_GEGLU) ? l.out / 2 : l.out;
  p.Out = out; p.ldo = ldo ? ldo : nout;
  if (p.R1 && !p.ldr1) p.ldr1 = nout;
  if (p.R2 && !p.ldr2) p.ldr2 = nout;
  run_gemm(c, p, 1, "gemm_linear", e.alg, e.so, e.stat_hw);
}

// implicit-GEMM convolution over channels-last sources
static void conv(Ctx& c, const f16* x0, int C0, const f16* x1, int C1, int T, int Hi, int Wi, const Conv& cv,
                 int stride, int pad_t, int pad_l, int ups, f16* out, const Epi& e = Epi(), long ldo = 0) {
  UG_REQUIRE(C0 + C1 == cv.cinp, "conv input channels do not match the bound weight");
  if (ups == 2 && cv.wphase && stride == 1 && pad_t == 1 && pad_l == 1) {
    // sub-pixel decomposition: four 2x2 convs on the source grid (4*Cin instead of 9*Cin MACs per output)
    for (int ph = 0; ph < 4; ++ph) {
      GemmP p; memset(&p, 0, sizeof(p));
      p.conv = 1; p.A0 = x0; p.A1 = x1; p.C0 = C0; p.C1 = C1;
      p.T = T; p.Hi = Hi; p.Wi = Wi; p.ups = 1; p.stride = 1; p.pad_t = (ph >> 1) ? 0 : 1; p.pad_l = (ph & 1) ? 0 : 1;
      p.kt = 1; p.ky = 2; p.kx = 2; p.Ho = Hi; p.Wo = Wi;
      p.M = T * Hi * Wi; p.N = cv.cout; p.K = cv.cinp * 4;
      p.W = cv.wphase + (long)ph * cv.cout * 4 * cv.cinp; p.ldw = p.K; p.bias = cv.b; p.bias2 = e.bias2;
      p.c0 = e.c0; p.act = e.act; p.flags = e.flags; p.Out = out; p.ldo = ldo ? ldo : cv.cout; p.up_phase = 1 + ph; p.kchunk = cv.kchunk;
      UG_REQUIRE(!e.R1 && !e.R2, "sub-pixel upsample conv does not take residuals");
      run_gemm(c, p, 1, "gemm_conv_up2x2");
    }
    return;
  }
  GemmP p; memset(&p, 0, sizeof(p));
  p.conv = 1; p.A0 = x0; p.A1 = x1; p.C0 = C0; p.C1 = C1;
  p.T = T; p.Hi = Hi; p.Wi = Wi; p.ups = ups; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
  p.kt = cv.kt; p.ky = cv.ky; p.kx = cv.kx;
  p.Ho = (Hi * ups) / stride; p.Wo = (Wi * ups) / stride;
  p.M = T * p.Ho * p.Wo; p.N = cv.cout; p.K = cv.cinp * cv.kt * cv.ky * cv.kx;
  p.W = cv.w; p.ldw = p.K; p.bias = cv.b; p.bias2 = e.bias2;
  p.R1 = e.R1; p.ldr1 = e.ldr1 ? e.ldr1 : cv.cout; p.c1 = e.c1;
  p.R2 = e.R2; p.ldr2 = e.ldr2 ? e.ldr2 : cv.cout; p.c2 = e.c2; p.c0 = e.c0; p.act = e.act; p.flags = e.flags;
  p.Out = out; p.ldo = ldo ? ldo : cv.cout; p.kchunk = cv.kchunk;
  run_gemm(c, p, 1, cv.kt > 1 ? "gemm_tconv" : (cv.ky > 1 ? "gemm_conv3x3" : "gemm_conv1x1"), e.alg, e.so);
}

static void groupnorm(Ctx& c, const f16* x0, int C0, const f16* x1, int C1, int T, int HW, int G, const Norm& n,
                      int temporal, int silu, f16* y, const StatPart* sp = nullptr) {
  UG_REQUIRE(C0 + C1 == n.c, "groupnorm channel mismatch");
  GroupNormP p; memset(&p, 0, sizeof(p));
  if (sp && sp->part && sp->rb > 0 && C1 == 0 && HW % sp->rb == 0) { p.part = sp->part; p.part_rb = sp->rb; }
  p.X0 = x0; p.X1 = x1; p.C0 = C0; p.C1 = C1; p.T = T; p.HW = HW; p.G = G; p.eps = n.eps;
  p.temporal = temporal; p.silu = silu; p.gamma = n.g; p.beta = n.b; p.Y = y;
  const size_t mk = c.ws.mark();
  p.ws = c.ws.get<float>((long)groupnorm_ws_floats(T, HW, n.c, G));
  {
    char nm[96];
    if (c.prof_on && c.prof_shapes) snprintf(nm, sizeof(nm), "groupnorm:T%dxHW%dxC%d%s", T, HW, n.c, temporal ? "t" : "");
    else snprintf(nm, sizeof(nm), "groupnorm");
    ProfScope ps(c, nm, 0, (double)T * HW * n.c * 2.0 * 3.0);
    launch_groupnorm(p, c.stream);
  }
  c.ws.release(mk);   // stream-ordered: later kernels that reuse this memory run after the GN kernels
}

// MX-fp8 image of a LayerNorm output for the fp8 linear path: carved from c.ws by the caller's mark, filled by layernorm()
struct QAct { unsigned char* a8 = nullptr; unsigned* sa = nullptr; long ld = 0; };
static QAct qact_alloc(Ctx& c, long M, int K) {
  QAct q; q.a8 = (unsigned char*)c.ws.alloc((size_t)M * K); q.ld = pad256(M);
  q.sa = (unsigned*)c.ws.alloc((size_t)(K / 128) * q.ld * 4);
  return q;
}
static void layernorm(Ctx& c, const f16* x, long M, const Norm& n, f16* y, const f16* addvec = nullptr,
                      long rows_per_vec = 1, f16* xout = nullptr, const QAct* q = nullptr, long row0 = 0) {
  LayerNormP p; memset(&p, 0, sizeof(p));
  p.X = x; p.Y = y; p.M = (int)M; p.C = n.c; p.eps = n.eps; p.gamma = n.g; p.beta = n.b;
  p.addvec = addvec; p.rows_per_vec = (int)rows_per_vec; p.Xout = xout; p.row0 = row0;
  if (q) { p.Y8 = q->a8; p.S8 = q->sa; p.ld_s8 = q->ld; }
  ProfScope ps(c, "layernorm", 0, (double)M * n.c * 2.0 * (addvec ? 3.0 : 2.0));
  launch_layernorm(p, c.stream);
}


// ------------------------------------------------------------------ raw tensor registry
void upload_raw(Ctx& c, const std::string& name, int dtype, const std::vector<long>& shape, const void* host) {
  UG_REQUIRE(dtype == 0 || dtype == 1, "dtype must be 0 (fp16) or 1 (fp32)");
  RawTensor t; t.dtype = dtype; t.shape = shape; t.numel = 1;
  for (long d : shape) t.numel *= d;
  const size_t bytes = (size_t)t.numel * (dtype == 0 ? 2 : 4);
  UG_CHECK(hipMalloc(&t.dev, bytes ? bytes : 4));
  UG_CHECK(hipMemcpy(t.dev, host, bytes, hipMemcpyHostToDevice));
  auto it = c.raw.find(name);
  if (it != c.raw.end()) { (void)hipFree(it->second.dev); c.raw.erase(it); }
  c.raw[name] = t;
}

static RawTensor& raw_get(Ctx& c, const std::string& name, std::initializer_list<long> shape) {
  auto it = c.raw.find(name);
  if (it == c.raw.end()) throw std::runtime_error("missing weight tensor: " + name);
  RawTensor& t = it->second;
  std::vector<long> want(shape);
  if (t.shape != want) {
    std::string s = "weight " + name + " has shape [";
    for (long d : t.shape) s += std::to_string(d) + ",";
    s += "] expected [";
    for (long d : want) s += std::to_string(d) + ",";
    throw std::runtime_error(s + "]");
  }
  t.used = true;
  return t;
}

// fp16 view of a raw tensor (cast through the persist arena when it was uploaded as fp32)
static const f16* raw_f16(Ctx& c, RawTensor& t, bool copy) {
  if (t.dtype == 0 && !copy) return (const f16*)t.dev;
  f16* d = c.persist.get<f16>(t.numel);
  if (t.dtype == 0) UG_CHECK(hipMemcpyAsync(d, t.dev, (size_t)t.numel * 2, hipMemcpyDeviceToDevice, c.stream));
  else launch_cast_f32_f16((const float*)t.dev, d, t.numel, c.stream);
  return d;
}
static const f16* persist_f16(Ctx& c, RawTensor& t) { return raw_f16(c, t, true); }

static float raw_scalar(Ctx& c, const std::string& name) {
  RawTensor& t = raw_get(c, name, {1});
  UG_CHECK(hipStreamSynchronize(c.stream));
  if (t.dtype == 1) { float v; UG_CHECK(hipMemcpy(&v, t.dev, 4, hipMemcpyDeviceToHost)); return v; }
  uint16_t h; UG_CHECK(hipMemcpy(&h, t.dev, 2, hipMemcpyDeviceToHost));
  f16 hv; memcpy(&hv, &h, 2);
  return (float)hv;
}

static Norm bind_norm(Ctx& c, const std::string& p, int ch, float eps) {
  Norm n; n.c = ch; n.eps = eps;
  n.g = persist_f16(c, raw_get(c, p + ".weight", {ch}));
  n.b = persist_f16(c, raw_get(c, p + ".bias", {ch}));
  return n;
}
static Lin bind_lin(Ctx& c, const std::string& p, int in, int out, bool bias) {
  Lin l; l.in = in; l.out = out;
  l.w = persist_f16(c, raw_get(c, p + ".weight", {out, in}));
  l.b = bias ? persist_f16(c, raw_get(c, p + ".bias", {out})) : nullptr;
  return l;
}
// rows of several [out_i][in] weights stacked into one [sum out_i][in] matrix (fused q|k|v)
static Lin bind_lin_cat(Ctx& c, const std::vector<std::string>& ps, int in, int out_each, bool bias) {
  Lin l; l.in = in; l.out = out_each * (int)ps.size();
  f16* w = c.persist.get<f16>((long)l.out * in);
  f16* b = bias ? c.persist.get<f16>(l.out) : nullptr;
  for (size_t i = 0; i < ps.size(); ++i) {
    RawTensor& t = raw_get(c, ps[i] + ".weight", {out_each, in});
    const size_t mk = c.persist.mark();
    const f16* src = raw_f16(c, t, false);
    UG_CHECK(hipMemcpyAsync(w + (long)i * out_each * in, src, (size_t)out_each * in * 2, hipMemcpyDeviceToDevice, c.stream));
    UG_CHECK(hipStreamSynchronize(c.stream));
    c.persist.release(mk);
    if (bias) {
      RawTensor& tb = raw_get(c, ps[i] + ".bias", {out_each});
      const size_t mk2 = c.persist.mark();
      const f16* sb = raw_f16(c, tb, false);
      UG_CHECK(hipMemcpyAsync(b + (long)i * out_each, sb, (size_t)out_each * 2, hipMemcpyDeviceToDevice, c.stream));
      UG_CHECK(hipStreamSynchronize(c.stream));
      c.persist.release(mk2);
    }
  }
  l.w = w; l.b = b;
  return l;
}
// GEGLU projection [2*inner][in]: rows re-ordered in blocks of 16 = [8 value rows | 8 gate rows] so that the
// GEMM epilogue finds value and gate of the same output column in one lane.
static Lin bind_geglu(Ctx& c, const std::string& p, int in, int inner) {
  const int out = 2 * inner;
  UG_REQUIRE(inner % 8 == 0, "GEGLU inner dim must be a multiple of 8");
  RawTensor& tw = raw_get(c, p + ".weight", {out, in});
  RawTensor& tb = raw_get(c, p + ".bias", {out});
  std::vector<int> map(out);
  for (int v = 0; v < out; ++v) {
    const int blk = v / 16, wv = v % 16;
    map[v] = wv < 8 ? blk * 8 + wv : inner + blk * 8 + (wv - 8);
  }
  int* dmap = (int*)c.persist.alloc(out * sizeof(int));
  UG_CHECK(hipMemcpy(dmap, map.data(), out * sizeof(int), hipMemcpyHostToDevice));
  Lin l; l.in = in; l.out = out;
  f16* w = c.persist.get<f16>((long)out * in);
  f16* b = c.persist.get<f16>(out);
  const size_t mk = c.persist.mark();
  launch_gather_rows(raw_f16(c, tw, false), w, dmap, out, in, c.stream);
  launch_gather_rows(raw_f16(c, tb, false), b, dmap, out, 1, c.stream);
  UG_CHECK(hipStreamSynchronize(c.stream));
  c.persist.release(mk);
  l.w = w; l.b = b;
  return l;
}
static inline int pad8(int x) { return (x + 7) & ~7; }
// MX-fp8 copy of a bound linear layer's weight (rows are quantised independently along K, so fused / re-ordered rows stay valid)
static void quant_lin(Ctx& c, Lin& l) {
  if (l.in % 128 != 0 || l.out < 64) return;
  unsigned char* w8 = (unsigned char*)c.persist.alloc((size_t)l.out * l.in);
  const long ld = pad256(l.out);
  unsigned* sw = (unsigned*)c.persist.alloc((size_t)(l.in / 128) * ld * 4);
  UG_CHECK(hipMemsetAsync(sw, 0, (size_t)(l.in / 128) * ld * 4, c.stream));
  launch_quant_mx8(l.w, l.in, l.out, l.in, w8, sw, ld, c.stream);
  l.w8 = w8; l.sw8 = sw; l.ld_sw8 = ld;
}
// Conv2d [O][I][k][k] -> [O][k*k][Ipad];  Conv3d [O][I][3][1][1] -> [O][3][Ipad]
static Conv bind_conv(Ctx& c, const std::string& p, int cin, int cout, int kt, int k, bool bias = true, bool upsampler = false) {
  Conv cv; cv.cin = cin; cv.cinp = pad8(cin); cv.cout = cout; cv.kt = kt; cv.ky = k; cv.kx = k;
  RawTensor* t;
  if (kt > 1) t = &raw_get(c, p + ".weight", {cout, cin, kt, 1, 1});
  else t = &raw_get(c, p + ".weight", {cout, cin, k, k});
  const int taps = kt * k * k;
  f16* w = c.persist.get<f16>((long)cout * taps * cv.cinp);
  const size_t mk = c.persist.mark();
  launch_permute_conv_w(raw_f16(c, *t, false), w, cout, cin, taps, cv.cinp, cout, c.stream);
  UG_CHECK(hipStreamSynchronize(c.stream));
  c.persist.release(mk);
  cv.w = w;
  if (upsampler && kt == 1 && k == 3 && !getenv("UG_NO_SUBPIXEL")) {
    f16* w4 = c.persist.get<f16>((long)4 * cout * 4 * cv.cinp);
    const size_t mk2 = c.persist.mark();
    launch_upsample_phase_w(raw_f16(c, *t, false), w4, cout, cin, cv.cinp, c.stream);
    UG_CHECK(hipStreamSynchronize(c.stream));
    c.persist.release(mk2);
    cv.wphase = w4;
  }
  cv.b = bias ? persist_f16(c, raw_get(c, p + ".bias", {cout})) : nullptr;
  // chunk-major K order for every multi-tap conv whose channels come in whole 64-blocks (GemmP::kchunk): taps of a chunk are consecutive
  if (taps > 1 && cv.cinp % 64 == 0) {
    const size_t mk3 = c.persist.mark();
    f16* tmp = c.persist.get<f16>((long)cout * taps * cv.cinp);
    UG_CHECK(hipMemcpyAsync(tmp, w, (size_t)cout * taps * cv.cinp * 2, hipMemcpyDeviceToDevice, c.stream));
    launch_rechunk_conv_w(tmp, w, cout, taps, cv.cinp, c.stream);
    if (cv.wphase) {
      f16* t4 = c.persist.get<f16>((long)4 * cout * 4 * cv.cinp);
      UG_CHECK(hipMemcpyAsync(t4, cv.wphase, (size_t)4 * cout * 4 * cv.cinp * 2, hipMemcpyDeviceToDevice, c.stream));
      launch_rechunk_conv_w(t4, (f16*)cv.wphase, (long)4 * cout, 4, cv.cinp, c.stream);
    }
    UG_CHECK(hipStreamSynchronize(c.stream));
    c.persist.release(mk3);
    cv.kchunk = 1;
  }
  return cv;
}

// K-doubled copies for fp16 hi/lo-pair activations (kernels/wide.hip): [O][taps][Cinp] -> [O][taps][Cinp | Cinp], [N][K] -> [N][K | K]
static Conv dup_conv(Ctx& c, const Conv& cv) {
  Conv d = cv; d.wphase = nullptr;
  const int taps = cv.kt * cv.ky * cv.kx;
  // tap-major rows are [taps][Cinp] -> [taps][Cinp | Cinp]; chunk-major rows are [chunks][taps][64] -> the lo half is simply a second
  // run of chunks, i.e. the whole row twice (a doubled, 64-divisible channel count stays chunk-major)
  const long rows = cv.kchunk ? (long)cv.cout : (long)cv.cout * taps;
  const long cols = cv.kchunk ? (long)taps * cv.cinp : cv.cinp;
  f16* w = c.persist.get<f16>(rows * 2 * cols);
  launch_copy2d(cv.w, cols, w, 2 * cols, rows, (int)cols, c.stream);
  launch_copy2d(cv.w, cols, w + cols, 2 * cols, rows, (int)cols, c.stream);
  // a tap-major base (Cinp % 64 != 0) whose doubled channel count is a multiple of 64 keeps the tap-major layout: the kernels then take
  // the general (non single-tap) path unless it is one chunk, where both orders coincide
  d.w = w; d.cin = 2 * cv.cinp; d.cinp = 2 * cv.cinp;
  return d;
}
static Lin dup_lin(Ctx& c, const Lin& l) {
  Lin d = l;
  f16* w = c.persist.get<f16>((long)l.out * 2 * l.in);
  launch_copy2d(l.w, l.in, w, 2 * l.in, l.out, l.in, c.stream);
  launch_copy2d(l.w, l.in, w + l.in, 2 * l.in, l.out, l.in, c.stream);
  d.w = w; d.in = 2 * l.in;
  return d;
}
static Res2D dup_res2d(Ctx& c, const Res2D& r) {
  Res2D d = r;
  d.c1 = dup_conv(c, r.c1); d.c2 = dup_conv(c, r.c2);
  if (r.has_sc) d.sc = dup_conv(c, r.sc);
  return d;
}

static Res2D bind_res2d(Ctx& c, const std::string& p, int cin, int cout, int temb, float eps) {
  Res2D r;
  r.n1 = bind_norm(c, p + ".norm1", cin, eps);
  r.c1 = bind_conv(c, p + ".conv1", cin, cout, 1, 3);
  if (temb) { r.temb = bind_lin(c, p + ".time_emb_proj", temb, cout, true); r.has_temb = true; }
  r.n2 = bind_norm(c, p + ".norm2", cout, eps);
  r.c2 = bind_conv(c, p + ".conv2", cout, cout, 1, 3);
  if (cin != cout) { r.sc = bind_conv(c, p + ".conv_shortcut", cin, cout, 1, 1); r.has_sc = true; }
  return r;
}
static ResT bind_rest(Ctx& c, const std::string& p, int ch, int temb, float eps) {
  ResT r;
  r.n1 = bind_norm(c, p + ".norm1", ch, eps);
  r.c1 = bind_conv(c, p + ".conv1", ch, ch, 3, 1);
  if (temb) { r.temb = bind_lin(c, p + ".time_emb_proj", temb, ch, true); r.has_temb = true; }
  r.n2 = bind_norm(c, p + ".norm2", ch, eps);
  r.c2 = bind_conv(c, p + ".conv2", ch, ch, 3, 1);
  return r;
}
// UNet: alpha = sigmoid(mix) ("learned_with_images", indicator 0); VAE decoder: "learned" + switch => 1 - sigmoid
static STRes bind_stres(Ctx& c, const std::string& p, int cin, int cout, int temb, float eps, float teps, bool sw) {
  STRes r; r.cin = cin; r.cout = cout;
  r.s = bind_res2d(c, p + ".spatial_res_block", cin, cout, temb, eps);
  r.t = bind_rest(c, p + ".temporal_res_block", cout, temb, teps);
  const float mix = raw_scalar(c, p + ".time_mixer.mix_factor");
  const float a = 1.f / (1.f + expf(-mix));
  r.alpha = sw ? 1.f - a : a;
  return r;
}

static void touch(Ctx& c, const std::string& name, std::initializer_list<long> shape) { raw_get(c, name, shape); }

static Transformer bind_transformer(Ctx& c, const std::string& p, int C, int heads, int cross) {
  Transformer t; t.C = C; t.heads = heads;
  UG_REQUIRE(C / heads == 64 && C % heads == 0, "UNet attention head_dim must be 64");
  t.gn = bind_norm(c, p + ".norm", C, 1e-6f);
  t.proj_in = bind_lin(c, p + ".proj_in", C, C, true);
  t.proj_out = bind_lin(c, p + ".proj_out", C, C, true);
  const std::string b = p + ".transformer_blocks.0", tb = p + ".temporal_transformer_blocks.0";
  t.ln1 = bind_norm(c, b + ".norm1", C, 1e-5f);
  t.qkv1 = bind_lin_cat(c, {b + ".attn1.to_q", b + ".attn1.to_k", b + ".attn1.to_v"}, C, C, false);
  t.o1 = bind_lin(c, b + ".attn1.to_out.0", C, C, true);
  // cross-attention over ONE context token: softmax == 1, so norm2 / to_q / to_k never influence the result.
  touch(c, b + ".norm2.weight", {C}); touch(c, b + ".norm2.bias", {C});
  touch(c, b + ".attn2.to_q.weight", {C, C}); touch(c, b + ".attn2.to_k.weight", {C, cross});
  t.v2 = bind_lin(c, b + ".attn2.to_v", cross, C, false);
  t.o2 = bind_lin(c, b + ".attn2.to_out.0", C, C, true);
  t.ln3 = bind_norm(c, b + ".norm3", C, 1e-5f);
  t.ff1 = bind_geglu(c, b + ".ff.net.0.proj", C, 4 * C);
  t.ff2 = bind_lin(c, b + ".ff.net.2", 4 * C, C, true);
  t.ln_in = bind_norm(c, tb + ".norm_in", C, 1e-5f);
  t.ffin1 = bind_geglu(c, tb + ".ff_in.net.0.proj", C, 4 * C);
  t.ffin2 = bind_lin(c, tb + ".ff_in.net.2", 4 * C, C, true);
  t.tln1 = bind_norm(c, tb + ".norm1", C, 1e-5f);
  t.tqkv = bind_lin_cat(c, {tb + ".attn1.to_q", tb + ".attn1.to_k", tb + ".attn1.to_v"}, C, C, false);
  t.to1 = bind_lin(c, tb + ".attn1.to_out.0", C, C, true);
  touch(c, tb + ".norm2.weight", {C}); touch(c, tb + ".norm2.bias", {C});
  touch(c, tb + ".attn2.to_q.weight", {C, C}); touch(c, tb + ".attn2.to_k.weight", {C, cross});
  t.tv2 = bind_lin(c, tb + ".attn2.to_v", cross, C, false);
  t.to2 = bind_lin(c, tb + ".attn2.to_out.0", C, C, true);
  t.tln3 = bind_norm(c, tb + ".norm3", C, 1e-5f);
  t.tff1 = bind_geglu(c, tb + ".ff.net.0.proj", C, 4 * C);
  t.tff2 = bind_lin(c, tb + ".ff.net.2", 4 * C, C, true);
  t.tpe1 = bind_lin(c, p + ".time_pos_embed.linear_1", C, 4 * C, true);
  t.tpe2 = bind_lin(c, p + ".time_pos_embed.linear_2", 4 * C, C, true);
  const float mix = raw_scalar(c, p + ".time_mixer.mix_factor");
  t.alpha = 1.f / (1.f + expf(-mix));
  for (Lin* l : {&t.proj_in, &t.proj_out, &t.qkv1, &t.o1, &t.ff1, &t.ff2, &t.ffin1, &t.ffin2, &t.tqkv, &t.to1, &t.tff1, &t.tff2}) quant_lin(c, *l);
  return t;
}

void bind_unet(Ctx& c, const UNetCfg& cfg, const std::string& pre) {
  UNet& u = c.unet; u = UNet(); u.cfg = cfg;
  const int n = cfg.nlev, temb = cfg.boc[0] * 4;
  u.conv_in = bind_conv(c, pre + "conv_in", cfg.in_ch, cfg.boc[0], 1, 3);
  u.te1 = bind_lin(c, pre + "time_embedding.linear_1", cfg.boc[0], temb, true);
  u.te2 = bind_lin(c, pre + "time_embedding.linear_2", temb, temb, true);
  u.ae1 = bind_lin(c, pre + "add_embedding.linear_1", cfg.proj_in_dim, temb, true);
  u.ae2 = bind_lin(c, pre + "add_embedding.linear_2", temb, temb, true);
  u.down.resize(n);
  int ch = cfg.boc[0];
  for (int i = 0; i < n; ++i) {
    const float eps = cfg.has_attn[i] ? cfg.eps_xattn : cfg.eps_down;
    for (int j = 0; j < cfg.layers; ++j) {
      const std::string p = pre + "down_blocks." + std::to_string(i);
      u.down[i].res.push_back(bind_stres(c, p + ".resnets." + std::to_string(j), j == 0 ? ch : cfg.boc[i], cfg.boc[i], temb, eps, eps, false));
      if (cfg.has_attn[i])
        u.down[i].attn.push_back(bind_transformer(c, p + ".attentions." + std::to_string(j), cfg.boc[i], cfg.heads[i], cfg.cross_dim));
    }
    if (i != n - 1) {
      u.down[i].down = bind_conv(c, pre + "down_blocks." + std::to_string(i) + ".downsamplers.0.conv", cfg.boc[i], cfg.boc[i], 1, 3);
      u.down[i].has_down = true;
    }
    ch = cfg.boc[i];
  }
  const int cm = cfg.boc[n - 1];
  u.mid0 = bind_stres(c, pre + "mid_block.resnets.0", cm, cm, temb, cfg.eps_mid, cfg.eps_mid, false);
  u.mid_attn = bind_transformer(c, pre + "mid_block.attentions.0", cm, cfg.heads[n - 1], cfg.cross_dim);
  u.mid1 = bind_stres(c, pre + "mid_block.resnets.1", cm, cm, temb, cfg.eps_mid, cfg.eps_mid, false);
  u.up.resize(n);
  int out = cfg.boc[n - 1];
  const int L = cfg.layers + 1;
  for (int i = 0; i < n; ++i) {
    const int prev = out; out = cfg.boc[n - 1 - i];
    const int cin = cfg.boc[n - 1 - std::min(i + 1, n - 1)];
    const int lev = n - 1 - i;
    for (int j = 0; j < L; ++j) {
      const int skip = (j == L - 1) ? cin : out, rin = (j == 0) ? prev : out;
      const std::string p = pre + "up_blocks." + std::to_string(i);
      u.up[i].res.push_back(bind_stres(c, p + ".resnets." + std::to_string(j), rin + skip, out, temb, cfg.eps_up, cfg.eps_up, false));
      if (cfg.has_attn[lev])
        u.up[i].attn.push_back(bind_transformer(c, p + ".attentions." + std::to_string(j), out, cfg.heads[lev], cfg.cross_dim));
    }
    if (i != n - 1) {
      u.up[i].up = bind_conv(c, pre + "up_blocks." + std::to_string(i) + ".upsamplers.0.conv", out, out, 1, 3, true, true);
      u.up[i].has_up = true;
    }
  }
  u.norm_out = bind_norm(c, pre + "conv_norm_out", cfg.boc[0], 1e-5f);
  u.conv_out = bind_conv(c, pre + "conv_out", cfg.boc[0], cfg.out_ch, 1, 3);
  // collect the res-blocks that consume the timestep embedding (pointers into the now-stable vectors)
  auto add = [&](STRes& r) {
    if (r.s.has_temb) { r.s.tidx = (int)u.temb_s.size(); u.temb_s.push_back(&r.s); }
    if (r.t.has_temb) { r.t.tidx = (int)u.temb_t.size(); u.temb_t.push_back(&r.t); }
  };
  for (auto& d : u.down) for (auto& r : d.res) add(r);
  add(u.mid0); add(u.mid1);
  for (auto& d : u.up) for (auto& r : d.res) add(r);
  u.bound = true;
}

static VAttn bind_vattn(Ctx& c, const std::string& p, int C) {
  VAttn a; a.C = C;
  a.gn = bind_norm(c, p + ".group_norm", C, 1e-6f);
  a.qkv = bind_lin_cat(c, {p + ".to_q", p + ".to_k", p + ".to_v"}, C, C, true);
  a.out = bind_lin(c, p + ".to_out.0", C, C, true);
  return a;
}

static Res2D bind_res2d(Ctx& c, const std::string& p, int cin, int cout, int temb, float eps);
// temporal_decoder: AutoencoderKLTemporalDecoder (SVD / DepthCrafter); otherwise the plain SD AutoencoderKL decoder + post_quant_conv
void bind_vae_into(Ctx& c, VAE& v, const VAECfg& cfg, const std::string& pre, bool temporal_decoder) {
  v = VAE(); v.cfg = cfg;
  const int n = cfg.nlev;
  v.e_in = bind_conv(c, pre + "encoder.conv_in", cfg.in_ch, cfg.boc[0], 1, 3);
  v.edown.resize(n);
  int ch = cfg.boc[0];
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < cfg.layers; ++j)
      v.edown[i].res.push_back(bind_res2d(c, pre + "encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j),
                                          j == 0 ? ch : cfg.boc[i], cfg.boc[i], 0, 1e-6f));
    if (i != n - 1) {
      v.edown[i].down = bind_conv(c, pre + "encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv", cfg.boc[i], cfg.boc[i], 1, 3);
      v.edown[i].has_down = true;
    }
    ch = cfg.boc[i];
  }
  v.emid0 = bind_res2d(c, pre + "encoder.mid_block.resnets.0", ch, ch, 0, 1e-6f);
  v.eattn = bind_vattn(c, pre + "encoder.mid_block.attentions.0", ch);
  v.emid1 = bind_res2d(c, pre + "encoder.mid_block.resnets.1", ch, ch, 0, 1e-6f);
  v.e_norm = bind_norm(c, pre + "encoder.conv_norm_out", ch, 1e-6f);
  v.e_out = bind_conv(c, pre + "encoder.conv_out", ch, 2 * cfg.lat, 1, 3);
  v.quant = bind_conv(c, pre + "quant_conv", 2 * cfg.lat, 2 * cfg.lat, 1, 1);
  v.d_in = bind_conv(c, pre + "decoder.conv_in", cfg.lat, cfg.boc[n - 1], 1, 3);
  v.dattn = bind_vattn(c, pre + "decoder.mid_block.attentions.0", cfg.boc[n - 1]);
  v.d_norm = bind_norm(c, pre + "decoder.conv_norm_out", cfg.boc[0], 1e-6f);
  v.d_out = bind_conv(c, pre + "decoder.conv_out", cfg.boc[0], cfg.out_ch, 1, 3);
  if (!temporal_decoder) {
    v.post_quant = bind_conv(c, pre + "post_quant_conv", cfg.lat, cfg.lat, 1, 1);
    v.d2_mid.push_back(bind_res2d(c, pre + "decoder.mid_block.resnets.0", cfg.boc[n - 1], cfg.boc[n - 1], 0, 1e-6f));
    v.d2_mid.push_back(bind_res2d(c, pre + "decoder.mid_block.resnets.1", cfg.boc[n - 1], cfg.boc[n - 1], 0, 1e-6f));
    v.dup.resize(n); v.d2_up.resize(n);
    int out2 = cfg.boc[n - 1];
    for (int i = 0; i < n; ++i) {
      const int prev = out2; out2 = cfg.boc[n - 1 - i];
      for (int j = 0; j < cfg.layers + 1; ++j)
        v.d2_up[i].push_back(bind_res2d(c, pre + "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j),
                                        j == 0 ? prev : out2, out2, 0, 1e-6f));
      if (i != n - 1) {
        v.dup[i].up = bind_conv(c, pre + "decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", out2, out2, 1, 3, true, true);
        v.dup[i].has_up = true;
      }
    }
    v.dec2d = true;
  }
  if (temporal_decoder) {
  for (int j = 0; j < cfg.layers; ++j)
    v.dmid.push_back(bind_stres(c, pre + "decoder.mid_block.resnets." + std::to_string(j), cfg.boc[n - 1], cfg.boc[n - 1], 0, 1e-6f, 1e-5f, true));
  v.dup.resize(n);
  int out = cfg.boc[n - 1];
  for (int i = 0; i < n; ++i) {
    const int prev = out; out = cfg.boc[n - 1 - i];
    for (int j = 0; j < cfg.layers + 1; ++j)
      v.dup[i].res.push_back(bind_stres(c, pre + "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j),
                                        j == 0 ? prev : out, out, 0, 1e-6f, 1e-5f, true));
    if (i != n - 1) {
      v.dup[i].up = bind_conv(c, pre + "decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", out, out, 1, 3, true, true);
      v.dup[i].has_up = true;
    }
  }
  UG_REQUIRE(cfg.out_ch == 3, "time_conv_out kernel is written for 3 output channels");
  v.tco_w = persist_f16(c, raw_get(c, pre + "decoder.time_conv_out.weight", {3, 3, 3, 1, 1}));
  v.tco_b = persist_f16(c, raw_get(c, pre + "decoder.time_conv_out.bias", {3}));
  }
  // float32-grade encoder (reference: force_upcast): K-doubled weights, +2x the encoder's 34 M parameters
  v.edown_w.resize(n);
  for (int i = 0; i < n; ++i) {
    for (auto& r : v.edown[i].res) v.edown_w[i].res.push_back(dup_res2d(c, r));
    v.edown_w[i].has_down = v.edown[i].has_down;
    if (v.edown[i].has_down) v.edown_w[i].down = dup_conv(c, v.edown[i].down);
  }
  v.emid0_w = dup_res2d(c, v.emid0); v.emid1_w = dup_res2d(c, v.emid1);
  v.eattn_w = v.eattn; v.eattn_w.qkv = dup_lin(c, v.eattn.qkv); v.eattn_w.out = dup_lin(c, v.eattn.out);
  v.e_out_w = dup_conv(c, v.e_out); v.quant_w = dup_conv(c, v.quant);
  UG_CHECK(hipStreamSynchronize(c.stream));
  v.wide_bound = true;
  v.bound = true;
}
void bind_vae(Ctx& c, const VAECfg& cfg, const std::string& pre) { bind_vae_into(c, c.vae, cfg, pre, true); }

void bind_clip(Ctx& c, const CLIPCfg& cfg, const std::string& pre) {
  CLIP& m = c.clip; m = CLIP(); m.cfg = cfg;
  const int d = cfg.hidden, P = cfg.patch, ntok = (cfg.image / P) * (cfg.image / P) + 1;
  UG_REQUIRE((d / cfg.heads) % 8 == 0, "CLIP head_dim must be a multiple of 8");
  const std::string v = pre + "vision_model.";
  m.cls = persist_f16(c, raw_get(c, v + "embeddings.class_embedding", {d}));
  m.pos = persist_f16(c, raw_get(c, v + "embeddings.position_embedding.weight", {ntok, d}));
  {
    RawTensor& t = raw_get(c, v + "embeddings.patch_embedding.weight", {d, 3, P, P});
    const int K = 3 * P * P; m.patch_k = pad8(K);
    f16* w = c.persist.get<f16>((long)d * m.patch_k);
    const size_t mk = c.persist.mark();
    launch_permute_conv_w(raw_f16(c, t, false), w, d, K, 1, m.patch_k, d, c.stream);
    UG_CHECK(hipStreamSynchronize(c.stream));
    c.persist.release(mk);
    m.patch_w = w;
  }
  m.pre = bind_norm(c, v + "pre_layrnorm", d, cfg.eps);
  for (int i = 0; i < cfg.layers; ++i) {
    const std::string p = v + "encoder.layers." + std::to_string(i);
    CLIPLayer l;
    l.ln1 = bind_norm(c, p + ".layer_norm1", d, cfg.eps);
    l.qkv = bind_lin_cat(c, {p + ".self_attn.q_proj", p + ".self_attn.k_proj", p + ".self_attn.v_proj"}, d, d, true);
    l.out = bind_lin(c, p + ".self_attn.out_proj", d, d, true);
    l.ln2 = bind_norm(c, p + ".layer_norm2", d, cfg.eps);
    l.fc1 = bind_lin(c, p + ".mlp.fc1", d, cfg.inter, true);
    l.fc2 = bind_lin(c, p + ".mlp.fc2", cfg.inter, d, true);
    m.layers.push_back(l);
  }
  m.post = bind_norm(c, v + "post_layernorm", d, cfg.eps);
  m.proj = bind_lin(c, pre + "visual_projection", d, cfg.proj, false);
  m.bound = true;
}

void finish_binding(Ctx& c, const std::string& prefix) {
  UG_CHECK(hipStreamSynchronize(c.stream));
  std::vector<std::string> unused;
  for (auto it = c.raw.begin(); it != c.raw.end();) {
    if (it->first.compare(0, prefix.size(), prefix) != 0) { ++it; continue; }
    const bool ignorable = it->first.size() >= 12 && it->first.compare(it->first.size() - 12, 12, "position_ids") == 0;
    if (!it->second.used && !ignorable) unused.push_back(it->first);
    (void)hipFree(it->second.dev);
    it = c.raw.erase(it);
  }
  if (!unused.empty()) {
    std::string s = "unexpected tensors in state dict (architecture restatement mismatch): ";
    for (size_t i = 0; i < unused.size() && i < 8; ++i) s += unused[i] + " ";
    throw std::runtime_error(s + "(" + std::to_string(unused.size()) + " total)");
  }
}

// ======================================================================= graphs
static void unfused_attention(Ctx& c, const f16* qkv, long ldqkv, int B, int S, int heads, int d, f16* out, long ldo) {
  // qkv [B*S, 3*heads*d]; batches (frame, head); scores fp32 -> softmax -> P.V with V^T built by a transpose
  const int Spad = pad8(S), C = heads * d;
  const size_t mk = c.ws.mark();
  float* sc = c.ws.get<float>((long)B * heads * S * Spad);
  f16* P = c.ws.get<f16>((long)B * heads * S * Spad);
  f16* Vt = c.ws.get<f16>((long)B * heads * d * Spad);
  GemmP p; memset(&p, 0, sizeof(p));
  p.A0 = qkv; p.C0 = (int)ldqkv; p.M = S; p.N = S; p.K = d; p.W = qkv + C; p.ldw = ldqkv;
  p.c0 = 1.0f / sqrtf((float)d); p.Out = sc; p.ldo = Spad; p.flags = UG_F_OUT_F32;
  p.nb_inner = heads; p.sA_o = (long)S * ldqkv; p.sA_i = d; p.sW_o = (long)S * ldqkv; p.sW_i = d;
  p.sO_o = (long)heads * S * Spad; p.sO_i = (long)S * Spad;
  run_gemm(c, p, B * heads, "gemm_attn_qk");
  {
    ProfScope ps(c, "softmax_rows", 0, (double)B * heads * S * Spad * 6.0);
    launch_softmax_rows(sc, Spad, P, Spad, (long)B * heads * S, S, c.stream);
  }
  {
    ProfScope ps(c, "transpose", 0, (double)B * heads * S * d * 4.0);
    launch_transpose(qkv + 2 * C, ldqkv, (long)S * ldqkv, Vt, Spad, (long)heads * d * Spad, B * heads, S, d, heads, d,
                     (long)d * Spad, c.stream);
  }
  GemmP q; memset(&q, 0, sizeof(q));
  q.A0 = P; q.C0 = Spad; q.M = S; q.N = d; q.K = Spad; q.W = Vt; q.ldw = Spad; q.c0 = 1.f;
  q.Out = out; q.ldo = ldo; q.nb_inner = heads;
  q.sA_o = (long)heads * S * Spad; q.sA_i = (long)S * Spad; q.sW_o = (long)heads * d * Spad; q.sW_i = (long)d * Spad;
  q.sO_o = (long)S * ldo; q.sO_i = d;
  run_gemm(c, q, B * heads, "gemm_attn_pv");
  c.ws.release(mk);
}

void test_unfused_attention(Ctx& c, const f16* qkv, long ld, int B, int S, int H, int d, f16* out, long ldo) {
  unfused_attention(c, qkv, ld, B, S, H, d, out, ldo);
}

// ResnetBlock2D over a (virtual concat) input; out [M, cout] must be pre-allocated
static void res2d_forward(Ctx& c, const Res2D& r, const f16* x0, int C0, const f16* x1, int C1, int T, int h, int w,
                          int G, const f16* tproj, f16* out, StatPart* out_stats = nullptr, const StatPart* in_stats = nullptr) {
  const long M = (long)T * h * w;
  const int cin = C0 + C1, cout = r.c1.cout;
  const size_t mk = c.ws.mark();
  f16* a = c.ws.get<f16>(M * cin);
  groupnorm(c, x0, C0, x1, C1, T, h * w, G, r.n1, 0, 1, a, C1 == 0 ? in_stats : nullptr);
  f16* hb = c.ws.get<f16>(M * cout);
  StatPart sh = stat_alloc(c, M, cout, T, h * w, G, 0);      // the statistics of the second GroupNorm come out of the first convolution's epilogue
  { Epi e; e.bias2 = tproj; e.so = &sh; conv(c, a, cin, nullptr, 0, T, h, w, r.c1, 1, 1, 1, 1, hb, e); }
  f16* b = c.ws.get<f16>(M * cout);
  groupnorm(c, hb, cout, nullptr, 0, T, h * w, G, r.n2, 0, 1, b, &sh);
  const f16* res = x0;
  if (r.has_sc) {
    f16* scb = hb;   // hb is dead after the second GroupNorm
    conv(c, x0, C0, x1, C1, T, h, w, r.sc, 1, 0, 0, 1, scb);
    res = scb;
  } else {
    UG_REQUIRE(C1 == 0, "identity shortcut needs a single source");
  }
  { Epi e; e.R1 = res; e.ldr1 = cout; e.so = out_stats; conv(c, b, cout, nullptr, 0, T, h, w, r.c2, 1, 1, 1, 1, out, e); }
  c.ws.release(mk);
}

// in_stats: epilogue statistics of x0 (single source) from whatever produced it; out_stats: filled with those of the block's output (the caller hands
// them to the next block's / the transformer's first GroupNorm) - allocated next to `out`, same lifetime
static f16* stres_forward(Ctx& c, const STRes& rb, const f16* x0, int C0, const f16* x1, int C1, int T, int h, int w,
                          int G, const f16* tproj_s, const f16* tproj_t, StatPart* out_stats = nullptr, const StatPart* in_stats = nullptr) {
  const long M = (long)T * h * w;
  const int cout = rb.cout;
  f16* out = c.ws.get<f16>(M * cout);
  if (out_stats) *out_stats = stat_alloc(c, M, cout, T, h * w, G, 0);
  const size_t mk = c.ws.mark();
  f16* xs = c.ws.get<f16>(M * cout);
  StatPart sx = stat_alloc(c, M, cout, T, h * w, G, 1), sh = stat_alloc(c, M, cout, T, h * w, G, 1);
  res2d_forward(c, rb.s, x0, C0, x1, C1, T, h, w, G, tproj_s, xs, &sx, in_stats);
  f16* a = c.ws.get<f16>(M * cout);
  groupnorm(c, xs, cout, nullptr, 0, T, h * w, G, rb.t.n1, 1, 1, a, &sx);
  f16* hb = c.ws.get<f16>(M * cout);
  { Epi e; e.bias2 = tproj_t; e.so = &sh; conv(c, a, cout, nullptr, 0, T, h, w, rb.t.c1, 1, 0, 0, 1, hb, e); }
  groupnorm(c, hb, cout, nullptr, 0, T, h * w, G, rb.t.n2, 1, 1, a, &sh);
  // blend: alpha*xs + (1-alpha)*(xs + conv) = xs + (1-alpha)*conv
  { Epi e; e.c0 = 1.f - rb.alpha; e.R1 = xs; e.ldr1 = cout; e.c1 = 1.f; e.so = out_stats; conv(c, a, cout, nullptr, 0, T, h, w, rb.t.c2, 1, 0, 0, 1, out, e); }
  c.ws.release(mk);
  return out;
}

// GEGLU feed-forward pair out = ff2(geglu(ff1(a))) * c0 + c1*R1 + c2*R2.  When the [M, 4C] intermediate is larger than
// what survives in the 256 MiB Infinity Cache between the two GEMMs (level 0: 197 MB), the pair CAN run in row chunks so
// each chunk's intermediate is consumed while still cache resident (rows are independent in both GEMMs).  Measured on
// MI355X: -3 % end to end (the smaller launches lose more than residency gains), so it is opt-in (UG_FF_CHUNK=1).
static bool ff_pair_fusable(const Ctx& c, long M, const Lin& f1, const Lin& f2, const Epi& e2) {
  const int C4 = f1.out / 2, C = f2.out;
  return (c.ff_fused & 1) && !c.fp8_linears && ff_fused_supported(C) && f1.in == C && C4 == 4 * C && M >= 32768 && f1.b && !e2.act && !e2.flags &&
         (!e2.R1 || !e2.ldr1 || e2.ldr1 == C) && (!e2.R2 || !e2.ldr2 || e2.ldr2 == C) && !e2.bias2;
}
// pre-norm handed to the fused kernel (ln_ff): LayerNorm parameters + the broadcast row added to the residual stream before it
struct PreNorm { const f16* g; const f16* b; float eps; const f16* addvec; long rows_per_vec; };
// The fused kernel runs one 128-row tile per CU at a time: 600 tiles (level 0 of the clip) are 2.34 rounds of 256, and the third round
// keeps 88 CUs busy while 168 wait.  Rows that would fall into such a thin last round (<= 160 tiles) go through the two-GEMM path
// instead - the fused kernel then runs whole rounds only (ff_fused_rows).  (Round 4 ran those rows through the same kernel on 64-row tiles, 176 tiles on
// 176 CUs: 59.9 us against 55.2 + the LayerNorm launch for 11264 rows, bit-identical, 961.7 vs 961.1 ms per clip - every workgroup streams the whole
// 4.9 MB of W1 | W2 whatever its row count, so halving the rows does not halve the tile.  No gain, removed.  Likewise a persistent grid of 200 workgroups
// x 3 tiles instead of 256 x 2 + the tail launches: 951.5 against 944 ms per clip - a tile costs the same whether 200 or 256 run beside it.)
static long ff_fused_rows(const Ctx& c, long M) {
  const long ntile = (M + 127) / 128, full = ntile / 256 * 256, rem = ntile - full;
  static const bool nosplit = getenv("UG_FF_NOSPLIT") != nullptr;
  // (co-scheduled contexts: the CUs a thin last round leaves idle run the other clip's kernels - one launch for all rows, 28.4 -> 28.7 frames/s with two clips in flight)
  return (full > 0 && rem > 0 && rem <= 160 && !nosplit && !c.cosched) ? full * 128 : M;
}
static void ff_pair(Ctx& c, const f16* a, long M, const Lin& f1, const Lin& f2, f16* mid, f16* out, const Epi& e2, const QAct* q = nullptr,
                    const PreNorm* pre = nullptr, bool no_fuse = false) {
  const int C4 = f1.out / 2, C = f2.out;
  UG_REQUIRE(!pre || ff_pair_fusable(c, M, f1, f2, e2), "ff_pair: pre-norm only with the fused kernel");
  // narrow blocks (level 0: C = 320): one fused kernel, the [M, 4C] intermediate never leaves the CU (kernels/ff_fused.hip)
  if (ff_pair_fusable(c, M, f1, f2, e2) && !no_fuse) {
    FFusedP p; memset(&p, 0, sizeof(p));
    p.X = a; p.W1 = f1.w; p.b1 = f1.b; p.W2 = f2.w; p.b2 = f2.b; p.R1 = e2.R1; p.R2 = e2.R2; p.c0 = e2.c0; p.c1 = e2.c1; p.c2 = e2.c2;
    p.Out = out; p.M = (int)M; p.C = C; p.zero = c.zero; p.variant = c.ff_variant;
    if (pre) { p.ln_g = pre->g; p.ln_b = pre->b; p.ln_eps = pre->eps; p.addvec = pre->addvec; p.rows_per_vec = (int)pre->rows_per_vec; }
    char nm[64];
    if (c.prof_on && c.prof_shapes) snprintf(nm, sizeof(nm), "gemm_ff_fused:%ldx%d", M, C); else snprintf(nm, sizeof(nm), "gemm_ff_fused");
    ProfScope ps(c, nm, 2.0 * M * (double)(2 * C4) * C + 2.0 * M * (double)C * C4,
                 2.0 * ((double)M * C * (2 + (e2.R1 ? 1 : 0) + (e2.R2 ? 1 : 0)) + 3.0 * (double)C4 * C));
    launch_ff_fused(p, c.stream);
    return;
  }
  const long bytes = M * C4 * 2;
  int nchunk = 1;
  // row chunking is an opt-in experiment; never together with a pre-quantised MX-fp8 input (q: the LayerNorm then wrote ONLY the fp8 image of
  // `a`, whole rows x all K steps - a chunk would need offsets into it, and the fp16 `a` the chunks read was never written)
  if (bytes > (96L << 20) && getenv("UG_FF_CHUNK") && !q) nchunk = (int)((bytes + (48L << 20) - 1) / (48L << 20));
  const long rows = ((M + nchunk - 1) / nchunk + 255) / 256 * 256;
  for (long r0 = 0; r0 < M; r0 += rows) {
    const long m = std::min(rows, M - r0);
    { Epi e; e.flags = UG_F_GEGLU; if (q && nchunk == 1) { e.a8 = q->a8; e.sa8 = q->sa; e.ld_sa8 = q->ld; } linear(c, a + r0 * f1.in, m, f1, mid + r0 * C4, e); }
    Epi e = e2;
    if (e.R1) e.R1 += r0 * (e.ldr1 ? e.ldr1 : C);
    if (e.R2) e.R2 += r0 * (e.ldr2 ? e.ldr2 : C);
    linear(c, mid + r0 * C4, m, f2, out + r0 * C, e);
  }
}

// out[M, l.out] = LayerNorm(x) . W^T (+ bias): the LayerNorm launch (writing t1) followed by the GEMM.  (Round 3 also had ONE kernel with the
// activation tile resident in LDS for the narrow level; 130 - 137 us against 113 - 117 for the pair at 76800 x 960 x 320 - removed in round 4.  Round 4
// tried it again inside the streaming GEMM (kernels/gemm_stream.hip): correct, +2.5 ms per clip - removed as well.)
static void ln_linear(Ctx& c, const f16* x, long M, const Norm& ln, f16* t1, const Lin& l, f16* out, const QAct* q) {
  layernorm(c, x, M, ln, t1, nullptr, 1, nullptr, q);
  Epi e;
  if (q) { e.a8 = q->a8; e.sa8 = q->sa; e.ld_sa8 = q->ld; }
  linear(c, t1, M, l, out, e);
}

// y = FF(LayerNorm(x')) combined with the residual stream x' = x (+ addvec row): on the narrow level the LayerNorm runs inside the fused
// feed-forward kernel - neither LayerNorm(x') nor x' are materialised (e2.R1 must be the stream: xout when addvec is given, else x);
// otherwise the LayerNorm launch (writing t1 and xout) followed by ff_pair.
static void ln_ff(Ctx& c, const f16* x, long M, const Norm& ln, const f16* addvec, long rows_per_vec, f16* xout, f16* t1,
                  const Lin& f1, const Lin& f2, f16* mid, f16* out, const Epi& e2, const QAct* q) {
  const int C = f2.out;
  if ((c.ff_fused & 2) && ff_pair_fusable(c, M, f1, f2, e2) && ln.g && ln.b && e2.R1 == (addvec ? xout : x) && (!addvec || xout)) {
    const long M1 = ff_fused_rows(c, M);     // whole rounds of the fused kernel ...
    Epi e = e2; e.R1 = x;
    const PreNorm pre = {ln.g, ln.b, ln.eps, addvec, rows_per_vec};
    ff_pair(c, x, M1, f1, f2, mid, out, e, nullptr, &pre);
    if (M1 < M) {                          // ... the thin last round's rows: LayerNorm launch + two GEMMs
      const long o = M1 * C;
      layernorm(c, x + o, M - M1, ln, t1 + o, addvec, rows_per_vec, addvec ? xout + o : nullptr, nullptr, M1);
      Epi et = e2; et.R1 = (addvec ? xout : x) + o; if (et.R2) et.R2 += o;
      ff_pair(c, t1 + o, M - M1, f1, f2, mid, out + o, et, nullptr, nullptr, true);
    }
    return;
  }
  layernorm(c, x, M, ln, t1, addvec, rows_per_vec, xout, q);
  if (ff_pair_fusable(c, M, f1, f2, e2)) {
    const long M1 = ff_fused_rows(c, M), o = M1 * C;
    ff_pair(c, t1, M1, f1, f2, mid, out, e2, q);
    if (M1 < M) {
      Epi et = e2; if (et.R1) et.R1 += o; if (et.R2) et.R2 += o;
      ff_pair(c, t1 + o, M - M1, f1, f2, mid, out + o, et, q, nullptr, true);
    }
    return;
  }
  ff_pair(c, t1, M, f1, f2, mid, out, e2, q);
}

static f16* transformer_forward(Ctx& c, const Transformer& tr, const f16* x, int T, int h, int w, int G, const StatPart* in_stats = nullptr) {
  const int C = tr.C, HW = h * w;
  const long M = (long)T * HW;
  f16* out = c.ws.get<f16>(M * C);
  const size_t mk = c.ws.mark();
  f16* t1 = c.ws.get<f16>(M * C);
  groupnorm(c, x, C, nullptr, 0, T, HW, G, tr.gn, 0, 0, t1, in_stats);
  f16* h0 = c.ws.get<f16>(M * C);
  // fp8 linear path: the LayerNorms feeding a linear layer write MX-fp8 directly (no fp16 copy, no separate quantiser pass)
  const bool q8 = c.fp8_linears && !getenv("UG_NO_LNQ") && C % 128 == 0 && M >= 256 && tr.qkv1.w8 && tr.ff1.w8 && tr.ffin1.w8 && tr.tqkv.w8 && tr.tff1.w8;
  QAct qa; if (q8) qa = qact_alloc(c, M, C);
  const QAct* qp = q8 ? &qa : nullptr;
  auto qepi = [&](Epi e) { if (q8) { e.a8 = qa.a8; e.sa8 = qa.sa; e.ld_sa8 = qa.ld; } return e; };
  // ---- spatial block
  f16* qkv = c.ws.get<f16>(M * 3 * C);
  linear(c, t1, M, tr.proj_in, h0);
  ln_linear(c, h0, M, tr.ln1, t1, tr.qkv1, qkv, qp);
  f16* ao = c.ws.get<f16>(M * C);
  {
    FlashP p; p.Q = qkv; p.K = qkv + C; p.V = qkv + 2 * C; p.ldq = p.ldk = p.ldv = 3 * C; p.O = ao; p.ldo = C;
    p.B = T; p.H = tr.heads; p.S = HW; p.scale = 0.125f; p.variant = c.flash_variant;
    char nm[96];
    if (c.prof_on && c.prof_shapes) snprintf(nm, sizeof(nm), "flash_attn:B%dxH%dxS%d", T, tr.heads, HW);
    else snprintf(nm, sizeof(nm), "flash_attn");
    ProfScope ps(c, nm, 4.0 * T * tr.heads * (double)HW * HW * 64, 0);
    launch_flash_attn64(p, c.stream);
  }
  f16* h1 = c.ws.get<f16>(M * C);
  f16* h2 = c.ws.get<f16>(M * C);
  f16* ffm = c.ws.get<f16>(M * 4 * C);
  f16* hs = c.ws.get<f16>(M * C);
  { Epi e; e.R1 = h0; linear(c, ao, M, tr.o1, h1, e); }
  { Epi e; e.R1 = h2; ln_ff(c, h1, M, tr.ln3, tr.cross_sp, HW, h2, t1, tr.ff1, tr.ff2, ffm, hs, e, qp); }
  // ---- temporal block (token order kept; only the attention gathers over frames)
  f16* xm = h0;   // h0 is dead
  f16* g1 = h1;   // h1 is dead
  { Epi e; e.R1 = xm; ln_ff(c, hs, M, tr.ln_in, tr.frame_emb, HW, xm, t1, tr.ffin1, tr.ffin2, ffm, g1, e, qp); }
  ln_linear(c, g1, M, tr.tln1, t1, tr.tqkv, qkv, qp);
  {
    TemporalAttnP p; p.Q = qkv; p.K = qkv + C; p.V = qkv + 2 * C; p.ld = 3 * C; p.O = ao; p.ldo = C;
    p.T = T; p.HW = HW; p.H = tr.heads; p.scale = 0.125f;
    ProfScope ps(c, "temporal_attn", 4.0 * HW * tr.heads * (double)T * T * 64, (double)M * C * 2.0 * 4.0);
    launch_temporal_attn64(p, c.stream);
  }
  f16* g2 = h2;   // h2 is dead
  f16* g3 = xm;   // xm is dead
  f16* mix = g1;
  {
    Epi e; e.c0 = 1.f - tr.alpha; e.R1 = g3; e.c1 = 1.f - tr.alpha; e.R2 = hs; e.c2 = tr.alpha;
    { Epi e1; e1.R1 = g1; linear(c, ao, M, tr.to1, g2, e1); }
    ln_ff(c, g2, M, tr.tln3, tr.cross_tm, M, g3, t1, tr.tff1, tr.tff2, ffm, mix, e, qp);
  }
  { Epi e; e.R1 = x; linear(c, mix, M, tr.proj_out, out, e); }
  c.ws.release(mk);
  return out;
}

static void host_sinusoid(const float* t, int n, int dim, std::vector<f16>& out) {
  const int half = dim / 2;
  out.resize((size_t)n * dim);
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < half; ++k) {
      const float ex = expf(-logf(10000.0f) * (float)k / (float)half);
      const float a = t[i] * ex;
      out[(size_t)i * dim + k] = (f16)cosf(a);          // flip_sin_to_cos=True => [cos | sin]
      out[(size_t)i * dim + half + k] = (f16)sinf(a);
    }
}
static f16* upload_f16(Ctx& c, const std::vector<f16>& v) {
  f16* d = c.ws.get<f16>((long)v.size());
  UG_CHECK(hipMemcpyAsync(d, v.data(), v.size() * 2, hipMemcpyHostToDevice, c.stream));
  UG_CHECK(hipStreamSynchronize(c.stream));   // v may go out of scope
  return d;
}

// Everything that depends on the clip (CLIP embeddings, frame count) or the timestep table but not on the
// latents is computed ONCE here: per-step time-embedding projections of all res-blocks, per-transformer
// frame-position embeddings and the single-token cross-attention outputs.  The denoise loop then runs with
// zero host synchronisation.  Allocations come from c.ws and stay live until the caller releases its mark.
void unet_prepare(Ctx& c, int T, const f16* clip_emb, const float* timesteps, int nsteps) {
  UNet& u = c.unet;
  UG_REQUIRE(u.bound, "UNet weights are not bound");
  const UNetCfg& cfg = u.cfg;
  const int temb = cfg.boc[0] * 4;
  std::vector<f16> hs;
  host_sinusoid(timesteps, nsteps, cfg.boc[0], hs);
  f16* sin_t = upload_f16(c, hs);
  f16* e1 = c.ws.get<f16>((long)nsteps * temb);
  f16* emb = c.ws.get<f16>((long)nsteps * temb);
  { Epi e; e.act = UG_ACT_SILU; linear(c, sin_t, nsteps, u.te1, e1, e); }
  linear(c, e1, nsteps, u.te2, emb);
  const float ids[3] = {7.f, 127.f, 0.02f};   // fps-1... as passed by DepthCrafter (7), motion bucket 127, noise_aug 0.02
  UG_REQUIRE(cfg.add_dim * 3 == cfg.proj_in_dim, "added_time_ids width");
  host_sinusoid(ids, 3, cfg.add_dim, hs);
  f16* sin_a = upload_f16(c, hs);             // [3][add_dim] row-major == reshape(1, 3*add_dim)
  f16* a1 = c.ws.get<f16>(temb); f16* aug = c.ws.get<f16>(temb);
  { Epi e; e.act = UG_ACT_SILU; linear(c, sin_a, 1, u.ae1, a1, e); }
  linear(c, a1, 1, u.ae2, aug);
  launch_add_rowvec(emb, aug, emb, nsteps, temb, nsteps, c.stream);   // emb = emb + aug_emb (fp16)
  f16* se = c.ws.get<f16>((long)nsteps * temb);
  launch_silu_f16(emb, se, (long)nsteps * temb, c.stream);   // F.silu on the half tensor
  long total = 0;
  u.tproj_off_s.clear(); u.tproj_off_t.clear();
  for (auto* r : u.temb_s) { u.tproj_off_s.push_back(total); total += (long)nsteps * r->temb.out; }
  for (auto* r : u.temb_t) { u.tproj_off_t.push_back(total); total += (long)nsteps * r->temb.out; }
  u.tproj = c.ws.get<f16>(total); u.tproj_steps = nsteps;
  for (size_t i = 0; i < u.temb_s.size(); ++i) linear(c, se, nsteps, u.temb_s[i]->temb, u.tproj + u.tproj_off_s[i]);
  for (size_t i = 0; i < u.temb_t.size(); ++i) linear(c, se, nsteps, u.temb_t[i]->temb, u.tproj + u.tproj_off_t[i]);
  // transformers
  std::vector<float> fidx(T);
  for (int i = 0; i < T; ++i) fidx[i] = (float)i;
  auto prep = [&](Transformer& tr) {
    host_sinusoid(fidx.data(), T, tr.C, hs);
    f16* s = upload_f16(c, hs);
    f16* m1 = c.ws.get<f16>((long)T * 4 * tr.C);
    tr.frame_emb = c.ws.get<f16>((long)T * tr.C); tr.frame_emb_T = T;
    { Epi e; e.act = UG_ACT_SILU; linear(c, s, T, tr.tpe1, m1, e); }
    linear(c, m1, T, tr.tpe2, tr.frame_emb);
    f16* v = c.ws.get<f16>((long)T * tr.C);
    tr.cross_sp = c.ws.get<f16>((long)T * tr.C);
    linear(c, clip_emb, T, tr.v2, v);
    linear(c, v, T, tr.o2, tr.cross_sp);
    f16* v1 = c.ws.get<f16>(tr.C);
    tr.cross_tm = c.ws.get<f16>(tr.C);
    linear(c, clip_emb, 1, tr.tv2, v1);        // first frame's token
    linear(c, v1, 1, tr.to2, tr.cross_tm);
  };
  for (auto& d : u.down) for (auto& t : d.attn) prep(t);
  prep(u.mid_attn);
  for (auto& d : u.up) for (auto& t : d.attn) prep(t);
}

f16* unet_forward(Ctx& c, const f16* x, int T, int h, int w, int step) {
  UNet& u = c.unet;
  const UNetCfg& cfg = u.cfg;
  UG_REQUIRE(u.bound && u.tproj && step < u.tproj_steps, "unet_prepare must run first");
  UG_REQUIRE(h % 8 == 0 && w % 8 == 0, "latent height/width must be multiples of 8 (image multiples of 64)");
  const int G = cfg.groups, n = cfg.nlev;
  auto tp_s = [&](const STRes& r) { return r.s.has_temb ? u.tproj + u.tproj_off_s[r.s.tidx] + (long)step * r.s.temb.out : nullptr; };
  auto tp_t = [&](const STRes& r) { return r.t.has_temb ? u.tproj + u.tproj_off_t[r.t.tidx] + (long)step * r.t.temb.out : nullptr; };
  f16* out = c.ws.get<f16>((long)T * h * w * cfg.out_ch);
  const size_t mk = c.ws.mark();
  struct Skip { f16* p; int C; };
  std::vector<Skip> skips;
  int ch = cfg.boc[0], ch_h = h, ch_w = w;
  f16* cur = c.ws.get<f16>((long)T * h * w * ch);
  conv(c, x, cfg.in_ch, nullptr, 0, T, h, w, u.conv_in, 1, 1, 1, 1, cur);
  StatPart cs;      // epilogue statistics of `cur` (valid while its producer was a convolution that could write them)
  skips.push_back({cur, ch});
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < cfg.layers; ++j) {
      const STRes& r = u.down[i].res[j];
      StatPart so;
      cur = stres_forward(c, r, cur, ch, nullptr, 0, T, ch_h, ch_w, G, tp_s(r), tp_t(r), &so, cs.part ? &cs : nullptr);
      ch = r.cout; cs = so;
      if (cfg.has_attn[i]) { cur = transformer_forward(c, u.down[i].attn[j], cur, T, ch_h, ch_w, G, &cs); cs = StatPart(); }
      skips.push_back({cur, ch});
    }
    if (u.down[i].has_down) {
      f16* d = c.ws.get<f16>((long)T * (ch_h / 2) * (ch_w / 2) * ch);
      conv(c, cur, ch, nullptr, 0, T, ch_h, ch_w, u.down[i].down, 2, 1, 1, 1, d);
      ch_h /= 2; ch_w /= 2; cur = d; cs = StatPart();
      skips.push_back({cur, ch});
    }
  }
  { StatPart so;
    cur = stres_forward(c, u.mid0, cur, ch, nullptr, 0, T, ch_h, ch_w, G, tp_s(u.mid0), tp_t(u.mid0), &so, cs.part ? &cs : nullptr);
    cur = transformer_forward(c, u.mid_attn, cur, T, ch_h, ch_w, G, &so); }
  cur = stres_forward(c, u.mid1, cur, ch, nullptr, 0, T, ch_h, ch_w, G, tp_s(u.mid1), tp_t(u.mid1));
  for (int i = 0; i < n; ++i) {
    const int lev = n - 1 - i;
    for (size_t j = 0; j < u.up[i].res.size(); ++j) {
      const STRes& r = u.up[i].res[j];
      const Skip sk = skips.back(); skips.pop_back();
      StatPart so;
      cur = stres_forward(c, r, cur, ch, sk.p, sk.C, T, ch_h, ch_w, G, tp_s(r), tp_t(r), &so);
      ch = r.cout;
      if (cfg.has_attn[lev]) cur = transformer_forward(c, u.up[i].attn[j], cur, T, ch_h, ch_w, G, &so);
    }
    if (u.up[i].has_up) {
      f16* d = c.ws.get<f16>((long)T * (ch_h * 2) * (ch_w * 2) * ch);
      conv(c, cur, ch, nullptr, 0, T, ch_h, ch_w, u.up[i].up, 1, 1, 1, 2, d);
      ch_h *= 2; ch_w *= 2; cur = d;
    }
  }
  f16* a = c.ws.get<f16>((long)T * h * w * ch);
  groupnorm(c, cur, ch, nullptr, 0, T, h * w, G, u.norm_out, 0, 1, a);
  conv(c, a, ch, nullptr, 0, T, h, w, u.conv_out, 1, 1, 1, 1, out);
  c.ws.release(mk);
  return out;
}

// ----------------------------------------------------------------------------- VAE
static f16* vattn_forward(Ctx& c, const VAttn& at, const f16* x, int T, int hw, int G) {
  const int C = at.C; const long M = (long)T * hw;
  f16* out = c.ws.get<f16>(M * C);
  const size_t mk = c.ws.mark();
  f16* xn = c.ws.get<f16>(M * C);
  groupnorm(c, x, C, nullptr, 0, T, hw, G, at.gn, 0, 0, xn);
  f16* qkv = c.ws.get<f16>(M * 3 * C);
  linear(c, xn, M, at.qkv, qkv);
  f16* ao = xn;
  unfused_attention(c, qkv, 3 * C, T, hw, 1, C, ao, C);
  { Epi e; e.R1 = x; linear(c, ao, M, at.out, out, e); }
  c.ws.release(mk);
  return out;
}

// ---- float32-grade encoder: fp32 residual stream / norms / softmax, GEMMs on fp16 hi/lo pairs (kernels/wide.hip) ----
static void gn32(Ctx& c, const float* x, int T, int HW, int G, const Norm& n, int silu, f16* ypair) {
  const size_t mk = c.ws.mark();
  void* ws = c.ws.alloc(gn32_ws_bytes(T, HW, n.c, G));
  {
    ProfScope ps(c, "groupnorm_f32", 0, (double)T * HW * n.c * 12.0);
    launch_gn32_pair(x, ypair, T, HW, n.c, G, n.eps, silu, n.g, n.b, ws, c.stream);
  }
  c.ws.release(mk);
}
static void split_pair(Ctx& c, const float* x, f16* y, long M, int C) {
  ProfScope ps(c, "split_pair", 0, (double)M * C * 8.0);
  launch_split_pair(x, y, M, C, c.stream);
}
// conv over a pair tensor [M, 2C] with K-doubled weights -> fp32 [M, cout]
// res (optional): float32 residual [M, cout] added in the GEMM epilogue (may alias `out`: every element is read and written by the same lane)
static void conv_w(Ctx& c, const f16* xpair, int T, int Hi, int Wi, const Conv& cv, int stride, int pad_t, int pad_l, float* out, const float* res = nullptr) {
  Epi e; e.flags = UG_F_OUT_F32; e.alg = 0.5f;
  if (res) { e.R1 = (const f16*)res; e.ldr1 = cv.cout; e.c1 = 1.f; e.flags |= UG_F_R1_F32; }
  conv(c, xpair, cv.cinp, nullptr, 0, T, Hi, Wi, cv, stride, pad_t, pad_l, 1, (f16*)out, e);
}
static void res2d_wide(Ctx& c, const Res2D& r, const float* x, int cin, int T, int h, int w, int G, float* out) {
  const long M = (long)T * h * w;
  const int cout = r.c1.cout;
  const size_t mk = c.ws.mark();
  f16* a = c.ws.get<f16>(M * 2 * std::max(cin, cout));
  gn32(c, x, T, h * w, G, r.n1, 1, a);
  float* hb = c.ws.get<float>(M * cout);
  conv_w(c, a, T, h, w, r.c1, 1, 1, 1, hb);
  const float* res = x;
  if (r.has_sc) {                                  // 1x1 shortcut of the channel-changing blocks -> out, which then is the residual of conv2
    split_pair(c, x, a, M, cin);                   // `a` is dead after conv1 (stream order)
    conv_w(c, a, T, h, w, r.sc, 1, 0, 0, out);
    res = out;
  }
  gn32(c, hb, T, h * w, G, r.n2, 1, a);
  // round 3: the float32 residual is added in conv2's epilogue (UG_F_R1_F32) - same fp32 sum as the former add_f32 pass, one tensor write + read less
  conv_w(c, a, T, h, w, r.c2, 1, 1, 1, out, res);
  c.ws.release(mk);
}
static void vattn_wide(Ctx& c, const VAttn& at, const float* x, int T, int hw, int G, float* out) {
  const int C = at.C, S = hw, Spad = pad8(S); const long M = (long)T * hw;
  const size_t mk = c.ws.mark();
  f16* xn = c.ws.get<f16>(M * 2 * C);
  gn32(c, x, T, hw, G, at.gn, 0, xn);
  float* qkv = c.ws.get<float>(M * 3 * C);
  { Epi e; e.flags = UG_F_OUT_F32; e.alg = 0.5f; linear(c, xn, M, at.qkv, (f16*)qkv, e); }
  f16* Aq = c.ws.get<f16>(M * 3 * C); f16* Bk = c.ws.get<f16>(M * 3 * C);
  launch_qk_terms(qkv, Aq, Bk, M, C, c.stream);
  f16* Vt = c.ws.get<f16>((long)T * C * 3 * Spad);
  launch_vt_terms(qkv, Vt, T, S, Spad, C, c.stream);
  float* sc = c.ws.get<float>((long)T * S * Spad);
  GemmP p; memset(&p, 0, sizeof(p));
  p.A0 = Aq; p.C0 = 3 * C; p.M = S; p.N = S; p.K = 3 * C; p.W = Bk; p.ldw = 3 * C;
  p.c0 = 1.0f / sqrtf((float)C); p.Out = sc; p.ldo = Spad; p.flags = UG_F_OUT_F32;
  p.nb_inner = 1; p.sA_o = (long)S * 3 * C; p.sW_o = (long)S * 3 * C; p.sO_o = (long)S * Spad;
  run_gemm(c, p, T, "gemm_attn_qk", 1.f / 3.f);
  f16* P = c.ws.get<f16>((long)T * S * 3 * Spad);
  {
    ProfScope ps(c, "softmax_rows", 0, (double)T * S * Spad * 10.0);
    launch_softmax_pair(sc, Spad, P, Spad, (long)T * S, S, c.stream);
  }
  float* ao = qkv;   // qkv is dead (Aq / Bk / Vt hold its terms)
  GemmP q; memset(&q, 0, sizeof(q));
  q.A0 = P; q.C0 = 3 * Spad; q.M = S; q.N = C; q.K = 3 * Spad; q.W = Vt; q.ldw = 3 * Spad; q.c0 = 1.f;
  q.Out = ao; q.ldo = C; q.flags = UG_F_OUT_F32; q.nb_inner = 1;
  q.sA_o = (long)S * 3 * Spad; q.sW_o = (long)C * 3 * Spad; q.sO_o = (long)S * C;
  run_gemm(c, q, T, "gemm_attn_pv", 1.f / 3.f);
  split_pair(c, ao, xn, M, C);
  { Epi e; e.flags = UG_F_OUT_F32 | UG_F_R1_F32; e.alg = 0.5f; e.R1 = (const f16*)x; e.ldr1 = C; e.c1 = 1.f; linear(c, xn, M, at.out, (f16*)out, e); }   // + float32 residual in the epilogue
  c.ws.release(mk);
}

static f16* vae_encode_wide(Ctx& c, VAE& v, const f16* x8, int T, int H, int W) {
  UG_REQUIRE(v.wide_bound, "float32-grade VAE encoder weights are not bound");
  const VAECfg& cfg = v.cfg; const int G = cfg.groups, n = cfg.nlev;
  int hh = H, ww = W;
  for (int i = 0; i < n - 1; ++i) { hh /= 2; ww /= 2; }
  f16* lat = c.ws.get<f16>((long)T * hh * ww * cfg.lat);
  const size_t mk = c.ws.mark();
  int ch = cfg.boc[0], ch_h = H, ch_w = W;
  float* cur = c.ws.get<float>((long)T * H * W * ch);
  { Epi e; e.flags = UG_F_OUT_F32; conv(c, x8, 8, nullptr, 0, T, H, W, v.e_in, 1, 1, 1, 1, (f16*)cur, e); }   // fp16 image: exact as is
  for (int i = 0; i < n; ++i) {
    for (auto& r : v.edown_w[i].res) {
      float* o = c.ws.get<float>((long)T * ch_h * ch_w * r.c1.cout);
      res2d_wide(c, r, cur, ch, T, ch_h, ch_w, G, o);
      cur = o; ch = r.c1.cout;
    }
    if (v.edown_w[i].has_down) {   // F.pad(0,1,0,1) + stride-2 conv without padding
      const long Mi = (long)T * ch_h * ch_w;
      float* d = c.ws.get<float>(Mi / 4 * ch);
      const size_t m2 = c.ws.mark();
      f16* xp = c.ws.get<f16>(Mi * 2 * ch);
      split_pair(c, cur, xp, Mi, ch);
      conv_w(c, xp, T, ch_h, ch_w, v.edown_w[i].down, 2, 0, 0, d);
      c.ws.release(m2);
      ch_h /= 2; ch_w /= 2; cur = d;
    }
  }
  const long M = (long)T * ch_h * ch_w;
  float* o = c.ws.get<float>(M * ch);
  res2d_wide(c, v.emid0_w, cur, ch, T, ch_h, ch_w, G, o); cur = o;
  o = c.ws.get<float>(M * ch);
  vattn_wide(c, v.eattn_w, cur, T, ch_h * ch_w, G, o); cur = o;
  o = c.ws.get<float>(M * ch);
  res2d_wide(c, v.emid1_w, cur, ch, T, ch_h, ch_w, G, o); cur = o;
  f16* a = c.ws.get<f16>(M * 2 * ch);
  gn32(c, cur, T, ch_h * ch_w, G, v.e_norm, 1, a);
  float* mom = c.ws.get<float>(M * 2 * cfg.lat);
  conv_w(c, a, T, ch_h, ch_w, v.e_out_w, 1, 1, 1, mom);
  f16* mp = c.ws.get<f16>(M * 4 * cfg.lat);
  split_pair(c, mom, mp, M, 2 * cfg.lat);
  float* q = c.ws.get<float>(M * 2 * cfg.lat);
  conv_w(c, mp, T, ch_h, ch_w, v.quant_w, 1, 0, 0, q);
  launch_take_cols_f16(q, 2 * cfg.lat, lat, M, cfg.lat, c.stream);   // latent_dist.mode() = mean half, cast to the pipeline's fp16
  c.ws.release(mk);
  return lat;
}

f16* vae_encode(Ctx& c, const f16* x8, int T, int H, int W) { return vae_encode_v(c, c.vae, x8, T, H, W, c.vae_encode_fp32 != 0); }
f16* vae_encode_v(Ctx& c, VAE& v, const f16* x8, int T, int H, int W, bool fp32_grade) {
  if (fp32_grade) return vae_encode_wide(c, v, x8, T, H, W);
  UG_REQUIRE(v.bound, "VAE weights are not bound");
  const VAECfg& cfg = v.cfg; const int G = cfg.groups;
  UG_REQUIRE(v.e_in.cinp == 8, "encoder input is staged as 8 channels");
  const int n = cfg.nlev;
  int hh = H, ww = W;
  for (int i = 0; i < n - 1; ++i) { hh /= 2; ww /= 2; }
  f16* lat = c.ws.get<f16>((long)T * hh * ww * cfg.lat);
  const size_t mk = c.ws.mark();
  int ch = cfg.boc[0], ch_h = H, ch_w = W;
  f16* cur = c.ws.get<f16>((long)T * H * W * ch);
  conv(c, x8, 8, nullptr, 0, T, H, W, v.e_in, 1, 1, 1, 1, cur);
  for (int i = 0; i < n; ++i) {
    for (auto& r : v.edown[i].res) {
      f16* o = c.ws.get<f16>((long)T * ch_h * ch_w * r.c1.cout);
      res2d_forward(c, r, cur, ch, nullptr, 0, T, ch_h, ch_w, G, nullptr, o);
      cur = o; ch = r.c1.cout;
    }
    if (v.edown[i].has_down) {   // F.pad(0,1,0,1) + stride-2 conv without padding
      f16* d = c.ws.get<f16>((long)T * (ch_h / 2) * (ch_w / 2) * ch);
      conv(c, cur, ch, nullptr, 0, T, ch_h, ch_w, v.edown[i].down, 2, 0, 0, 1, d);
      ch_h /= 2; ch_w /= 2; cur = d;
    }
  }
  const long M = (long)T * ch_h * ch_w;
  f16* o = c.ws.get<f16>(M * ch);
  res2d_forward(c, v.emid0, cur, ch, nullptr, 0, T, ch_h, ch_w, G, nullptr, o); cur = o;
  cur = vattn_forward(c, v.eattn, cur, T, ch_h * ch_w, G);
  o = c.ws.get<f16>(M * ch);
  res2d_forward(c, v.emid1, cur, ch, nullptr, 0, T, ch_h, ch_w, G, nullptr, o); cur = o;
  f16* a = c.ws.get<f16>(M * ch);
  groupnorm(c, cur, ch, nullptr, 0, T, ch_h * ch_w, G, v.e_norm, 0, 1, a);
  f16* mom = c.ws.get<f16>(M * 2 * cfg.lat);
  conv(c, a, ch, nullptr, 0, T, ch_h, ch_w, v.e_out, 1, 1, 1, 1, mom);
  f16* q = c.ws.get<f16>(M * 2 * cfg.lat);
  conv(c, mom, 2 * cfg.lat, nullptr, 0, T, ch_h, ch_w, v.quant, 1, 0, 0, 1, q);
  launch_copy2d(q, 2 * cfg.lat, lat, cfg.lat, M, cfg.lat, c.stream);   // latent_dist.mode() = mean half
  c.ws.release(mk);
  return lat;
}

void vae_decode(Ctx& c, const f16* z, int T, int h, int w, float* frames_out) {
  VAE& v = c.vae;
  UG_REQUIRE(v.bound && !v.dec2d, "temporal VAE decoder weights are not bound");
  const VAECfg& cfg = v.cfg; const int G = cfg.groups, n = cfg.nlev;
  UG_REQUIRE(v.d_in.cinp == 8, "decoder input is staged as 8 channels");
  const size_t mk = c.ws.mark();
  const long M0 = (long)T * h * w;
  f16* z8 = c.ws.get<f16>(M0 * 8);
  launch_pad_channels(z, cfg.lat, z8, 8, M0, c.stream);
  int ch = cfg.boc[n - 1], ch_h = h, ch_w = w;
  f16* cur = c.ws.get<f16>(M0 * ch);
  conv(c, z8, 8, nullptr, 0, T, h, w, v.d_in, 1, 1, 1, 1, cur);
  cur = stres_forward(c, v.dmid[0], cur, ch, nullptr, 0, T, ch_h, ch_w, G, nullptr, nullptr);
  StatPart cs;      // epilogue statistics of `cur` (res-block -> res-block chains: the next block's first GroupNorm needs no statistics pass)
  for (size_t j = 1; j < v.dmid.size(); ++j) {
    if (j == 1) cur = vattn_forward(c, v.dattn, cur, T, ch_h * ch_w, G);
    StatPart so;
    cur = stres_forward(c, v.dmid[j], cur, ch, nullptr, 0, T, ch_h, ch_w, G, nullptr, nullptr, &so, cs.part ? &cs : nullptr);
    cs = so;
  }
  for (int i = 0; i < n; ++i) {
    for (auto& r : v.dup[i].res) {
      StatPart so;
      cur = stres_forward(c, r, cur, ch, nullptr, 0, T, ch_h, ch_w, G, nullptr, nullptr, &so, cs.part ? &cs : nullptr);
      ch = r.cout; cs = so;
    }
    if (v.dup[i].has_up) {
      f16* d = c.ws.get<f16>((long)T * (ch_h * 2) * (ch_w * 2) * ch);
      conv(c, cur, ch, nullptr, 0, T, ch_h, ch_w, v.dup[i].up, 1, 1, 1, 2, d);
      ch_h *= 2; ch_w *= 2; cur = d; cs = StatPart();
    }
  }
  const long M = (long)T * ch_h * ch_w;
  f16* a = c.ws.get<f16>(M * ch);
  groupnorm(c, cur, ch, nullptr, 0, T, ch_h * ch_w, G, v.d_norm, 0, 1, a, cs.part ? &cs : nullptr);
  f16* rgb = c.ws.get<f16>(M * 8);
  conv(c, a, ch, nullptr, 0, T, ch_h, ch_w, v.d_out, 1, 1, 1, 1, rgb, Epi(), 8);
  {
    ProfScope ps(c, "time_conv_out", 0, (double)M * (16.0 + 12.0));
    launch_time_conv_out(rgb, v.tco_w, v.tco_b, frames_out, T, (long)ch_h * ch_w, 8, c.stream);
  }
  c.ws.release(mk);
}

// ----------------------------------------------------------------------------- CLIP
f16* clip_embed(Ctx& c, const f16* video_m11, int T, int H, int W) {
  CLIP& m = c.clip;
  UG_REQUIRE(m.bound, "CLIP weights are not bound");
  const CLIPCfg& cfg = m.cfg;
  const int d = cfg.hidden, P = cfg.patch, g = cfg.image / P, np = g * g, S = np + 1;
  f16* emb = c.ws.get<f16>((long)T * cfg.proj);
  const size_t mk = c.ws.mark();
  f16* patches = c.ws.get<f16>((long)T * np * m.patch_k);
  UG_CHECK(hipMemsetAsync(patches, 0, (size_t)T * np * m.patch_k * 2, c.stream));
  {
    ProfScope ps(c, "clip_preprocess", 0, 0);
    launch_clip_patchify(video_m11, patches, T, H, W, cfg.image, P, m.patch_k, c.stream);
  }
  f16* pe = c.ws.get<f16>((long)T * np * d);
  { Lin l; l.w = m.patch_w; l.b = nullptr; l.in = m.patch_k; l.out = d; linear(c, patches, (long)T * np, l, pe); }
  const long M = (long)T * S;
  f16* x = c.ws.get<f16>(M * d);
  launch_clip_assemble(pe, m.cls, m.pos, x, T, np, d, c.stream);
  f16* y = c.ws.get<f16>(M * d);
  layernorm(c, x, M, m.pre, y);
  f16* t1 = x;
  f16* qkv = c.ws.get<f16>(M * 3 * d);
  f16* ao = c.ws.get<f16>(M * d);
  f16* mid = c.ws.get<f16>(M * cfg.inter);
  f16* y2 = c.ws.get<f16>(M * d);
  for (auto& l : m.layers) {
    layernorm(c, y, M, l.ln1, t1);
    linear(c, t1, M, l.qkv, qkv);
    const int dh = d / cfg.heads;
    if (dh != 64 && flash_attn_dh_supported(dh) && !getenv("UG_CLIP_UNFUSED")) {
      // round 3: ViT-H/14 has 16 heads of 80 - one fused kernel instead of Q K^T batch + row softmax + V transpose + P V batch (~160 us per layer)
      FlashP fp; fp.Q = qkv; fp.K = qkv + d; fp.V = qkv + 2 * d; fp.ldq = fp.ldk = fp.ldv = 3 * d; fp.O = ao; fp.ldo = d;
      fp.B = T; fp.H = cfg.heads; fp.S = S; fp.scale = 1.0f / sqrtf((float)dh);
      ProfScope ps(c, "flash_attn_clip", 4.0 * T * cfg.heads * (double)S * S * dh, 0);
      launch_flash_attn_dh(fp, dh, c.stream);
    } else unfused_attention(c, qkv, 3 * d, T, S, cfg.heads, dh, ao, d);
    { Epi e; e.R1 = y; linear(c, ao, M, l.out, y2, e); }
    layernorm(c, y2, M, l.ln2, t1);
    { Epi e; e.act = UG_ACT_GELU; linear(c, t1, M, l.fc1, mid, e); }
    { Epi e; e.R1 = y2; linear(c, mid, M, l.fc2, y, e); }
  }
  f16* pooled = c.ws.get<f16>((long)T * d);
  launch_copy2d(y, (long)S * d, pooled, d, T, d, c.stream);      // CLS token of every frame
  f16* pn = c.ws.get<f16>((long)T * d);
  layernorm(c, pooled, T, m.post, pn);
  linear(c, pn, T, m.proj, emb);
  c.ws.release(mk);
  return emb;
}

// ----------------------------------------------------------------------------- pipeline
void dc_set_inputs(Ctx& c, const float* frames, int T, int H, int W, const float* noise_lat, const float* noise_aug,
                   const float* K33) {
  UG_REQUIRE(H % 64 == 0 && W % 64 == 0, "height and width must be multiples of 64 (VAE /8, UNet /8)");
  UG_REQUIRE(T >= 1 && T <= 4096, "1..4096 frames (at most 128 per denoising window)");
  if (c.io_ready) { c.ws.release(c.io_mark); c.io_ready = false; }
  c.io_mark = c.ws.mark();
  c.T = T; c.H = H; c.W = W;
  const long px = (long)T * H * W, lp = (long)T * (H / 8) * (W / 8);
  c.d_frames = c.ws.get<float>(px * 3); c.d_noise_aug = c.ws.get<float>(px * 3);
  c.d_noise_lat = c.ws.get<float>(lp * 4); c.d_K = c.ws.get<float>((long)T * 9);
  c.d_out_frames = c.ws.get<float>(px * 3); c.d_depth = c.ws.get<float>(px); c.d_normals = c.ws.get<float>(px * 3);
  c.d_mm = c.ws.get<float>(64);
  UG_CHECK(hipMemcpyAsync(c.d_frames, frames, px * 3 * 4, hipMemcpyHostToDevice, c.stream));
  UG_CHECK(hipMemcpyAsync(c.d_noise_aug, noise_aug, px * 3 * 4, hipMemcpyHostToDevice, c.stream));
  UG_CHECK(hipMemcpyAsync(c.d_noise_lat, noise_lat, lp * 4 * 4, hipMemcpyHostToDevice, c.stream));
  if (K33) UG_CHECK(hipMemcpyAsync(c.d_K, K33, (size_t)T * 9 * 4, hipMemcpyHostToDevice, c.stream));
  else UG_CHECK(hipMemsetAsync(c.d_K, 0, (size_t)T * 9 * 4, c.stream));
  UG_CHECK(hipStreamSynchronize(c.stream));
  c.io_ready = true;
}

static void karras_sigmas(int n, std::vector<float>& sig, std::vector<float>& ts) {
  const double smin = 0.002, smax = 700.0, rho = 7.0;
  sig.resize(n + 1); ts.resize(n);
  for (int i = 0; i < n; ++i) {
    const double ramp = (n == 1) ? 0.0 : (double)i / (double)(n - 1);
    const double lo = pow(smin, 1.0 / rho), hi = pow(smax, 1.0 / rho);
    sig[i] = (float)pow(hi + ramp * (lo - hi), rho);
    ts[i] = 0.25f * logf(sig[i]);
  }
  sig[n] = 0.f;
}

// window == 0 (or >= T): the reference path, one denoising pass over the whole clip (model/depthcrafter.py:87-88 passes
// window_size = len(frames)).  0 < window < T: upstream DepthCrafter's long-video mode, restated from the
// published pipeline - UNPINNED, the reference never takes it: windows of `window` frames advance by window - overlap; a
// window after the first starts its first `overlap` frames from the previous window's result re-noised to sigma_0
// (latents_all[-overlap:] + noise * sigma_0), the unit noise of the window is the previous one rotated by `overlap` frames, and the
// overlap is cross-faded linearly into the running result.  CLIP / VAE conditioning is computed once for all frames.
void dc_run(Ctx& c, int steps, int chunk, int with_normals, int window, int overlap) {
  UG_REQUIRE(c.io_ready, "ug_dc_set_inputs must be called first");
  UG_REQUIRE(steps >= 1 && chunk >= 1, "steps/chunk");
  const int T = c.T, H = c.H, W = c.W, h = H / 8, w = W / 8;
  const bool windows = window > 0 && window < T;
  if (windows) UG_REQUIRE(window <= 128 && overlap >= 0 && overlap < window, "window must be <= 128 frames and overlap < window");
  else UG_REQUIRE(T <= 128, "more than 128 frames need latent sliding windows (window <= 128)");
  const long px = (long)T * H * W, lp = (long)T * h * w;
  RunGuard guard(c);
  // 1. inputs -> fp16, [-1,1], noise augmentation
  f16* clip_src = c.ws.get<f16>(px * 3);
  f16* vae_in = c.ws.get<f16>(px * 8);
  launch_prep_video(c.d_frames, c.d_noise_aug, clip_src, vae_in, T, H, W, 0.02f, c.stream);
  // 2. CLIP image embeddings (per frame) and 3. VAE encode -> conditioning latents (mode of the posterior, unscaled), `chunk` frames at a
  // time as the reference does.  The CLIP tower and the encoder chunks are independent of each other: run_lanes spreads them over streams.
  f16* emb = c.ws.get<f16>((long)T * c.clip.cfg.proj);
  f16* cond = c.ws.get<f16>(lp * 4);
  {
    const int nchunks = (T + chunk - 1) / chunk;
    auto kind = [&](int i) {
      char b[96];
      if (i == nchunks) snprintf(b, sizeof(b), "clip:%dx%dx%d", T, H, W);
      else snprintf(b, sizeof(b), "enc%d:%dx%dx%d", c.vae_encode_fp32, std::min(chunk, T - i * chunk), H, W);
      return std::string(b);
    };
    run_lanes(c, nchunks + 1, kind, [&](int i) {
      const size_t m2 = c.ws.mark();
      if (i == nchunks) {
        f16* e = clip_embed(c, clip_src, T, H, W);
        UG_CHECK(hipMemcpyAsync(emb, e, (size_t)T * c.clip.cfg.proj * 2, hipMemcpyDeviceToDevice, c.stream));
      } else {
        const int t0 = i * chunk, tc = std::min(chunk, T - t0);
        f16* l = vae_encode(c, vae_in + (long)t0 * H * W * 8, tc, H, W);
        UG_CHECK(hipMemcpyAsync(cond + (long)t0 * h * w * 4, l, (size_t)tc * h * w * 4 * 2, hipMemcpyDeviceToDevice, c.stream));
      }
      c.ws.release(m2);
    });
  }
  // 4. scheduler tables + latents
  std::vector<float> sig, ts;
  karras_sigmas(steps, sig, ts);
  const float sigma0 = sqrtf(sig[0] * sig[0] + 1.f);    // init_noise_sigma, "leading" spacing
  f16* lat = c.ws.get<f16>(lp * 4);      // running result (all frames)
  if (!windows) {
    launch_init_latents2(c.d_noise_lat, lat, sigma0, T, (long)h * w, c.stream);
    unet_prepare(c, T, emb, ts.data(), steps);
    f16* xin = c.ws.get<f16>(lp * 8);
    // 5. denoise loop: no host sync inside
    const bool host_timing = getenv("UG_HOST_TIMING") != nullptr;   // measurement aid: is the host ahead of the stream?
    if (host_timing) UG_CHECK(hipStreamSynchronize(c.stream));
    const auto ht0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; ++i) {
      const size_t m2 = c.ws.mark();
      launch_make_unet_input(lat, cond, xin, lp, sqrtf(sig[i] * sig[i] + 1.f), c.stream);
      f16* v = unet_forward(c, xin, T, h, w, i);
      launch_euler_step(v, lat, lp * 4, sig[i], sig[i + 1], c.stream);
      if (c.trace_host && i < c.trace_steps) {   // parity instrumentation only (tests/): latents [T,h,w,4] after step i
        float* tf = c.ws.get<float>(lp * 4);
        launch_cast_f16_f32(lat, tf, lp * 4, c.stream);
        UG_CHECK(hipMemcpyAsync(c.trace_host + (size_t)i * lp * 4, tf, (size_t)lp * 4 * 4, hipMemcpyDeviceToHost, c.stream));
        UG_CHECK(hipStreamSynchronize(c.stream));
      }
      c.ws.release(m2);
    }
    if (host_timing) {
      const auto ht1 = std::chrono::steady_clock::now();
      UG_CHECK(hipStreamSynchronize(c.stream));
      const auto ht2 = std::chrono::steady_clock::now();
      fprintf(stderr, "[ug] denoise loop: host enqueue %.1f ms, stream done after %.1f ms\n",
              std::chrono::duration<double, std::milli>(ht1 - ht0).count(), std::chrono::duration<double, std::milli>(ht2 - ht0).count());
    }
  } else {
    const long fe = (long)h * w * 4;                     // latent elements per frame
    const int stride = window - overlap;
    f16* init = c.ws.get<f16>(fe * window);              // latents_init: unit noise * sigma0 for one window
    f16* rot = c.ws.get<f16>(fe * window);
    f16* cur = c.ws.get<f16>(fe * window);
    f16* xin = c.ws.get<f16>(fe * window * 2);
    launch_init_latents2(c.d_noise_lat, init, sigma0, window, (long)h * w, c.stream);
    int n_all = 0;
    for (int s0 = 0; s0 < T - overlap; s0 += stride) {
      const int Tw = std::min(window, T - s0);
      const long lw = fe * Tw;
      UG_CHECK(hipMemcpyAsync(cur, init, (size_t)lw * 2, hipMemcpyDeviceToDevice, c.stream));
      // latents_init <- cat(latents_init[-overlap:], latents_init[:stride])
      if (overlap) UG_CHECK(hipMemcpyAsync(rot, init + fe * (window - overlap), (size_t)fe * overlap * 2, hipMemcpyDeviceToDevice, c.stream));
      UG_CHECK(hipMemcpyAsync(rot + fe * overlap, init, (size_t)fe * stride * 2, hipMemcpyDeviceToDevice, c.stream));
      std::swap(init, rot);
      if (n_all > 0 && overlap)   // first `overlap` frames: previous result + noise at sigma_0 (cur holds noise * sigma0)
        launch_axpby_f16(lat + fe * (n_all - overlap), 1.f, cur, sig[0] / sigma0, cur, fe * overlap, c.stream);
      const size_t mw = c.ws.mark();
      unet_prepare(c, Tw, emb + (long)s0 * c.clip.cfg.proj, ts.data(), steps);
      for (int i = 0; i < steps; ++i) {
        const size_t m2 = c.ws.mark();
        launch_make_unet_input(cur, cond + fe * s0, xin, (long)Tw * h * w, sqrtf(sig[i] * sig[i] + 1.f), c.stream);
        f16* v = unet_forward(c, xin, Tw, h, w, i);
        launch_euler_step(v, cur, lw, sig[i], sig[i + 1], c.stream);
        c.ws.release(m2);
      }
      c.ws.release(mw);
      if (n_all == 0) {
        UG_CHECK(hipMemcpyAsync(lat, cur, (size_t)lw * 2, hipMemcpyDeviceToDevice, c.stream));
        n_all = Tw;
      } else {
        if (overlap) launch_crossfade_f16(cur, lat + fe * (n_all - overlap), fe, overlap, c.stream);
        UG_CHECK(hipMemcpyAsync(lat + fe * n_all, cur + fe * overlap, (size_t)fe * (Tw - overlap) * 2, hipMemcpyDeviceToDevice, c.stream));
        n_all += Tw - overlap;
      }
    }
    UG_REQUIRE(n_all == T, "window bookkeeping");
  }
  // 6. decode in chunks of `chunk` frames (temporal layers only see the chunk, as in the reference)
  f16* z = c.ws.get<f16>(lp * 4);
  launch_scale_f16(lat, z, 1.0f / c.vae.cfg.scaling, lp * 4, c.stream);
  {
    const int nchunks = (T + chunk - 1) / chunk;
    auto kind = [&](int i) { char b[96]; snprintf(b, sizeof(b), "dec:%dx%dx%d", std::min(chunk, T - i * chunk), h, w); return std::string(b); };
    run_lanes(c, nchunks, kind, [&](int i) {
      const int t0 = i * chunk, tc = std::min(chunk, T - t0);
      vae_decode(c, z + (long)t0 * h * w * 4, tc, h, w, c.d_out_frames + (long)t0 * H * W * 3);
    });
  }
  // 7. wrapper post-processing on device
  launch_depth_post(c.d_out_frames, c.d_depth, c.d_mm, px, c.stream);
  if (with_normals) {
    ProfScope ps(c, "normals", 0, (double)px * 16.0);
    launch_normals(c.d_depth, c.d_K, c.d_normals, T, H, W, c.stream);
  }
  UG_CHECK(hipStreamSynchronize(c.stream));
}

void dc_get_outputs(Ctx& c, float* frames, float* depth, float* normals) {
  UG_REQUIRE(c.io_ready, "no outputs");
  const long px = (long)c.T * c.H * c.W;
  if (frames) UG_CHECK(hipMemcpy(frames, c.d_out_frames, px * 3 * 4, hipMemcpyDeviceToHost));
  if (depth) UG_CHECK(hipMemcpy(depth, c.d_depth, px * 4, hipMemcpyDeviceToHost));
  if (normals) UG_CHECK(hipMemcpy(normals, c.d_normals, px * 3 * 4, hipMemcpyDeviceToHost));
}

#include "sn_graphs.inc"

}  // namespace ug
