// Halo-staged 3x3 convolution for gfx950 (round 4): implicit GEMM whose activation operand is staged ONCE per 64-channel chunk.
//
// The im2col GEMM (kernels/gemm.hip, gemm_ws_kernel<.., CONV>) fetches a [256 x 64] activation tile per K step = per (tap, chunk): every
// activation row travels L2 -> LDS nine times.  The counters of round 4 (profiles/r04_gemm_l2_counters.txt) say the L2 is ~10 % busy while
// these kernels run - what paces a K step on the 256 x 128 / 256 x 160 tiles is the CU's own vector-memory -> LDS path (46 B/clk measured):
// (256 + 160) x 64 x 2 B = 52 KiB per step against 1288 cycles of MFMAs.  Here a tile is TH x TW = 256 output pixels of ONE frame and the
// loader brings the (TH + 2) x (TW + 2) pixel halo of a 64-channel chunk into LDS once; the nine taps of that chunk are nine K steps that
// read the SAME LDS image at a shifted pixel offset.  Per K step the CU now pulls BN x 128 B of weights + 1/9 of a <= 50 KiB halo:
// 26 KiB instead of 52 (BN = 160), 21 instead of 48 (BN = 128).
//
// Structure = the producer / consumer kernel: 8 consumer waves (4 x 2, wave tile 64 x BN/2, 16x16x32 f16 MFMA, fp32 accumulate) + 4 fetch
// waves, one s_barrier per K step, weights through a 3-slot ring two steps ahead (direct-to-LDS buffer loads).  The halo is double
// buffered: the halo of chunk c + 1 is fetched in slices piggy-backed on the weight fetches of chunk c's steps 2 .. 8 (the buffer it
// goes to was last read in chunk c - 1, which ended one barrier before step 0 of chunk c is computed - so slices start with step 2's
// fetch, which is issued while step 0 is computed).  Weights are in the chunk-major K order ([N][C / 64][9][64], GemmP::kchunk) the
// im2col path already uses, the W-row permutation and the epilogue are gemm_kernel's: same products, same K order, same fp32 chains
// => outputs are bit-identical to the im2col GEMM.
// Geometry is compile time (template LG: TW = 1 << LG): the halo's row pitch is TW + 2 rounded up to 8 pixels, so every MFMA row block of a wave
// sits a multiple of 8 pixels from the first one (same swizzle) and a tap's dy moves by whole pitches - ALL fragment addresses of a K step are
// one per-lane base (per dx and k half: six registers, loop invariant) + an immediate; the nine taps of a chunk are unrolled, the weight ring
// slot is tap % 3.  The consumers execute no address arithmetic inside the K loop.
// LDS: 2 x halo (<= 432 px x 128 B) + 3 x BN x 128 B <= 160 KiB.
#include "gemm_common.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

template <int BM, int BN, int LG, bool ST = false>   // ST: GroupNorm statistics of the output from the epilogue (GemmP::stat_part)
__global__ __launch_bounds__(12 * 64, 3) void conv_halo_kernel(const GemmP p) {
  constexpr int BK = 64, NCW = 8, WMW = 4, WNW = 2;
  constexpr unsigned SENT = 0x80000000u;
  constexpr int WTM = BM / WMW, WTN = BN / WNW;
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int WID = 4 * NT;
  constexpr int LB = BN / 32;                 // 1 KiB weight loads per fetch wave per K step
  constexpr int WSLOT = BN * BK;              // halves per weight slot
  constexpr int TW = 1 << LG, TH = BM >> LG;
  constexpr int PITCH = (TW + 2 + 7) & ~7;    // halo row pitch in pixels
  constexpr int GPR = PITCH / 8;              // 8-pixel load groups per halo row
  constexpr int NG = (TH + 2) * GPR;          // load groups per halo
  constexpr int HALO = NG * 8 * BK;           // halves per halo buffer
  constexpr int HL = 2;                       // halo slices (1 KiB loads) per fetch wave per K step, steps 2 .. 8 of a chunk
  constexpr int HSLOTS = 7 * HL;
  static_assert(TH * TW == BM && TW >= 16 && (NG + 3) / 4 <= HSLOTS, "tile geometry");
  static_assert((2 * HALO + 3 * WSLOT) * 2 <= 160 * 1024, "LDS");
  extern __shared__ __attribute__((aligned(16))) f16 smem[];
  f16* halo = smem;                           // [2][HALO]
  f16* ring = smem + 2 * HALO;                // [3][BN][64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= NCW;
  const int ntn = (p.N + BN - 1) / BN, ntm = p.M / BM;
  const int ntiles = ntm * ntn;
  const int Cin = p.C0 + p.C1, nchunks = Cin / BK, nk = nchunks * 9;
  const int txn = p.Wo >> LG, tpf = (p.Ho / TH) * txn;     // tiles per image row / per frame

  const int nwg = gridDim.x, w = blockIdx.x;
  const TileWalk tw = tile_walk(p.flags, ntiles, nwg, w);    // XCD-owned runs of the walk order (kernels/gemm_common.h)
  const int my_tiles = tw.count;
  const int total_it = my_tiles * nk;

  // tile -> frame / first output pixel (y0, x0) / first output column
  auto tile_origin = [&](int tile, int& t, int& y0, int& x0, int& n0) {
    int tm, tn; tile_coord_p(p, tile, ntm, ntn, tm, tn);
    const int tmg = tm + p.m_off / BM;
    t = tmg / tpf;
    const int rem = tmg - t * tpf;
    const int tyi = rem / txn, txi = rem - tyi * txn;
    y0 = tyi * TH; x0 = txi * TW; n0 = tn * BN;
  };

  if (producer) {
    // ================================= fetch waves =================================
    const int pw = wave - NCW;
    const int pc = lane & 7, lrow = lane >> 3;
    const int lc16_0 = (pc ^ ((0 + (lrow >> 1)) & 7)) * 16, lc16_1 = (pc ^ ((4 + (lrow >> 1)) & 7)) * 16;   // weight rows: gemm_kernel's swizzle
    // halo rows are read at ARBITRARY pixel offsets (tap shifts), where (row >> 1) & 7 gives two-way bank conflicts for three of four alignments
    // (ds_read_b128 serves lanes {0-3, 12-15, 20-27} together: the k-chunk bit that separates lanes 16+ flips in mid-window).  2 * ((row >> 1) & 3)
    // is conflict-free for every alignment (checked exhaustively); it does not depend on the 8-pixel group
    const int lc16_h = (pc ^ (((lrow >> 1) & 3) * 2)) * 16;
    const __amdgpu_buffer_rsrc_t rA0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A0, 0, (int)SENT, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A1 ? p.A1 : p.A0), 0, (int)SENT, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)SENT, 0x00020000);
    // halo slot k of this wave = 8-pixel group pw + 4 k = (halo row, 8-pixel column block); this lane's pixel of it
    constexpr int ngw_max = (NG + 3) / 4;
    const int ngw = (NG - pw + 3) >> 2;                 // groups of this wave
    int h_pix[HSLOTS];                                    // source pixel of every halo slot for the tile being fetched, or -1
    auto halo_setup = [&](int tile) {
      int t, y0, x0, n0; tile_origin(tile, t, y0, x0, n0);
#pragma unroll
      for (int k = 0; k < ngw_max; ++k) {
        const int grp = pw + 4 * k;
        const int hy = grp / GPR, hx = (grp - hy * GPR) * 8 + lrow;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool ok = grp < NG && hx < TW + 2 && y >= 0 && y < p.Hi && x >= 0 && x < p.Wi;
        h_pix[k] = ok ? (t * p.Hi + y) * p.Wi + x : -1;
      }
    };
    // one halo slice: slot k of chunk `chunk` (channel base chunk * 64) into halo buffer `hb`
    auto halo_issue = [&](int k, int chunk, int hb) {
      const int cb = chunk * BK;
      const bool src0 = cb < p.C0;
      const int Cs2 = (src0 ? p.C0 : p.C1) * 2;
      const int soff = (src0 ? cb : cb - p.C0) * 2;
      const unsigned voff = h_pix[k] >= 0 ? (unsigned)h_pix[k] * (unsigned)Cs2 + (unsigned)lc16_h : SENT;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(src0 ? rA0 : rA1, (lptr_t)(halo + hb * HALO + (pw + 4 * k) * 8 * BK), 16, (int)voff, soff, 0, 0);
    };
    unsigned b_off[LB];
    int ld_ti = 0, ld_ks = 0, ld_slot = 0;      // weight fetch position: tile (of this workgroup), K step inside the tile, ring slot
    int ld_gc = 0;                              // global chunk counter of the weight fetch position (halo buffer = parity)
    int h_tile = -1;                            // tile h_pix belongs to
    // weights of one K step (+ the halo slices that ride with it); returns the number of halo loads issued
    auto issue = [&]() -> int {
      if (ld_ks == 0) {
        int t, y0, x0, n0; tile_origin(tw.first + ld_ti * tw.step, t, y0, x0, n0);
#pragma unroll
        for (int l = 0; l < LB; ++l) {
          const int rg = pw * LB + l;
          const int lr = rg * 8 + lrow;                 // LDS row of the W tile; holds W row n0 + perm(lr)
          const int part = lr / WTN, rem = lr % WTN;
          const int jj = rem >> 4, i = rem & 15;
          const int n = n0 + part * WTN + (i >> 2) * WID + jj * 4 + (i & 3);
          b_off[l] = n < p.N ? (unsigned)n * (unsigned)(p.ldw * 2) + ((rg & 1) ? lc16_1 : lc16_0) : SENT;
        }
      }
      f16* Bs = ring + ld_slot * WSLOT;
      const int kt2 = ld_ks * BK * 2;
#pragma unroll
      for (int l = 0; l < LB; ++l)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(Bs + (pw * LB + l) * 8 * BK), 16, (int)b_off[l], kt2, 0, 0);
      int nh = 0;
      const int chunk = ld_ks / 9, sp = ld_ks - chunk * 9;
      if (sp >= 2) {
        // halo of the NEXT chunk (same tile, or chunk 0 of this workgroup's next tile) -> the other halo buffer
        int tchunk = chunk + 1, ttile = ld_ti;
        if (tchunk == nchunks) { tchunk = 0; ++ttile; }
        if (ttile < my_tiles) {
          if (ttile != h_tile) { halo_setup(tw.first + ttile * tw.step); h_tile = ttile; }
#pragma unroll
          for (int u = 0; u < HL; ++u) {
            const int k = (sp - 2) * HL + u;
            if (k < ngw) {
              // (k is uniform: select the slot's registers without dynamic indexing)
#pragma unroll
              for (int kk = 0; kk < ngw_max; ++kk) if (kk == k) halo_issue(kk, tchunk, (ld_gc + 1) & 1);
              ++nh;
            }
          }
        }
      }
      if (++ld_ks == nk) { ld_ks = 0; ++ld_ti; }
      if (ld_ks % 9 == 0) ++ld_gc;
      if (++ld_slot == 3) ld_slot = 0;
      return nh;
    };
    // prologue: the whole halo of the first chunk, then steps 0 and 1; step 0 (and the halo, older) must have landed before the first barrier
    if (total_it > 0) {
      halo_setup(tw.first); h_tile = 0;
#pragma unroll
      for (int k = 0; k < ngw_max; ++k) if (k < ngw) halo_issue(k, 0, 0);
      issue();
    }
    if (total_it > 1) issue();            // steps 0 / 1 carry no halo slices (sp < 2)
    if (total_it > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int fi = 0; fi < total_it; ++fi) {
      if (fi + 2 < total_it) {
        const int nh = issue();           // step fi + 2 -> the slot the consumers left before the previous barrier
        if (nh == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LB + 2) : "memory");       // step fi + 1 has landed; only this step's loads stay in flight
        else if (nh == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LB + 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LB) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // ================================= compute waves =================================
  const int wm = wave / WNW, wn = wave % WNW;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, g = lane >> 4;
  const int sw = swz<BK>(l15);
  // Row block i of this wave = tile rows wm*WTM + i*16 + [0, 16): pixel ((r >> LG), (r & (TW-1))) of the tile, halo pixel y * PITCH + x at tap
  // (0, 0).  Block i sits DI(i) pixels after block 0 - a multiple of 8 (pitch and 16-pixel runs), so it shares block 0's swizzle.
  const int r0 = wm * WTM;
  const int hpb = (r0 >> LG) * PITCH + (r0 & (TW - 1)) + l15;
  auto DI = [](int i) constexpr { return TW >= 64 ? i * 16 : TW == 32 ? (i >> 1) * PITCH + (i & 1) * 16 : i * PITCH; };
  static_assert(TW >= 64 || (TW == 32 && WTM % 32 == 0) || TW == 16, "row-block offsets");
  // byte offsets (inside a halo buffer) of this lane's fragment of block 0 at tap (0, dx), k half kk; weights: (inside a ring slot)
  int a_pre[3][2], b_pre[2];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int hp = hpb + dx;
      a_pre[dx][kk] = (hp * BK + (((kk * 4 + g) ^ (((hp >> 1) & 3) * 2)) * 8)) * 2;
    }
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) b_pre[kk] = ((wn * WTN + l15) * BK + (((kk * 4 + g) ^ sw) * 8)) * 2;
  const char* ringb = (const char*)ring;
  int cp_ti = 0, cp_gc = 0;
  __builtin_amdgcn_s_barrier();          // step 0 (and the first halo) published
  asm volatile("" ::: "memory");
  const char* Hb = nullptr;
  int ti = 0;
  bool last_chunk = false;
  const bool lean = !ST && BN != 160 && tile_epilogue_lean_ok(p, WTN, true) && !(p.tune_knobs & 8388608);   // (the 160-column tiles have no register to spare)   // the epilogue without its run-time variants (kernels/gemm_common.h)
  // one K step = one tap of the current chunk; TAP is a compile-time constant (the nine taps are nine instantiations), so dx selects a_pre
  // statically and dy / the row-block offset / the ring slot are immediates
  auto step = [&](auto TAP) {
    constexpr int tap = decltype(TAP)::value;
    constexpr int dy = tap / 3, dx = tap - dy * 3;
    constexpr int slot = tap % 3;                              // nine steps per chunk: the ring phase is the tap's
    {
      f16x8 af[2][MT], bf[2][NT];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[kk][j] = *(const f16x8*)(ringb + b_pre[kk] + (slot * WSLOT + j * 16 * BK) * 2);
#pragma unroll
        for (int i = 0; i < MT; ++i) af[kk][i] = *(const f16x8*)(Hb + a_pre[dx][kk] + (dy * PITCH + DI(i)) * BK * 2);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
      constexpr int R = MT + NT, Q = MT * NT;
      __builtin_amdgcn_sched_group_barrier(0x100, R, 0);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, Q / R, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, Q - R * (Q / R), 0);
      __builtin_amdgcn_sched_group_barrier(0x008, Q, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // all fragment reads of this slot / this halo are done before they are handed back
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (tap == 8 && last_chunk) {   // after the barrier: the fetch waves go on with the next tile's steps while the compute waves are in the epilogue (gemm_ws_kernel, round 6)
      int t, y0, x0, n0; tile_origin(tw.first + ti * tw.step, t, y0, x0, n0);
      const int m0 = (t * p.Ho + y0) * p.Wo + x0 - p.m_off;
      if constexpr (ST) tile_epilogue_stats<MT, NT, WTM, WTN>(p, acc, m0, n0, wm, wn, lane, (t * tpf + (y0 / TH) * txn + (x0 >> LG)) * (BM / WTM) + wm);
      else if constexpr (BN != 160) { if (lean) tile_epilogue_lean<MT, NT, WTM, WTN, false, true>(p, acc, m0, n0, wm, wn, lane, 0, nullptr); else tile_epilogue<MT, NT, WTM, WTN>(p, acc, m0, n0, wm, wn, lane, 0); }
      else tile_epilogue<MT, NT, WTM, WTN>(p, acc, m0, n0, wm, wn, lane, 0);
    }
  };
  for (ti = 0; ti < my_tiles; ++ti) {
    for (int c = 0; c < nchunks; ++c, ++cp_gc) {
      Hb = (const char*)(halo + (cp_gc & 1) * HALO);
      last_chunk = c == nchunks - 1;
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
      step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
    }
  }
  (void)cp_ti;
}

// tile geometry: TH x TW = BM output pixels of one frame (TW = 1 << lg dividing Wo, TH dividing Ho); row-split launches need whole bands of TH image rows
static bool halo_geom_ok(const GemmP& p, int bm, int lg) {
  const int tw = 1 << lg, th = bm >> lg;
  if (th < 1 || th * tw != bm || p.Wo % tw || p.Ho % th) return false;
  const long band = (long)th * p.Wo;
  return p.m_off % band == 0 && p.M % band == 0;
}
// the geometries that are instantiated, per tile, in order of preference (smallest halo first)
static int halo_pick_lg(const GemmP& p, int bm, int bn) {
  if (bm == 256 && bn == 160) return halo_geom_ok(p, 256, 5) ? 5 : -1;            // 8 x 32: 2 x 50 KiB + 60 KiB = exactly 160 KiB
  if (bm == 256) { for (int lg : {4, 5, 6}) if (halo_geom_ok(p, 256, lg)) return lg; return -1; }
  return halo_geom_ok(p, 192, 4) ? 4 : -1;                                          // 12 x 16
}

bool conv_halo_supported(const GemmP& p, int batch, int bm, int bn) {
  if (!p.conv || batch != 1 || p.kt != 1 || p.ky != 3 || p.kx != 3 || p.stride != 1 || p.ups != 1 || p.pad_t != 1 || p.pad_l != 1) return false;
  if (!p.kchunk || p.up_phase || p.splitk > 1 || p.Hi != p.Ho || p.Wi != p.Wo) return false;
  const int Cin = p.C0 + p.C1;
  if (Cin % 64 || p.C0 % 64 || p.M % bm || p.m_off % bm || p.N % 16) return false;
  const long lim = (1L << 31) - 64;
  if ((long)p.T * p.Hi * p.Wi * std::max(p.C0, p.C1) * 2 >= lim || (long)p.N * p.ldw * 2 >= lim) return false;
  if (p.m_off == 0 && p.M != p.T * p.Ho * p.Wo) return false;
  return halo_pick_lg(p, bm, bn) >= 0;
}

template <int BM, int BN, int LG, bool ST = false>
static void launch_halo_t(const GemmP& p, hipStream_t s) {
  constexpr int TW = 1 << LG, TH = BM >> LG, PITCH = (TW + 2 + 7) & ~7, NG = (TH + 2) * (PITCH / 8);
  const size_t lds = 2 * (size_t)NG * 1024 + 3 * (size_t)BN * 128;
  static bool attr[32] = {};
  bool& at = attr[ug_dev_slot()];
  if (!at) {
    UG_CHECK(hipFuncSetAttribute((const void*)conv_halo_kernel<BM, BN, LG, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    at = true;
  }
  const int ntiles = (p.M / BM) * cdiv(p.N, BN);
  int gx = std::min(256, ntiles);
  static const bool dbg = getenv("UG_HALO_DEBUG") != nullptr;
  if (dbg) fprintf(stderr, "[halo] %dx%d M %d (m_off %d) N %d C %d+%d  %dx%d  tile %dx%d pitch %d  lds %zu  grid %d\n", BM, BN, p.M, p.m_off, p.N, p.C0, p.C1, p.Ho, p.Wo, TH, TW, PITCH, lds, gx);
  GemmP q = p; q.halo_lg = LG; q.halo_tw = TW; q.splitk = 1;
  hipLaunchKernelGGL((conv_halo_kernel<BM, BN, LG, ST>), dim3(gx), dim3(12 * 64), lds, s, q);
}

void launch_conv_halo(const GemmP& p, int bm, int bn, hipStream_t s) {
  const int lg = halo_pick_lg(p, bm, bn);
  UG_REQUIRE(lg >= 0, "halo conv: no tile geometry");
  if (p.stat_part) {   // (launch_gemm hands the statistics buffer only to the 128-column tiles)
    UG_REQUIRE(bn == 128, "halo conv: epilogue statistics on 128-column tiles only");
    if (bm == 192) launch_halo_t<192, 128, 4, true>(p, s);
    else if (lg == 4) launch_halo_t<256, 128, 4, true>(p, s); else if (lg == 5) launch_halo_t<256, 128, 5, true>(p, s); else launch_halo_t<256, 128, 6, true>(p, s);
    return;
  }
  if (bm == 256 && bn == 160) launch_halo_t<256, 160, 5>(p, s);
  else if (bm == 256) { if (lg == 4) launch_halo_t<256, 128, 4>(p, s); else if (lg == 5) launch_halo_t<256, 128, 5>(p, s); else launch_halo_t<256, 128, 6>(p, s); }
  else if (bn == 160) launch_halo_t<192, 160, 4>(p, s);
  else launch_halo_t<192, 128, 4>(p, s);
}
