// fp16 MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X, CDNA4).
//
//   Out[M,N] = epilogue( A[M,K] . W[N,K]^T )
//
// * BM x BN x BK block tile, WMW x WNW wave64 grid, every wave owns a (BM/WMW) x (BN/WNW) accumulator made of
//   16x16x32 f16 MFMA tiles (fp32 accumulate).  Tile shapes are picked per problem (launch_gemm).
// * Operand tiles are staged global -> LDS with the direct-to-LDS 16-byte loads (global_load_lds_dwordx4): no
//   VGPR round trip.  The LDS image is lane-linear, so the XOR swizzle that makes the ds_read_b128 fragment
//   reads bank-conflict free is applied on the per-lane *source* address and mirrored on the read side.
// * Double-buffered LDS, one barrier per K-tile: the prefetch of tile t+1 is issued right after the barrier
//   and lands while tile t is multiplied.
// * The A operand is either a dense row-major matrix or the im2col view of channels-last tensors generated on
//   the fly by the address computation (taps over t/y/x, stride, nearest-2x upsample, concat of two sources).
//   Padding / out-of-range rows read a zero page.
// * MFMA operands are swapped (W fragment as "A", activation fragment as "B") and the W rows of a tile are
//   permuted when they are staged, so every lane ends up with 4*NT *contiguous* output columns of one output
//   row: bias / residual / GEGLU / store are all 16-byte vector operations.
// * split-K (long-K, few-tile problems: the low-resolution UNet levels): gridDim.y slices of K write fp32
//   partial tiles; splitk_epilogue sums them in a fixed order (deterministic) and applies the epilogue.
#include "../common.h"
#include <algorithm>

#include "gemm_common.h"

// waves per SIMD the LDS footprint allows (workgroups per CU x waves per workgroup / 4 SIMDs): handed to
// __launch_bounds__ so the register allocator does not trade that occupancy away.
template <int BM, int BN, int BK, int NST, int WMW, int WNW> struct GemmOcc {
  static constexpr int lds = NST * (BM + BN) * BK * 2;
  static constexpr int wg_lds = (160 * 1024) / lds;
  static constexpr int wg = wg_lds < 1 ? 1 : (wg_lds > 4 ? 4 : wg_lds);
  static constexpr int wps_raw = wg * WMW * WNW / 4;
  static constexpr int wps = wps_raw < 1 ? 1 : (wps_raw > 4 ? 4 : wps_raw);
};

#ifdef UG_GEMM_TRACE
#define UG_STAMP(slot)                                                                                   \
  do {                                                                                                   \
    if (traced && fi >= 8 && fi < 32) {                                                                  \
      unsigned long long c_;                                                                             \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c_)::"memory");                         \
      if (lane == 0) tr_lds[(fi - 8) * 5 + (slot)] = (unsigned)c_;                                       \
    }                                                                                                    \
  } while (0)
#else
#define UG_STAMP(slot) do {} while (0)
#endif

// BUFA: operands are fetched with buffer addressing (buffer_load_dwordx4 ... offen lds): the per-lane byte offset of a
// row is computed once per tile, the K / tap advance is a scalar offset, and out-of-range rows / padding taps point the
// lane past num_records, where the hardware returns zeros.  That takes the ~16 VALU instructions per load that the
// 64-bit flat addressing (pointer arithmetic + zero-page select) costs down to a compare-and-select; measured with
// SQ_ACTIVE_INST_VALU that address arithmetic was ~1/4 of a K-step on the 8-wave tiles and more on the small ones.
// Requires 32-bit offsets (operands < 2 GiB), whole K tiles (dense K % BK == 0) and, for im2col, the single-tap-per-
// K-tile form without the nearest-2x source mapping; launch_mode checks this and otherwise uses the flat-address form.
// MX: the operands are OCP e4m3 bytes with e8m0 block scales (v_mfma_scale_f32_16x16x128_f8f6f4, 2x the fp16 MFMA rate).  The
// loaders are byte movers, so an fp8 [rows][K] matrix is staged exactly like an fp16 [rows][K/2] one (the launcher halves K / lda /
// ldw): a 128-byte LDS row is one 128-deep K step.  The block scales of a K step (one dword = 4 e8m0 per row) ride the same ring:
// wave 0 / wave 1 fetch the A / W scale dwords of the tile's rows with one extra 1 KiB direct-to-LDS load each.
template <int BM, int BN, int BK, int NST, int WMW, int WNW, bool CONV, bool UNI, bool BUFA = false, bool MX = false, bool ST = false>   // ST: statistics epilogue (GemmP::stat_part)
__global__ __launch_bounds__(WMW * WNW * 64, (GemmOcc<BM, BN, BK, NST, WMW, WNW>::wps)) void gemm_kernel(const GemmP p) {
  static_assert(!BUFA || !CONV || UNI, "buffer addressing needs the single-tap K tiles");
  static_assert(!MX || (!CONV && BK == 64 && BM <= 256 && BN <= 256), "MX path: dense, 128-byte K steps, <= 256 scale rows per operand");
  constexpr unsigned SENT = 0x80000000u;   // == num_records: every lane offset >= SENT reads zeros
  constexpr int NWAVE = WMW * WNW;
  constexpr int WTM = BM / WMW, WTN = BN / WNW;     // wave tile
  constexpr int MT = WTM / 16, NT = WTN / 16;       // MFMA tiles per wave
  constexpr int WID = 4 * NT;                       // contiguous output columns per lane
  constexpr int CPR = BK / 8;                       // 16-byte chunks per LDS row
  constexpr int RPI = 64 / CPR;                     // rows covered by one wave-wide 1 KiB load
  constexpr int AI = BM / RPI / NWAVE;              // A load instructions per wave per K-tile
  constexpr int BI = BN / RPI / NWAVE;              // W load instructions per wave per K-tile
  constexpr int STAGE = (BM + BN) * BK;             // halves per pipeline stage
  static_assert(AI >= 1 && BI >= 1 && (BM / RPI) % NWAVE == 0 && (BN / RPI) % NWAVE == 0, "tile/wave mismatch");
  static_assert(NST >= 2 && NST <= 4, "pipeline depth");
  extern __shared__ __attribute__((aligned(16))) f16 smem[];   // [NST][A tile | W tile]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNW, wn = wave % WNW;
  // static priority for the second-dispatched half of an 8-wave workgroup (it loses every arbitration against its SIMD partner
  // otherwise): +0..6 % on the 256x256 tile, neutral-to-negative on the others (knob 16 forces it there for A/B runs)
  if (NWAVE == 8 && wave >= 4 && ((BM == 256 && BN == 256) || (p.flags & UG_F_PRIO))) __builtin_amdgcn_s_setprio(1);
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int ntiles = ntm * ntn;

  const int bz = blockIdx.z;
  const int bo = bz / p.nb_inner, bi = bz - bo * p.nb_inner;
  const f16* A0 = p.A0 + bo * p.sA_o + bi * p.sA_i;
  const f16* Wb = p.W + bo * p.sW_o + bi * p.sW_i;
  const long out_off = bo * p.sO_o + bi * p.sO_i;

  const int pc = lane % CPR;         // physical 16-byte chunk inside the LDS row
  const int lrow = lane / CPR;       // row inside one 1 KiB load
  const int Cin = p.C0 + p.C1;

  // K range of this split
  const int nk_all = (p.K + BK - 1) / BK;
  int kt_lo = 0, kt_hi = nk_all;
  if (p.splitk > 1) {
    const int per = (nk_all + p.splitk - 1) / p.splitk;
    kt_lo = min(nk_all, (int)blockIdx.y * per);
    kt_hi = min(nk_all, kt_lo + per);
  }
  const int nk = max(kt_hi - kt_lo, 1);   // an empty split still walks one (all-zero-source) step

  // Persistent tile walk.  Workgroup w (observed to run on XCD w % 8) walks the contiguous run of tiles its XCD owns (tile_walk):
  // the workgroups of one XCD work on neighbouring tiles (same A rows, neighbouring W panels) at the same time and from one
  // round to the next, so their operand panels hit in that XCD's L2.
  const int nwg = gridDim.x, w = blockIdx.x;
  const TileWalk tw = tile_walk(p.flags, ntiles, nwg, w);    // tiles tw.first, tw.first + tw.step, ... (XCD-owned runs: kernels/gemm_common.h)
  const int my_tiles = tw.count;
  const int total_it = my_tiles * nk;

  // ---- loader state for the tile currently being fetched ----
  int a_lc[AI]; bool a_ok[AI]; const f16* a_ptr[AI]; int a_t[AI], a_y[AI], a_x[AI];
  int a_pix[AI]; unsigned a_mask[AI];   // fast conv path: pixel index of tap (0,0,0) and per-tap validity bits
  const bool fastconv = CONV && UNI;
  int a_par[AI];                        // ups == 2: parity bits (y&1)<<1 | (x&1) of the row's first tap in upsampled coords
  int b_lc[BI]; const f16* b_ptr[BI]; bool b_ok[BI];
  unsigned a_off[AI], a_off1[AI], b_off[BI];   // BUFA: byte offsets from the (shifted) operand bases, or SENT
  // im2col: the base is moved back by the largest negative tap displacement so that every row offset is >= 0
  const int cshift = CONV ? ((p.kt >> 1) * p.Hi + p.pad_t) * p.Wi + p.pad_l : 0;
  __amdgpu_buffer_rsrc_t rA0, rA1, rW;
  if (BUFA) {
    rA0 = __builtin_amdgcn_make_buffer_rsrc((void*)(A0 - (long)cshift * p.C0), 0, (int)SENT, 0x00020000);
    rA1 = __builtin_amdgcn_make_buffer_rsrc((void*)((CONV && p.A1) ? p.A1 - (long)cshift * p.C1 : A0), 0, (int)SENT, 0x00020000);
    rW = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, (int)SENT, 0x00020000);
  }
#pragma unroll
  for (int j = 0; j < AI; ++j) a_lc[j] = pc ^ swz<BK>((wave * AI + j) * RPI + lrow);
#pragma unroll
  for (int j = 0; j < BI; ++j) b_lc[j] = pc ^ swz<BK>((wave * BI + j) * RPI + lrow);

  int ld_m0 = 0, ld_n0 = 0;   // MX: origin of the tile being fetched (scale rows)
  auto setup_tile = [&](int tile) {
    int tm, tn; tile_coord_p(p, tile, ntm, ntn, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    ld_m0 = m0; ld_n0 = n0;
#pragma unroll
    for (int j = 0; j < AI; ++j) {
      const int m = m0 + (wave * AI + j) * RPI + lrow;
      a_ok[j] = m < p.M;
      if (CONV) {
        const int hw = p.Ho * p.Wo;
        const int t = m / hw, rem = m - t * hw;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        a_t[j] = t; a_y[j] = oy * p.stride - p.pad_t; a_x[j] = ox * p.stride - p.pad_l;
        a_ptr[j] = nullptr;
        if (fastconv) {
          unsigned mk = 0; int bit = 0;
          for (int it = 0; it < p.kt; ++it)
            for (int iy = 0; iy < p.ky; ++iy)
              for (int ix = 0; ix < p.kx; ++ix, ++bit) {
                const int tt = t + it - (p.kt >> 1), y = a_y[j] + iy, x = a_x[j] + ix;
                if (tt >= 0 && tt < p.T && y >= 0 && y < p.Hi * p.ups && x >= 0 && x < p.Wi * p.ups) mk |= 1u << bit;
              }
          a_mask[j] = a_ok[j] ? mk : 0u;
          if (BUFA) {
            const unsigned ap = (unsigned)(((t - (p.kt >> 1)) * p.Hi + a_y[j]) * p.Wi + a_x[j] + cshift);
            a_off[j] = ap * (unsigned)(p.C0 * 2) + a_lc[j] * 16;
            a_off1[j] = ap * (unsigned)(p.C1 * 2) + a_lc[j] * 16;
          } else if (p.ups == 1) {
            a_pix[j] = ((t - (p.kt >> 1)) * p.Hi + a_y[j]) * p.Wi + a_x[j];
            a_par[j] = 0;
          } else {   // nearest-2x source: pixel of tap (iy,ix) = base + ((iy+py)>>1)*Wi + ((ix+px)>>1)
            a_pix[j] = ((t - (p.kt >> 1)) * p.Hi + (a_y[j] >> 1)) * p.Wi + (a_x[j] >> 1);
            a_par[j] = ((a_y[j] & 1) << 1) | (a_x[j] & 1);
          }
        }
      } else {
        a_ptr[j] = A0 + (long)m * p.C0;
        a_t[j] = a_y[j] = a_x[j] = 0;
        if (BUFA) a_off[j] = (a_ok[j] && kt_lo < kt_hi) ? (unsigned)m * (unsigned)(p.C0 * 2) + a_lc[j] * 16 : SENT;
      }
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {   // LDS row lr holds W row n0 + perm(lr)
      const int lr = (wave * BI + j) * RPI + lrow;
      const int part = lr / WTN, rem = lr % WTN;
      const int jj = rem >> 4, i = rem & 15;
      const int n = n0 + part * WTN + (i >> 2) * WID + jj * 4 + (i & 3);
      b_ok[j] = n < p.N;
      b_ptr[j] = Wb + (long)n * p.ldw;
      if (BUFA) b_off[j] = (b_ok[j] && kt_lo < kt_hi) ? (unsigned)n * (unsigned)(p.ldw * 2) + b_lc[j] * 16 : SENT;
    }
  };

  int ld_ti = 0, ld_ks = 0, ld_slot = 0;         // loader position: tile index (of this workgroup), K-step, ring slot
  int u_it = 0, u_iy = 0, u_ix = 0, u_cb = 0;   // UNI conv: tap (t,y,x) and channel base of the next staged K tile
  auto stage = [&](int kt0, int buf) {
    f16* As = smem + buf * STAGE;
    f16* Bs = As + BM * BK;
    int it = 0, iy = 0, ix = 0, cb = 0;
    if (CONV && UNI) {  // whole K tile sits inside one tap
      if (ld_ks == 0) {   // first K tile of a tile: decode once (split-K may start mid-way)
        // chunk-major K: step q = kt0 / BK is (channel chunk q / taps, tap q % taps)
        const int q = kt0 / BK, ntap = p.kt * p.ky * p.kx;
        const int chk = q / ntap;
        const int tap = q - chk * ntap;
        u_cb = chk * BK;
        u_it = tap / (p.ky * p.kx);
        const int r2 = tap - u_it * (p.ky * p.kx);
        u_iy = r2 / p.kx; u_ix = r2 - u_iy * p.kx;
      }
      it = u_it; iy = u_iy; ix = u_ix; cb = u_cb;
      if (fastconv) {
        // every lane: source = base + (row pixel + uniform tap pixel offset) * C + channel, or the zero page
        const int tapbit = (it * p.ky + iy) * p.kx + ix;
        const int tappix = (it * p.Hi + iy) * p.Wi + ix;   // ups == 1
        const int up2 = p.ups == 2;
        const bool src0 = cb < p.C0;               // C0 % BK == 0 is checked by the launcher: a K tile never straddles
        const f16* base = src0 ? A0 : p.A1;
        const int Cs = src0 ? p.C0 : p.C1, c0 = src0 ? cb : cb - p.C0;
        if (BUFA) {
          const int soff = (tappix * Cs + c0) * 2;
          const __amdgpu_buffer_rsrc_t rs = src0 ? rA0 : rA1;
#pragma unroll
          for (int j = 0; j < AI; ++j) {
            const unsigned voff = ((a_mask[j] >> tapbit) & 1u) ? (src0 ? a_off[j] : a_off1[j]) : SENT;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(As + (wave * AI + j) * RPI * BK), 16, (int)voff, soff, 0, 0);
          }
        } else
#pragma unroll
        for (int j = 0; j < AI; ++j) {
          const bool ok = (a_mask[j] >> tapbit) & 1u;
          const int tp = up2 ? it * p.Hi * p.Wi + ((iy + (a_par[j] >> 1)) >> 1) * p.Wi + ((ix + (a_par[j] & 1)) >> 1) : tappix;
          const f16* src = ok ? base + ((long)(a_pix[j] + tp) * Cs + c0 + a_lc[j] * 8) : p.zero;
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (wave * AI + j) * RPI * BK), 16, 0, 0);
        }
      }
      // chunk-major K order (the only one the single-tap-per-K-tile paths take): next tap of the same 64-channel chunk, then the next chunk
      if (++u_ix == p.kx) { u_ix = 0; if (++u_iy == p.ky) { u_iy = 0; if (++u_it == p.kt) { u_it = 0; u_cb += BK; } } }
    }
    if (BUFA && !CONV) {
#pragma unroll
      for (int j = 0; j < AI; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA0, (lptr_t)(As + (wave * AI + j) * RPI * BK), 16, (int)a_off[j], kt0 * 2, 0, 0);
    }
    if (!fastconv && !(BUFA && !CONV))
#pragma unroll
    for (int j = 0; j < AI; ++j) {
      const f16* src = p.zero;
      const int k = kt0 + a_lc[j] * 8;
      if (a_ok[j] && k < p.K) {
        if (!CONV) {
          src = a_ptr[j] + k;
        } else {
          int c;
          if (UNI) {
            c = cb + a_lc[j] * 8;
          } else {
            int tap = k / Cin;
            c = k - tap * Cin;
            if (p.kchunk) { const int q = k / 64, ntap = p.kt * p.ky * p.kx, chk = q / ntap; tap = q - chk * ntap; c = chk * 64 + (k & 63); }
            it = tap / (p.ky * p.kx);
            const int r2 = tap - it * (p.ky * p.kx);
            iy = r2 / p.kx; ix = r2 - iy * p.kx;
          }
          const int tt = a_t[j] + it - (p.kt >> 1);
          const int y = a_y[j] + iy, x = a_x[j] + ix;
          if (tt >= 0 && tt < p.T && y >= 0 && y < p.Hi * p.ups && x >= 0 && x < p.Wi * p.ups) {
            const long pix = ((long)tt * p.Hi + (y / p.ups)) * p.Wi + (x / p.ups);
            src = (c < p.C0) ? (A0 + pix * p.C0 + c) : (p.A1 + pix * p.C1 + (c - p.C0));
          }
        }
      }
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (wave * AI + j) * RPI * BK), 16, 0, 0);
    }
    if (MX && wave < 2) {   // block scales of this K step: 4 rows' dwords per lane -> [BM | BN] dwords behind the operand ring
      unsigned* sc = (unsigned*)(smem + NST * STAGE) + buf * (BM + BN) + (wave ? BM : 0);
      const int rows = wave ? BN : BM;
      const unsigned* g = wave ? p.sw + (long)(kt0 / BK) * p.ld_sw + ld_n0 : p.sa + (long)(kt0 / BK) * p.ld_sa + ld_m0;
      // lanes past the tile's rows stay masked off: the 1 KiB image of a full-wave load would run over the neighbouring scale area
      if (4 * lane < rows) __builtin_amdgcn_global_load_lds((gptr_t)(g + 4 * lane), (lptr_t)sc, 16, 0, 0);
    }
    if (BUFA) {
#pragma unroll
      for (int j = 0; j < BI; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(Bs + (wave * BI + j) * RPI * BK), 16, (int)b_off[j], kt0 * 2, 0, 0);
    } else
#pragma unroll
    for (int j = 0; j < BI; ++j) {
      const int k = kt0 + b_lc[j] * 8;
      const f16* src = (b_ok[j] && k < p.K && kt0 < kt_hi * BK) ? (b_ptr[j] + k) : p.zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Bs + (wave * BI + j) * RPI * BK), 16, 0, 0);
    }
  };
  // fetch the next flat iteration (tile ld_ti of this workgroup, K-step ld_ks) into ring slot ld_slot
  auto issue = [&]() {
    if (ld_ks == 0) setup_tile(tw.first + ld_ti * tw.step);
    stage((kt_lo + ld_ks) * BK, ld_slot);
    if (++ld_ks == nk) { ld_ks = 0; ++ld_ti; }
    if (++ld_slot == NST) ld_slot = 0;
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, g = lane >> 4;
  const int sw = swz<BK>(l15);   // rows are (multiple of 16) + l15 and swz only looks at the low 4 row bits
#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0)
    if (s0 < total_it) issue();
  // 1: tile_epilogue_lean, 2: tile_epilogue_geglu_lean (64-column wave tiles), 0: tile_epilogue with all its run-time variants (kernels/gemm_common.h; knob 8388608 = always 0)
  // (the 256 x 256 tiles sit at the 256-register limit: only the GEGLU form - what the clip uses them for - and only in the dense kernels)
  const int lean_epi = (ST || (p.tune_knobs & 8388608)) ? 0 : (BM * BN < 65536 && tile_epilogue_lean_ok(p, WTN)) ? 1 : (NT == 4 && !CONV && tile_epilogue_geglu_lean_ok(p, WTN)) ? 2 : 0;
  bool drain = false;   // after an epilogue the in-flight count also holds its loads/stores: drain once
  int cp_ti = 0, cp_ks = 0, cp_slot = 0;
#ifdef UG_GEMM_TRACE
  // cycle stamps of waves 0 and 4 (same SIMD) of workgroup 0 for K-steps 8..31, kept in the LDS above the ring
  const bool traced = p.trace && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (wave == 0 || wave == 4);
  unsigned* tr_lds = (unsigned*)(smem + NST * STAGE) + (wave == 4 ? 24 * 5 : 0);
#endif
  for (int fi = 0; fi < total_it; ++fi) {
    UG_STAMP(0);   // previous step's MFMAs issued
    // wait for the loads of iteration fi (issued NST-1 iterations ago); up to NST-2 younger fetches stay in flight
    const int younger = min(NST - 2, total_it - 1 - fi);
    if (drain || younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (MX && wave < 2) {   // these two waves carry one scale load per stage on top
      if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AI + BI + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (AI + BI + 1)) : "memory");
    } else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AI + BI) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (AI + BI)) : "memory");
    drain = false;
    UG_STAMP(1);   // fetched operands have landed
    __builtin_amdgcn_s_barrier();    // raw barrier: does not drain the LDS-DMA queue
    asm volatile("" ::: "memory");   // keep this iteration's LDS reads / DMA issues below the barrier
    UG_STAMP(2);   // barrier passed
    if (fi + NST - 1 < total_it) issue();   // overwrites the slot every wave finished reading last iteration
    UG_STAMP(3);   // next fetch issued
    const f16* Ab = smem + cp_slot * STAGE + (wm * WTM + l15) * BK;
    const f16* Bb = smem + cp_slot * STAGE + BM * BK + (wn * WTN + l15) * BK;
    const unsigned* scl = (const unsigned*)(smem + NST * STAGE) + cp_slot * (BM + BN);
    if (++cp_slot == NST) cp_slot = 0;
    if (MX) {
      // operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 (measured: tools/microbench/mx8_probe.hip): lane (l15, g) holds K bytes
      // [16g, 16g+16) in registers 0-3 and [64+16g, 64+16g+16) in registers 4-7 of its row - two stacked 64-deep halves, i.e. the
      // logical 16-byte chunks g and 4+g - and supplies the e8m0 scale of K block g = K [32g, 32g+32) in byte 0 of the scale operand
      typedef int v8i __attribute__((ext_vector_type(8)));
      union Frag { v8i v; f16x8 h[2]; };
      Frag af[MT], bf[NT];
      int sA[MT], sW[NT];
      const int c0 = (g ^ sw) * 8, c1 = ((4 + g) ^ sw) * 8;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        bf[j].h[0] = *(const f16x8*)(Bb + j * 16 * BK + c0); bf[j].h[1] = *(const f16x8*)(Bb + j * 16 * BK + c1);
        sW[j] = (scl[BM + wn * WTN + (l15 >> 2) * WID + j * 4 + (l15 & 3)] >> (8 * g)) & 0xff;   // LDS row (wn, j, l15) holds this W row
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        af[i].h[0] = *(const f16x8*)(Ab + i * 16 * BK + c0); af[i].h[1] = *(const f16x8*)(Ab + i * 16 * BK + c1);
        sA[i] = (scl[wm * WTM + i * 16 + l15] >> (8 * g)) & 0xff;
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[j].v, af[i].v, acc[i][j], 0, 0, 0, sW[j], 0, sA[i]);
    } else {
      // Fragment reads are software-pipelined inside the K-step: all reads of the first 32-deep slice are issued up
      // front, the second slice's reads are interleaved under the first slice's MFMAs (order pinned below), so the
      // LDS latency is exposed once per step instead of once per MFMA group.
      constexpr int KK = BK / 32;
      f16x8 af[KK][MT], bf[KK][NT];
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const int ch = ((kk * 4 + g) ^ sw) * 8;
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[kk][j] = *(const f16x8*)(Bb + j * 16 * BK + ch);
#pragma unroll
        for (int i = 0; i < MT; ++i) af[kk][i] = *(const f16x8*)(Ab + i * 16 * BK + ch);
      }
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
      constexpr int R = MT + NT, Q = MT * NT;
      __builtin_amdgcn_sched_group_barrier(0x100, R, 0);               // slice 0 reads
      if (KK == 2) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          __builtin_amdgcn_sched_group_barrier(0x008, Q / R, 0);       // slice 0 MFMAs ...
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           // ... hiding slice 1 reads
        }
        __builtin_amdgcn_sched_group_barrier(0x008, Q - R * (Q / R), 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, Q, 0);
    }
#ifdef UG_GEMM_TRACE
    if (traced && fi == 40) {
      __builtin_amdgcn_s_waitcnt(0);
      for (int i = lane; i < 24 * 5; i += 64) ((unsigned*)p.trace)[(wave == 4 ? 24 * 5 : 0) + i] = tr_lds[i];
    }
#endif
    if (++cp_ks != nk) continue;
    cp_ks = 0;
    const int ti = cp_ti++;

    // ---- tile finished: epilogue (the next tile's operands keep streaming into the ring meanwhile) ----
    drain = true;
    const int tile = tw.first + ti * tw.step;
    { int etm, etn; tile_coord_p(p, tile, ntm, ntn, etm, etn);
      if constexpr (ST) tile_epilogue_stats<MT, NT, WTM, WTN>(p, acc, etm * BM, etn * BN, wm, wn, lane, etm * WMW + wm);
      else if constexpr (BM * BN < 65536) {
        if (lean_epi == 1) tile_epilogue_lean<MT, NT, WTM, WTN, false>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off, nullptr);
        else if constexpr (NT == 4 && !CONV) { if (lean_epi == 2) tile_epilogue_geglu_lean<MT, NT, WTM, WTN>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off); else tile_epilogue<MT, NT, WTM, WTN>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off); }
        else tile_epilogue<MT, NT, WTM, WTN>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off);
      } else if constexpr (NT == 4 && !CONV) { if (lean_epi == 2) tile_epilogue_geglu_lean<MT, NT, WTM, WTN>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off); else tile_epilogue<MT, NT, WTM, WTN>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off); }
      else tile_epilogue<MT, NT, WTM, WTN>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off); }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Asymmetric-loader variant of the one-barrier-per-K-step kernel for the 8-wave tiles (buffer addressing only).
//
// Cycle stamps of gemm_kernel (tools/gemm_trace.py, profiles/r01_gemm_kstep_trace.txt; 256x256x64, 4350 cycles per
// K-step): every wave spends ~1100 cycles right behind the barrier ISSUING its eight direct-to-LDS loads - the CU's 64
// loads drain through the texture-address path at ~17 cycles each and all eight waves queue on it while the matrix
// pipes idle - and then the two waves of each SIMD share the pipe for their 2 x 64 MFMAs.  Here only waves 4-7 (one
// per SIMD) fetch, each for itself and for its SIMD partner (waves 0-3): the partners go straight from the barrier to
// their fragment reads and MFMAs, so the matrix pipe runs the partner's K-step while the loader is queued on the
// address path, and the loader's K-step afterwards.  Same LDS image, ring, barrier and MFMA order as gemm_kernel
// (outputs are bit-identical).
template <int BM, int BN, int NST, int WMW, int WNW, bool CONV, bool ST = false>   // ST: statistics epilogue (GemmP::stat_part)
__global__ __launch_bounds__(512, 2) void gemm_ldr_kernel(const GemmP p) {
  constexpr int BK = 64;
  static_assert(WMW * WNW == 8, "two waves per SIMD");
  constexpr unsigned SENT = 0x80000000u;
  constexpr int WTM = BM / WMW, WTN = BN / WNW;
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int WID = 4 * NT;
  constexpr int AI = BM / 64, BI = BN / 64;         // 1 KiB loads per (virtual) wave per K-step
  constexpr int LA = 2 * AI, LB = 2 * BI;           // loads a loader wave issues per K-step
  constexpr int STAGE = (BM + BN) * BK;
  static_assert(AI >= 1 && BI >= 1 && NST >= 2 && NST <= 3, "tile / ring");
  static_assert(!CONV || AI % 2 == 0, "im2col swizzle parity trick needs an even AI");
  extern __shared__ __attribute__((aligned(16))) f16 smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNW, wn = wave % WNW;
  const bool loader = wave >= 4;
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int ntiles = ntm * ntn;

  const int bz = blockIdx.z;
  const int bo = bz / p.nb_inner, bi = bz - bo * p.nb_inner;
  const f16* A0 = p.A0 + bo * p.sA_o + bi * p.sA_i;
  const f16* Wb = p.W + bo * p.sW_o + bi * p.sW_i;
  const long out_off = bo * p.sO_o + bi * p.sO_i;

  const int pc = lane & 7, lrow = lane >> 3;
  const int Cin = p.C0 + p.C1;

  const int nk_all = (p.K + BK - 1) / BK;
  int kt_lo = 0, kt_hi = nk_all;
  if (p.splitk > 1) {
    const int per = (nk_all + p.splitk - 1) / p.splitk;
    kt_lo = min(nk_all, (int)blockIdx.y * per);
    kt_hi = min(nk_all, kt_lo + per);
  }
  const int nk = max(kt_hi - kt_lo, 1);

  const int nwg = gridDim.x, w = blockIdx.x;
  const TileWalk tw = tile_walk(p.flags, ntiles, nwg, w);    // tiles tw.first, tw.first + tw.step, ... (XCD-owned runs: kernels/gemm_common.h)
  const int my_tiles = tw.count;
  const int total_it = my_tiles * nk;

  // ---- loader state (waves 4-7): load l < AI belongs to the partner wave - 4, l >= AI to the wave itself ----
  // dense: a_st = byte offset of the row (or SENT).  im2col: a_st = tap-validity mask << 23 | (pixel of tap 0 + shift);
  // the byte offset is pixel * 2C + chunk, the chunk term only depends on the parity of the load index.
  unsigned a_st[LA], b_off[LB];
  const int lc16_0 = (pc ^ ((0 + (lrow >> 1)) & 7)) * 16, lc16_1 = (pc ^ ((4 + (lrow >> 1)) & 7)) * 16;
  const int cshift = CONV ? ((p.kt >> 1) * p.Hi + p.pad_t) * p.Wi + p.pad_l : 0;
  const __amdgpu_buffer_rsrc_t rA0 = __builtin_amdgcn_make_buffer_rsrc((void*)(A0 - (long)cshift * p.C0), 0, (int)SENT, 0x00020000);
  const __amdgpu_buffer_rsrc_t rA1 =
      __builtin_amdgcn_make_buffer_rsrc((void*)((CONV && p.A1) ? p.A1 - (long)cshift * p.C1 : A0), 0, (int)SENT, 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, (int)SENT, 0x00020000);

  auto setup_tile = [&](int tile) {
    int tm, tn; tile_coord_p(p, tile, ntm, ntn, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
#pragma unroll
    for (int l = 0; l < LA; ++l) {
      const int rg = (wave - 4 + 4 * (l / AI)) * AI + (l % AI);   // 8-row group of the A tile
      const int m = m0 + rg * 8 + lrow;
      const bool ok = m < p.M && kt_lo < kt_hi;
      if (CONV) {
        const int hw = p.Ho * p.Wo;
        const int t = m / hw, rem = m - t * hw;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const int y0 = oy * p.stride - p.pad_t, x0 = ox * p.stride - p.pad_l;
        unsigned mk = 0; int bit = 0;
        for (int it = 0; it < p.kt; ++it)
          for (int iy = 0; iy < p.ky; ++iy)
            for (int ix = 0; ix < p.kx; ++ix, ++bit) {
              const int tt = t + it - (p.kt >> 1), y = y0 + iy, x = x0 + ix;
              if (tt >= 0 && tt < p.T && y >= 0 && y < p.Hi && x >= 0 && x < p.Wi) mk |= 1u << bit;
            }
        const unsigned ap = (unsigned)(((t - (p.kt >> 1)) * p.Hi + y0) * p.Wi + x0 + cshift);
        a_st[l] = ok ? (mk << 23) | ap : 0u;
      } else {
        a_st[l] = ok ? (unsigned)m * (unsigned)(p.C0 * 2) + (((l % AI) & 1) ? lc16_1 : lc16_0) : SENT;
      }
    }
#pragma unroll
    for (int l = 0; l < LB; ++l) {
      const int rg = (wave - 4 + 4 * (l / BI)) * BI + (l % BI);
      const int lr = rg * 8 + lrow;                     // LDS row of the W tile; holds W row n0 + perm(lr)
      const int part = lr / WTN, rem = lr % WTN;
      const int jj = rem >> 4, i = rem & 15;
      const int n = n0 + part * WTN + (i >> 2) * WID + jj * 4 + (i & 3);
      b_off[l] = (n < p.N && kt_lo < kt_hi) ? (unsigned)n * (unsigned)(p.ldw * 2) + ((rg & 1) ? lc16_1 : lc16_0) : SENT;
    }
  };

  int ld_ti = 0, ld_ks = 0, ld_slot = 0;
  int u_it = 0, u_iy = 0, u_ix = 0, u_cb = 0;
  auto issue = [&]() {
    if (ld_ks == 0) {
      setup_tile(tw.first + ld_ti * tw.step);
      if (CONV) {
        const int kt0 = kt_lo * BK;
        // chunk-major K: step q = kt0 / BK is (channel chunk q / taps, tap q % taps)
        const int q = kt0 / BK, ntap = p.kt * p.ky * p.kx;
        const int chk = q / ntap;
        const int tap = q - chk * ntap;
        u_cb = chk * BK;
        u_it = tap / (p.ky * p.kx);
        const int r2 = tap - u_it * (p.ky * p.kx);
        u_iy = r2 / p.kx; u_ix = r2 - u_iy * p.kx;
      }
    }
    f16* As = smem + ld_slot * STAGE;
    f16* Bs = As + BM * BK;
    const int kt0 = (kt_lo + ld_ks) * BK;
    if (CONV) {
      const int tapbit = 23 + (u_it * p.ky + u_iy) * p.kx + u_ix;
      const int tappix = (u_it * p.Hi + u_iy) * p.Wi + u_ix;
      const bool src0 = u_cb < p.C0;
      const int Cs2 = (src0 ? p.C0 : p.C1) * 2;
      const int soff = tappix * Cs2 + (src0 ? u_cb : u_cb - p.C0) * 2;
      const __amdgpu_buffer_rsrc_t rs = src0 ? rA0 : rA1;
#pragma unroll
      for (int l = 0; l < LA; ++l) {
        const int rg = (wave - 4 + 4 * (l / AI)) * AI + (l % AI);
        const unsigned off = (a_st[l] & 0x7FFFFFu) * (unsigned)Cs2 + (((l % AI) & 1) ? lc16_1 : lc16_0);
        const unsigned voff = ((a_st[l] >> tapbit) & 1u) ? off : SENT;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(As + rg * 8 * BK), 16, (int)voff, soff, 0, 0);
      }
      // chunk-major K order (the only one the single-tap-per-K-tile paths take): next tap of the same 64-channel chunk, then the next chunk
      if (++u_ix == p.kx) { u_ix = 0; if (++u_iy == p.ky) { u_iy = 0; if (++u_it == p.kt) { u_it = 0; u_cb += BK; } } }
    } else {
#pragma unroll
      for (int l = 0; l < LA; ++l) {
        const int rg = (wave - 4 + 4 * (l / AI)) * AI + (l % AI);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA0, (lptr_t)(As + rg * 8 * BK), 16, (int)a_st[l], kt0 * 2, 0, 0);
      }
    }
#pragma unroll
    for (int l = 0; l < LB; ++l) {
      const int rg = (wave - 4 + 4 * (l / BI)) * BI + (l % BI);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(Bs + rg * 8 * BK), 16, (int)b_off[l], kt0 * 2, 0, 0);
    }
    if (++ld_ks == nk) { ld_ks = 0; ++ld_ti; }
    if (++ld_slot == NST) ld_slot = 0;
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, g = lane >> 4;
  const int sw = swz<BK>(l15);

  if (loader) {
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0)
      if (s0 < total_it) issue();
  }
  // 1: tile_epilogue_lean, 2: tile_epilogue_geglu_lean (64-column wave tiles), 0: tile_epilogue with all its run-time variants (kernels/gemm_common.h; knob 8388608 = always 0)
  // (the 256 x 256 tiles sit at the 256-register limit: only the GEGLU form - what the clip uses them for - and only in the dense kernels)
  const int lean_epi = (ST || (p.tune_knobs & 8388608)) ? 0 : (BM * BN < 65536 && tile_epilogue_lean_ok(p, WTN)) ? 1 : (NT == 4 && !CONV && tile_epilogue_geglu_lean_ok(p, WTN)) ? 2 : 0;
  bool drain = false;
  int cp_ti = 0, cp_ks = 0, cp_slot = 0;
#ifdef UG_GEMM_TRACE
  const bool traced = p.trace && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (wave == 0 || wave == 4);
  unsigned* tr_lds = (unsigned*)(smem + NST * STAGE) + (wave == 4 ? 24 * 5 : 0);
#endif
  for (int fi = 0; fi < total_it; ++fi) {
    UG_STAMP(0);
    const int younger = min(NST - 2, total_it - 1 - fi);
    if (drain || younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LA + LB) : "memory");
    drain = false;
    UG_STAMP(1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    UG_STAMP(2);
    if (loader && fi + NST - 1 < total_it) issue();
    UG_STAMP(3);
    const f16* Ab = smem + cp_slot * STAGE + (wm * WTM + l15) * BK;
    const f16* Bb = smem + cp_slot * STAGE + BM * BK + (wn * WTN + l15) * BK;
    if (++cp_slot == NST) cp_slot = 0;
    {
      f16x8 af[2][MT], bf[2][NT];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int ch = ((kk * 4 + g) ^ sw) * 8;
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[kk][j] = *(const f16x8*)(Bb + j * 16 * BK + ch);
#pragma unroll
        for (int i = 0; i < MT; ++i) af[kk][i] = *(const f16x8*)(Ab + i * 16 * BK + ch);
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
#ifdef UG_GEMM_TRACE
    if (traced && fi == 40) {
      __builtin_amdgcn_s_waitcnt(0);
      for (int i = lane; i < 24 * 5; i += 64) ((unsigned*)p.trace)[(wave == 4 ? 24 * 5 : 0) + i] = tr_lds[i];
    }
#endif
    if (++cp_ks != nk) continue;
    cp_ks = 0;
    const int tile = tw.first + (cp_ti++) * tw.step;
    drain = true;
    { int etm, etn; tile_coord_p(p, tile, ntm, ntn, etm, etn);
      if constexpr (ST) tile_epilogue_stats<MT, NT, WTM, WTN>(p, acc, etm * BM, etn * BN, wm, wn, lane, etm * WMW + wm);
      else if constexpr (BM * BN < 65536) {
        if (lean_epi == 1) tile_epilogue_lean<MT, NT, WTM, WTN, false>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off, nullptr);
        else if constexpr (NT == 4 && !CONV) { if (lean_epi == 2) tile_epilogue_geglu_lean<MT, NT, WTM, WTN>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off); else tile_epilogue<MT, NT, WTM, WTN>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off); }
        else tile_epilogue<MT, NT, WTM, WTN>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off);
      } else if constexpr (NT == 4 && !CONV) { if (lean_epi == 2) tile_epilogue_geglu_lean<MT, NT, WTM, WTN>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off); else tile_epilogue<MT, NT, WTM, WTN>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off); }
      else tile_epilogue<MT, NT, WTM, WTN>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off); }
  }
}

// buffer-addressing preconditions (see gemm_kernel BUFA); packed = the im2col state additionally fits mask << 23 | pixel
static bool gemm_can_bufa(const GemmP& p, int BK, bool packed) {
  const long lim = (1L << 31) - 64;
  if ((long)p.N * p.ldw * 2 >= lim) return false;
  if (p.conv) {
    const bool kc_ok = p.kchunk || p.kt * p.ky * p.kx == 1 || (p.C0 + p.C1) == BK;
    const bool uni = kc_ok && ((p.C0 + p.C1) % BK) == 0 && (p.C0 % BK) == 0 && p.kt * p.ky * p.kx <= (packed ? 9 : 32);
    const long px = (long)p.T * p.Hi * p.Wi + ((long)(p.kt >> 1) * p.Hi + p.pad_t) * p.Wi + p.pad_l + 1;
    return uni && p.ups == 1 && px * p.C0 * 2 < lim && px * p.C1 * 2 < lim && (!packed || px < (1L << 23));
  }
  return p.K % BK == 0 && (long)p.M * p.C0 * 2 < lim;
}

// ------------------------------------------------------------------------------------------------------------
// Producer / consumer kernel (256x128x64 tile, three-slot ring, buffer addressing only): 12 waves per workgroup, three
// per SIMD.  Waves 0-7 only compute (LDS fragment reads + MFMAs + the tile epilogue), waves 8-11 - one per SIMD - only
// fetch (12 direct-to-LDS loads per K-step each).
//
// Why: the K-step traces (profiles/r01_gemm_kstep_trace.txt) show address-path time and matrix time adding up whenever
// a wave does both jobs - a wave queued on the address path cannot issue its MFMAs - and every re-arrangement of the two
// jobs inside the same waves (spread loads, 8-phase groups, 4-slot software pipeline) stayed within +-5 %.  The
// microbenchmark tools/microbench/mfma_vs_dma.hip shows the hardware itself has no such coupling: an MFMA stream keeps its
// 16.3 cycles per instruction while the SIMD partner issues LDS-DMA loads back to back.  So the fetch gets its own waves.
// The accumulator of a 256x128 tile is 64 registers per consumer lane, which leaves room for a third wave per SIMD
// (<= 168 VGPRs each); the fetch waves need ~30.
//
// Synchronisation: one s_barrier per K-step for all 12 waves.  In step i the consumers read slot i % 3; the producers
// issue the fetch of step i+2 into slot (i+2) % 3 (the consumers left it before the previous barrier) and wait until only
// those 12 loads are in flight (the fetch of step i+1 has landed) before they arrive: the barrier that ends step i both
// publishes step i+1 and frees slot i % 3.  Only producers ever have loads in flight in the K loop, so the epilogue's
// global traffic (consumers) needs no drain logic.  LDS image, W row permutation, MFMA chain order and epilogue are those
// of gemm_kernel: outputs are bit-identical.
// ring depth of the producer / consumer kernel: three slots; -DUG_WS_NST4 (experiment, round 6): four on the 192 x 128 tile (4 x 40 KiB = the CU's 160 KiB)
#ifdef UG_WS_NST4
constexpr int ws_nst(int bm, int bn) { return (bm == 192 && bn == 128) ? 4 : 3; }
#else
constexpr int ws_nst(int, int) { return 3; }
#endif
template <int BM, int BN, int WMW, int WNW, bool CONV, bool ST = false>   // ST: GroupNorm statistics of the output from the epilogue (GemmP::stat_part)
__global__ __launch_bounds__((WMW * WNW + 4) * 64, 3) void gemm_ws_kernel(const GemmP p) {
  constexpr int NST = ws_nst(BM, BN);
  constexpr int BK = 64;
  static_assert(BN % 32 == 0 && (BM == 256 || BM == 192) && NST * (BM + BN) * BK * 2 <= 160 * 1024, "ring of three BM x BN x 64 slots must fit 160 KiB; BM / 32 A pieces per fetch wave");
  static_assert(WMW * WNW == 8, "eight consumer waves (three waves per SIMD with the fetch wave)");
  constexpr int NCW = WMW * WNW;
  constexpr unsigned SENT = 0x80000000u;
  constexpr int WTM = BM / WMW, WTN = BN / WNW;
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int WID = 4 * NT;
  constexpr int LA = BM / 32, LB = BN / 32;          // 1 KiB loads per producer wave per K-step (BM/8 A + BN/8 B pieces / 4 waves)
  constexpr int STAGE = (BM + BN) * BK;
  extern __shared__ __attribute__((aligned(16))) f16 smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= NCW;
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int ntiles = ntm * ntn;

  const int bz = blockIdx.z;
  const int bo = bz / p.nb_inner, bi = bz - bo * p.nb_inner;
  const f16* A0 = p.A0 + bo * p.sA_o + bi * p.sA_i;
  const f16* Wb = p.W + bo * p.sW_o + bi * p.sW_i;
  const long out_off = bo * p.sO_o + bi * p.sO_i;

  const int nk_all = (p.K + BK - 1) / BK;
  int kt_lo = 0, kt_hi = nk_all;
  if (p.splitk > 1) {
    const int per = (nk_all + p.splitk - 1) / p.splitk;
    kt_lo = min(nk_all, (int)blockIdx.y * per);
    kt_hi = min(nk_all, kt_lo + per);
  }
  const int nk = max(kt_hi - kt_lo, 1);

  const int nwg = gridDim.x, w = blockIdx.x;
  const TileWalk tw = tile_walk(p.flags, ntiles, nwg, w);    // tiles tw.first, tw.first + tw.step, ... (XCD-owned runs: kernels/gemm_common.h)
  const int my_tiles = tw.count;
  const int total_it = my_tiles * nk;

  if (producer) {
    // ================================= fetch waves =================================
    const int pw = wave - NCW;                        // 0..3
    const int pc = lane & 7, lrow = lane >> 3;
    const int Cin = p.C0 + p.C1;
    const int lc16_0 = (pc ^ ((0 + (lrow >> 1)) & 7)) * 16, lc16_1 = (pc ^ ((4 + (lrow >> 1)) & 7)) * 16;
    const int cshift = CONV ? ((p.kt >> 1) * p.Hi + p.pad_t) * p.Wi + p.pad_l : 0;
    const __amdgpu_buffer_rsrc_t rA0 = __builtin_amdgcn_make_buffer_rsrc((void*)(A0 - (long)cshift * p.C0), 0, (int)SENT, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA1 =
        __builtin_amdgcn_make_buffer_rsrc((void*)((CONV && p.A1) ? p.A1 - (long)cshift * p.C1 : A0), 0, (int)SENT, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, (int)SENT, 0x00020000);
    // dense: a_st = byte offset of the row (or SENT); im2col: tap-validity mask << 23 | (pixel of tap 0 + shift)
    unsigned a_st[LA], b_off[LB];
    int ld_ti = 0, ld_ks = 0, ld_slot = 0;
    int u_it = 0, u_iy = 0, u_ix = 0, u_cb = 0;
    auto issue = [&]() {
      if (ld_ks == 0) {
        const int tile = tw.first + ld_ti * tw.step;
        int tm, tn; tile_coord_p(p, tile, ntm, ntn, tm, tn);
        const int m0 = tm * BM, n0 = tn * BN;
#pragma unroll
        for (int l = 0; l < LA; ++l) {
          const int rg = pw * LA + l;                   // 8-row group of the A tile
          const int m = m0 + rg * 8 + lrow;
          const bool ok = m < p.M && kt_lo < kt_hi;
          if (CONV) {
            const int hw = p.Ho * p.Wo;
            const int mg = m + p.m_off;                   // row of the whole convolution (row-split launches)
            const int t = mg / hw, rem = mg - t * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            const int y0 = oy * p.stride - p.pad_t, x0 = ox * p.stride - p.pad_l;
            unsigned mk = 0; int bit = 0;
            for (int it = 0; it < p.kt; ++it)
              for (int iy = 0; iy < p.ky; ++iy)
                for (int ix = 0; ix < p.kx; ++ix, ++bit) {
                  const int tt = t + it - (p.kt >> 1), y = y0 + iy, x = x0 + ix;
                  if (tt >= 0 && tt < p.T && y >= 0 && y < p.Hi && x >= 0 && x < p.Wi) mk |= 1u << bit;
                }
            const unsigned ap = (unsigned)(((t - (p.kt >> 1)) * p.Hi + y0) * p.Wi + x0 + cshift);
            a_st[l] = ok ? (mk << 23) | ap : 0u;
          } else {
            a_st[l] = ok ? (unsigned)m * (unsigned)(p.C0 * 2) + ((l & 1) ? lc16_1 : lc16_0) : SENT;
          }
        }
#pragma unroll
        for (int l = 0; l < LB; ++l) {
          const int rg = pw * LB + l;
          const int lr = rg * 8 + lrow;                 // LDS row of the W tile; holds W row n0 + perm(lr)
          const int part = lr / WTN, rem = lr % WTN;
          const int jj = rem >> 4, i = rem & 15;
          const int n = n0 + part * WTN + (i >> 2) * WID + jj * 4 + (i & 3);
          b_off[l] = (n < p.N && kt_lo < kt_hi) ? (unsigned)n * (unsigned)(p.ldw * 2) + ((rg & 1) ? lc16_1 : lc16_0) : SENT;
        }
        if (CONV) {
          const int kt0 = kt_lo * BK;
          // chunk-major K: step q = kt0 / BK is (channel chunk q / taps, tap q % taps)
          const int q = kt0 / BK, ntap = p.kt * p.ky * p.kx;
          const int chk = q / ntap;
          const int tap = q - chk * ntap;
          u_cb = chk * BK;
          u_it = tap / (p.ky * p.kx);
          const int r2 = tap - u_it * (p.ky * p.kx);
          u_iy = r2 / p.kx; u_ix = r2 - u_iy * p.kx;
        }
      }
      f16* As = smem + ld_slot * STAGE;
      f16* Bs = As + BM * BK;
      const int kt2 = (kt_lo + ld_ks) * BK * 2;
      if (CONV) {
        const int tapbit = 23 + (u_it * p.ky + u_iy) * p.kx + u_ix;
        const int tappix = (u_it * p.Hi + u_iy) * p.Wi + u_ix;
        const bool src0 = u_cb < p.C0;
        const int Cs2 = (src0 ? p.C0 : p.C1) * 2;
        const int soff = tappix * Cs2 + (src0 ? u_cb : u_cb - p.C0) * 2;
        const __amdgpu_buffer_rsrc_t rs = src0 ? rA0 : rA1;
#pragma unroll
        for (int l = 0; l < LA; ++l) {
          const unsigned off = (a_st[l] & 0x7FFFFFu) * (unsigned)Cs2 + ((l & 1) ? lc16_1 : lc16_0);
          unsigned voff = ((a_st[l] >> tapbit) & 1u) ? off : SENT;
          if ((p.tune_knobs & 262144) && tapbit != 23) voff = SENT;   // TIMING-ONLY ablation (wrong results): activation rows fetched for ONE tap of nine - what a halo-resident tile would pull through the L2
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(As + (pw * LA + l) * 8 * BK), 16, (int)voff, soff, 0, 0);
        }
        // chunk-major K order (the only one the single-tap-per-K-tile paths take): next tap of the same 64-channel chunk, then the next chunk
        if (++u_ix == p.kx) { u_ix = 0; if (++u_iy == p.ky) { u_iy = 0; if (++u_it == p.kt) { u_it = 0; u_cb += BK; } } }
      } else {
#pragma unroll
        for (int l = 0; l < LA; ++l)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rA0, (lptr_t)(As + (pw * LA + l) * 8 * BK), 16, (int)a_st[l], kt2, 0, 0);
      }
#pragma unroll
      for (int l = 0; l < LB; ++l)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(Bs + (pw * LB + l) * 8 * BK), 16, (int)b_off[l], kt2, 0, 0);
      if (++ld_ks == nk) { ld_ks = 0; ++ld_ti; }
      if (++ld_slot == NST) ld_slot = 0;
    };
    // prologue: steps 0 .. NST-2; step 0 must have landed before the first barrier
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0) if (s0 < total_it) issue();
    if (total_it >= NST - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (LA + LB)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef UG_GEMM_TRACE
    const bool traced = p.trace && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && wave == NCW;
    unsigned* tr_lds = (unsigned*)(smem + NST * STAGE) + 2 * 24 * 5;
#endif
    for (int fi = 0; fi < total_it; ++fi) {
      UG_STAMP(0);
      if (fi + NST - 1 < total_it) {
        issue();                                                            // step fi+NST-1 -> the slot the consumers left before the previous barrier
        UG_STAMP(1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (LA + LB)) : "memory");       // step fi+1 has landed (NST-2 younger steps may be in flight)
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      UG_STAMP(2);
      __builtin_amdgcn_s_barrier();
      UG_STAMP(3);
#ifdef UG_GEMM_TRACE
      if (traced && fi == 40) {
        __builtin_amdgcn_s_waitcnt(0);
        for (int i = lane; i < 24 * 5; i += 64) ((unsigned*)p.trace)[2 * 24 * 5 + i] = tr_lds[i];
      }
#endif
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
  int cp_ti = 0, cp_ks = 0, cp_slot = 0;
  // epilogue operands are fetched while K steps are still to come (EpiPre, kernels/gemm_common.h): column operands at the tile's first step, row
  // operands three steps before its last.  Knob 2097152 = off (A/B); split-K launches write raw partial sums and take no operands.
  // Only the 192-row tiles have the registers (155 / 165 of the 168 three waves per SIMD leave; the 256-row tiles would spill 100 - 350 bytes per lane).
  constexpr bool PREK = (BM == 192);
  EpiPre<MT, NT> pre;
  const bool do_pre = PREK && p.splitk <= 1 && !(p.tune_knobs & 2097152);
  const bool lean = !ST && tile_epilogue_lean_ok(p, WTN) && !(p.tune_knobs & 8388608);   // the epilogue without its run-time variants (kernels/gemm_common.h); knob 8388608 = off (A/B)
  const bool lean_geglu = !ST && NT == 4 && tile_epilogue_geglu_lean_ok(p, WTN) && !(p.tune_knobs & 8388608);
  const int prow_ks = nk > 3 ? nk - 3 : 0;
  __builtin_amdgcn_s_barrier();          // step 0 published
  asm volatile("" ::: "memory");
#ifdef UG_GEMM_TRACE
  const bool traced = p.trace && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (wave == 0 || wave == 4);
  unsigned* tr_lds = (unsigned*)(smem + NST * STAGE) + (wave == 4 ? 24 * 5 : 0);
#endif
  for (int fi = 0; fi < total_it; ++fi) {
    UG_STAMP(0);
    if (PREK && do_pre && (cp_ks == 0 || cp_ks == prow_ks)) {
      int ptm, ptn; tile_coord_p(p, tw.first + cp_ti * tw.step, ntm, ntn, ptm, ptn);
      if (cp_ks == 0) epi_pre_cols<MT, NT, WTM, WTN>(p, ptn * BN, wn, lane, pre);
      if (cp_ks == prow_ks) epi_pre_rows<MT, NT, WTM, WTN>(p, ptm * BM, ptn * BN, wm, wn, lane, pre);
    }
    const f16* Ab = smem + cp_slot * STAGE + (wm * WTM + l15) * BK;
    const f16* Bb = smem + cp_slot * STAGE + BM * BK + (wn * WTN + l15) * BK;
    if (++cp_slot == NST) cp_slot = 0;
    {
      // (BN = 160: 80 accumulator + 72 fragment registers do not fit the 168 three waves per SIMD leave - 20 - 28 bytes of scratch in the K loop; one 32-deep slice
      // of fragments at a time there, round 5)
      constexpr bool SLICE = MT * NT * 4 + 2 * (MT + NT) * 4 > 150;
      if constexpr (SLICE) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int ch = ((kk * 4 + g) ^ sw) * 8;
          f16x8 a1[MT], b1[NT];
#pragma unroll
          for (int j = 0; j < NT; ++j) b1[j] = *(const f16x8*)(Bb + j * 16 * BK + ch);
#pragma unroll
          for (int i = 0; i < MT; ++i) a1[i] = *(const f16x8*)(Ab + i * 16 * BK + ch);
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1[j], a1[i], acc[i][j], 0, 0, 0);
        }
      } else {
      f16x8 af[2][MT], bf[2][NT];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int ch = ((kk * 4 + g) ^ sw) * 8;
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[kk][j] = *(const f16x8*)(Bb + j * 16 * BK + ch);
#pragma unroll
        for (int i = 0; i < MT; ++i) af[kk][i] = *(const f16x8*)(Ab + i * 16 * BK + ch);
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
    }
    UG_STAMP(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // all fragment reads of this slot are done before it is handed back
    UG_STAMP(2);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    UG_STAMP(3);
    // The epilogue of a tile runs AFTER the barrier that ends its last K step (round 6): the barrier hands the slot back, so the fetch waves issue the next
    // step's loads and wait for them WHILE the compute waves are in the epilogue - before, they sat at this barrier for the whole epilogue and the fetch
    // pipeline (what paces the K loop) stood still for it.  The epilogue touches no LDS, and the accumulators are not needed before the next step's MFMAs.
    if (++cp_ks == nk) {
      cp_ks = 0;
      const int tile = tw.first + (cp_ti++) * tw.step;
      { int etm, etn; tile_coord_p(p, tile, ntm, ntn, etm, etn);
        if constexpr (ST) tile_epilogue_stats<MT, NT, WTM, WTN>(p, acc, etm * BM, etn * BN, wm, wn, lane, etm * WMW + wm);
        else if (PREK && do_pre) {      // 192-row tiles: epilogue operands prefetched (EpiPre)
          if (lean) tile_epilogue_lean<MT, NT, WTM, WTN, PREK>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off, &pre);
          else if constexpr (NT == 4) { if (lean_geglu) tile_epilogue_geglu_lean<MT, NT, WTM, WTN>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off);
            else tile_epilogue<MT, NT, WTM, WTN, PREK>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off, &pre); }
          else tile_epilogue<MT, NT, WTM, WTN, PREK>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off, &pre);
        } else if constexpr (PREK) {    // (split-K partial sums, knob 2097152)
          tile_epilogue<MT, NT, WTM, WTN>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off);
        } else {
          if (lean) tile_epilogue_lean<MT, NT, WTM, WTN, false>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off, nullptr);
          else if constexpr (NT == 4) { if (lean_geglu) tile_epilogue_geglu_lean<MT, NT, WTM, WTN>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off);
            else tile_epilogue<MT, NT, WTM, WTN>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off); }
          else tile_epilogue<MT, NT, WTM, WTN>(p, acc, etm * BM, etn * BN, wm, wn, lane, out_off);
        } }
    }
#ifdef UG_GEMM_TRACE
    if (traced && fi == 40) {
      __builtin_amdgcn_s_waitcnt(0);
      for (int i = lane; i < 24 * 5; i += 64) ((unsigned*)p.trace)[(wave == 4 ? 24 * 5 : 0) + i] = tr_lds[i];
    }
#endif
  }
}

template <int BN, int WMW, int WNW, int BM = 256>
static void launch_ws(const GemmP& p, int batch, hipStream_t s) {
  const int ntiles = cdiv(p.M, BM) * cdiv(p.N, BN);
#ifdef UG_GEMM_TRACE
  const size_t lds = ws_nst(BM, BN) * (BM + BN) * 64 * sizeof(f16) + 2048;
#else
  const size_t lds = ws_nst(BM, BN) * (BM + BN) * 64 * sizeof(f16);
#endif
  static bool attr[32] = {};
  bool& at = attr[ug_dev_slot()];
  if (!at) {
    UG_CHECK(hipFuncSetAttribute((const void*)gemm_ws_kernel<BM, BN, WMW, WNW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    UG_CHECK(hipFuncSetAttribute((const void*)gemm_ws_kernel<BM, BN, WMW, WNW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    at = true;
  }
  const int split = p.splitk > 1 ? p.splitk : 1;
  int gx = std::max(8, 256 / (split * batch));
  gx = (gx / 8) * 8;
  gx = std::min(gx, ntiles);
  dim3 grid(gx, split, batch);
  if (p.stat_part) {   // launch_gemm: 128-column convolutions only
    if constexpr (BN == 128) {
      UG_REQUIRE(p.conv, "producer / consumer kernel: epilogue statistics for convolutions only");
      static bool attrs[32] = {};
      bool& ats = attrs[ug_dev_slot()];
      if (!ats) { UG_CHECK(hipFuncSetAttribute((const void*)gemm_ws_kernel<BM, BN, WMW, WNW, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); ats = true; }
      hipLaunchKernelGGL((gemm_ws_kernel<BM, BN, WMW, WNW, true, true>), grid, dim3((WMW * WNW + 4) * 64), lds, s, p);
      return;
    } else UG_REQUIRE(false, "producer / consumer kernel: epilogue statistics on 128-column tiles only");
  }
  if (p.conv) hipLaunchKernelGGL((gemm_ws_kernel<BM, BN, WMW, WNW, true>), grid, dim3((WMW * WNW + 4) * 64), lds, s, p);
  else hipLaunchKernelGGL((gemm_ws_kernel<BM, BN, WMW, WNW, false>), grid, dim3((WMW * WNW + 4) * 64), lds, s, p);
}

template <int BM, int BN, int NST, int WMW, int WNW>
static void launch_ldr(const GemmP& p, int batch, hipStream_t s) {
  UG_REQUIRE(p.m_off == 0, "row-split launches need the producer / consumer kernel");
  const int ntiles = cdiv(p.M, BM) * cdiv(p.N, BN);
#ifdef UG_GEMM_TRACE
  const size_t lds = (size_t)NST * (BM + BN) * 64 * sizeof(f16) + 1024;
#else
  const size_t lds = (size_t)NST * (BM + BN) * 64 * sizeof(f16);
#endif
  static bool attr[32] = {};
  bool& at = attr[ug_dev_slot()];
  if (!at) {
    UG_CHECK(hipFuncSetAttribute((const void*)gemm_ldr_kernel<BM, BN, NST, WMW, WNW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    UG_CHECK(hipFuncSetAttribute((const void*)gemm_ldr_kernel<BM, BN, NST, WMW, WNW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    at = true;
  }
  const int per_cu = (160 * 1024) / (int)lds < 1 ? 1 : std::min(2, (160 * 1024) / (int)lds);
  const int split = p.splitk > 1 ? p.splitk : 1;
  int gx = std::max(8, (per_cu * 256) / (split * batch));
  gx = (gx / 8) * 8;
  gx = std::min(gx, ntiles);
  dim3 grid(gx, split, batch);
  if (p.stat_part) {   // launch_gemm: the 256 x 256 im2col tile only (config 35: the wide temporal convolutions of the VAE decoder)
    if constexpr (BM == 256 && BN == 256) {
      UG_REQUIRE(p.conv, "loader kernel: epilogue statistics for convolutions only");
      static bool attrs[32] = {};
      bool& ats = attrs[ug_dev_slot()];
      if (!ats) { UG_CHECK(hipFuncSetAttribute((const void*)gemm_ldr_kernel<BM, BN, NST, WMW, WNW, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); ats = true; }
      hipLaunchKernelGGL((gemm_ldr_kernel<BM, BN, NST, WMW, WNW, true, true>), grid, dim3(512), lds, s, p);
      return;
    } else UG_REQUIRE(false, "loader kernel: epilogue statistics on the 256 x 256 tile only");
  }
  if (p.conv) hipLaunchKernelGGL((gemm_ldr_kernel<BM, BN, NST, WMW, WNW, true>), grid, dim3(512), lds, s, p);
  else hipLaunchKernelGGL((gemm_ldr_kernel<BM, BN, NST, WMW, WNW, false>), grid, dim3(512), lds, s, p);
}

// Sum the split-K partials in split order (deterministic) and apply the epilogue; 8 columns per thread.
__global__ __launch_bounds__(256) void splitk_epilogue(const GemmP p) {
  const long nvec = (long)p.M * (p.N / 8);
  const int npr = p.N / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < nvec; idx += (long)gridDim.x * blockDim.x) {
    const long m = idx / npr; const int n = (int)(idx - m * npr) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.splitk; ++s) {
      const float* P = p.partial + ((long)s * p.M + m) * p.N + n;
      const f32x4 a = *(const f32x4*)P, b = *(const f32x4*)(P + 4);
      v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3]; v[4] += b[0]; v[5] += b[1]; v[6] += b[2]; v[7] += b[3];
    }
    if (p.bias) { const f16x8 b = *(const f16x8*)(p.bias + n);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] += (float)b[q]; }
    if (p.bias2) { const f16x8 b = *(const f16x8*)(p.bias2 + n);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] += (float)b[q]; }
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] *= p.c0;
    if (p.R1) { const f16x8 r = *(const f16x8*)(p.R1 + m * p.ldr1 + n);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] += p.c1 * (float)r[q]; }
    if (p.R2) { const f16x8 r = *(const f16x8*)(p.R2 + m * p.ldr2 + n);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] += p.c2 * (float)r[q]; }
    if (p.act == UG_ACT_SILU) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = silu_f(v[q]);
    } else if (p.act == UG_ACT_GELU) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = gelu_f(v[q]);
    }
    f16x8 h;
#pragma unroll
    for (int q = 0; q < 8; ++q) h[q] = (f16)v[q];
    *(f16x8*)((f16*)p.Out + m * p.ldo + n) = h;
  }
}

template <int BM, int BN, int BK, int NST, int WMW, int WNW, bool CONV, bool UNI, bool BUFA = false, bool ST = false>
static void launch_t(const GemmP& p, int batch, hipStream_t s) {
  const int ntiles = cdiv(p.M, BM) * cdiv(p.N, BN);
#ifdef UG_GEMM_TRACE
  const size_t lds = (size_t)NST * (BM + BN) * BK * sizeof(f16) + 1024;
#else
  const size_t lds = (size_t)NST * (BM + BN) * BK * sizeof(f16);
#endif
  auto kern = gemm_kernel<BM, BN, BK, NST, WMW, WNW, CONV, UNI, BUFA, false, ST>;
  static bool attr[32] = {};
  bool& at = attr[ug_dev_slot()];
  if (!at) {
    if (lds > 64 * 1024) UG_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    at = true;
  }
  const int per_cu = GemmOcc<BM, BN, BK, NST, WMW, WNW>::wg;
  const int split = p.splitk > 1 ? p.splitk : 1;
  // persistent grid: at most (co-resident workgroups per CU) x 256 CUs, shared with the other grid dimensions
  const int cap = std::max(8, (per_cu * 256) / (split * batch));
  int gx = (cap / 8) * 8;                  // a multiple of 8 keeps the XCD-aware tile walk
  gx = std::min(gx, ntiles);
  if (gx < ntiles && cap >= ntiles) gx = ntiles;   // ... unless rounding down would push a few tiles into a second round (100 tiles, 5 K slices: 102 -> 96)
  if (p.tune_knobs & 1) gx = ntiles;
  dim3 grid(gx, split, batch);
  hipLaunchKernelGGL(kern, grid, dim3(WMW * WNW * 64), lds, s, p);
}

template <int BM, int BN, int BK, int NST, int WMW, int WNW>
static void launch_mode(const GemmP& p, int batch, hipStream_t s) {
  UG_REQUIRE(p.m_off == 0, "row-split launches need the producer / consumer kernel");
  const long lim = (1L << 31) - 64;   // buffer addressing: every byte offset must stay below num_records
  const bool bufw = (long)p.N * p.ldw * 2 < lim && !(p.tune_knobs & 4);
  if (p.conv) {
    // the single-tap-per-K-tile paths walk K chunk-major: the weights must be laid out that way (trivially true for 1 tap / 1 chunk)
    const bool kc_ok = p.kchunk || p.kt * p.ky * p.kx == 1 || (p.C0 + p.C1) == BK;
    const bool uni = kc_ok && ((p.C0 + p.C1) % BK) == 0 && (p.C0 % BK) == 0 && p.kt * p.ky * p.kx <= 32;
    const long px = (long)p.T * p.Hi * p.Wi + ((long)(p.kt >> 1) * p.Hi + p.pad_t) * p.Wi + p.pad_l + 1;
    const bool bufa = bufw && uni && p.ups == 1 && px * p.C0 * 2 < lim && px * p.C1 * 2 < lim;
    if (p.stat_part) {   // launch_gemm: only for the symmetric-kernel tiles instantiated with the statistics epilogue (configs 14 and 19, buffer-addressed im2col)
      if constexpr ((BM == 256 && BN == 64 && NST == 2 && WMW == 4) || (BM == 256 && BN == 128 && NST == 3 && WMW == 2)) { UG_REQUIRE(uni && bufa, "statistics epilogue: buffer-addressed im2col only"); launch_t<BM, BN, BK, NST, WMW, WNW, true, true, true, true>(p, batch, s); return; }
      else UG_REQUIRE(false, "statistics epilogue: tile not instantiated");
    }
    if (uni && bufa) launch_t<BM, BN, BK, NST, WMW, WNW, true, true, true>(p, batch, s);
    else if (uni) launch_t<BM, BN, BK, NST, WMW, WNW, true, true>(p, batch, s);
    else launch_t<BM, BN, BK, NST, WMW, WNW, true, false>(p, batch, s);
  } else {
    // dense: measured +4-6 % on the 8-wave tiles, -1..-4 % on the 4-wave 128x64 / 256x64 ones (profiles/r01_gemm_buffer_addressing.txt)
    const bool bufa = bufw && p.K % BK == 0 && (long)p.M * p.C0 * 2 < lim && (BM * BN >= 256 * 128 || (p.tune_knobs & 8));
    if (bufa) launch_t<BM, BN, BK, NST, WMW, WNW, false, false, true>(p, batch, s);
    else launch_t<BM, BN, BK, NST, WMW, WNW, false, false>(p, batch, s);
  }
}

// Tile configurations.  id -> (BM, BN, BK, stages, waves M x N, LDS).  Ids are stable (tools/tune_gemm.py, profiles/);
// the variants that lost everywhere in profiles/r01_gemm_tile_tuning.txt (BK=32 rings, 3-4 stage rings at lower
// occupancy, 64x128 / 4-wave 128x64 wave tiles) were removed from the build.
#ifdef UG_GEMM_TRACE
static constexpr bool ring3 = true;      // the trace dump needs 2 KiB of LDS next to the ring
#else
static constexpr bool ring3 = false;
#endif
static void launch_cfg(int cfg, const GemmP& p, int batch, hipStream_t s) {
  switch (cfg) {
    case 0: launch_mode<128, 128, 64, 2, 2, 2>(p, batch, s); break;   //  64 KiB, 2 WG/CU
    case 1: launch_mode<128, 64, 64, 2, 2, 2>(p, batch, s); break;    //  48 KiB, 3 WG/CU
    case 3: launch_mode<128, 64, 64, 3, 2, 2>(p, batch, s); break;    //  72 KiB, 2 WG/CU
    case 4: launch_mode<256, 128, 64, 3, 4, 2>(p, batch, s); break;   // 144 KiB, 1 WG/CU (8 waves)
    case 8: launch_mode<256, 128, 64, 2, 4, 2>(p, batch, s); break;   //  96 KiB, 1 WG/CU (8 waves)
    case 12: launch_mode<64, 64, 64, 2, 2, 2>(p, batch, s); break;    //  32 KiB, 5 WG/CU
    case 14: launch_mode<256, 64, 64, 2, 4, 2>(p, batch, s); break;   //  80 KiB, 2 WG/CU (8 waves)
    case 15: launch_mode<256, 256, 64, 2, 2, 4>(p, batch, s); break;  // 128 KiB, 8 waves, wave tile 128x64
    case 19: launch_mode<256, 128, 64, 3, 2, 4>(p, batch, s); break;  // 144 KiB, 8 waves, wave tile 128x32
    // asymmetric-loader forms of the 8-wave tiles (gemm_ldr_kernel); need buffer addressing, else the symmetric kernel
    case 35: if (gemm_can_bufa(p, 64, true)) launch_ldr<256, 256, 2, 2, 4>(p, batch, s); else launch_mode<256, 256, 64, 2, 2, 4>(p, batch, s); break;
    case 39: if (gemm_can_bufa(p, 64, true)) launch_ldr<256, 128, 3, 2, 4>(p, batch, s); else launch_mode<256, 128, 64, 3, 2, 4>(p, batch, s); break;
    // producer / consumer forms of the 256x128 tile (gemm_ws_kernel): wave tile 128x32 (59) and 64x64 (54, GEGLU-capable)
    case 59: if (gemm_can_bufa(p, 64, true)) launch_ws<128, 2, 4>(p, batch, s); else launch_mode<256, 128, 64, 3, 2, 4>(p, batch, s); break;
    case 54: if (gemm_can_bufa(p, 64, true)) launch_ws<128, 4, 2>(p, batch, s); else launch_mode<256, 128, 64, 3, 4, 2>(p, batch, s); break;
    case 60: if (gemm_can_bufa(p, 64, true)) launch_ws<160, 4, 2>(p, batch, s); else launch_mode<256, 64, 64, 2, 4, 2>(p, batch, s); break;   // 256x160: every UNet width is a multiple of 160
    case 34: if (gemm_can_bufa(p, 64, true)) launch_ldr<256, 64, 2, 4, 2>(p, batch, s); else launch_mode<256, 64, 64, 2, 4, 2>(p, batch, s); break;
    // 192-row tiles: M = 19200 / 4800 (levels 1 / 2 of the clip) are 100 / 25 tiles of 192 rows - with 128 columns the launch is 500 / 250
    // tiles for 256 CUs (fill 0.98) where 256 rows give 375 / 190 (fill 0.73)
    case 61: launch_mode<192, 128, 64, 3, 2, 4>(p, batch, s); break;  // 120 KiB, 8 waves, wave tile 96x32
    case 62: launch_mode<192, 128, 64, 3, 4, 2>(p, batch, s); break;  // 120 KiB, 8 waves, wave tile 48x64 (GEGLU-capable)
    case 63: if (gemm_can_bufa(p, 64, true)) launch_ws<128, 2, 4, 192>(p, batch, s); else launch_mode<192, 128, 64, 3, 2, 4>(p, batch, s); break;   // producer / consumer form of 61
    case 64: if (gemm_can_bufa(p, 64, true)) launch_ws<128, 4, 2, 192>(p, batch, s); else launch_mode<192, 128, 64, 3, 4, 2>(p, batch, s); break;   // ... of 62
    // halo-staged 3x3 convolutions (kernels/conv_halo.hip): the activation halo of a 64-channel chunk is fetched once for its nine taps
    case 70: UG_REQUIRE(conv_halo_supported(p, batch, 256, 160), "config 70: not a halo-stageable convolution"); launch_conv_halo(p, 256, 160, s); break;
    case 71: UG_REQUIRE(conv_halo_supported(p, batch, 256, 128), "config 71: not a halo-stageable convolution"); launch_conv_halo(p, 256, 128, s); break;
    case 72: UG_REQUIRE(conv_halo_supported(p, batch, 192, 128), "config 72: not a halo-stageable convolution"); launch_conv_halo(p, 192, 128, s); break;
    case 73: UG_REQUIRE(conv_halo_supported(p, batch, 192, 160), "config 73: not a halo-stageable convolution"); launch_conv_halo(p, 192, 160, s); break;
    case 80: launch_gemm_stream(p, s); break;   // weight-stationary streaming GEMM (kernels/gemm_stream.hip)
    default: UG_REQUIRE(false, "unknown / pruned GEMM tile config");
  }
}

// tuning overrides travel with the problem (GemmP::tune_*; the engine copies its context's - ug_tune_force): knobs 1 = one tile per workgroup, 2 = no XCD remap, 4 = flat addressing only
// ---- MX-fp8 dense GEMM (BASELINE configs[4]): the same persistent kernel on e4m3 operands with e8m0 block scales ----
template <int BM, int BN, int NST, int WMW, int WNW, bool BUFA>
static void launch_mx_t(const GemmP& p, hipStream_t s) {
  const int ntiles = cdiv(p.M, BM) * cdiv(p.N, BN);
  const size_t lds = (size_t)NST * (BM + BN) * 64 * sizeof(f16) + (size_t)NST * (BM + BN) * 4;
  auto kern = gemm_kernel<BM, BN, 64, NST, WMW, WNW, false, false, BUFA, true>;
  static bool attr[32] = {};
  bool& at = attr[ug_dev_slot()];
  if (!at) { UG_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); at = true; }
  const int per_cu = std::max(1, std::min(2, (int)((160 * 1024) / lds)));
  int gx = (per_cu * 256 / 8) * 8;
  gx = std::min(gx, ntiles);
  hipLaunchKernelGGL(kern, dim3(gx, 1, 1), dim3(WMW * WNW * 64), lds, s, p);
}
template <int BM, int BN, int NST, int WMW, int WNW>
static void launch_mx(const GemmP& p, hipStream_t s) {
  const long lim = (1L << 31) - 64;
  const bool bufa = (long)p.N * p.ldw * 2 < lim && (long)p.M * p.C0 * 2 < lim;
  if (bufa) launch_mx_t<BM, BN, NST, WMW, WNW, true>(p, s);
  else launch_mx_t<BM, BN, NST, WMW, WNW, false>(p, s);
}
void launch_gemm_mx8(const GemmP& p0, hipStream_t s) {
  GemmP p = p0;
  UG_REQUIRE(!p.conv && p.K % 128 == 0 && p.C0 % 16 == 0 && p.ldw % 16 == 0, "MX-fp8 GEMM: dense, K % 128 == 0, 16-byte aligned rows");
  UG_REQUIRE(p.sa && p.sw && p.zero, "MX-fp8 GEMM needs block scales");
  UG_REQUIRE(p.M > 0 && p.N > 0 && p.nb_inner >= 1 && p.splitk <= 1, "MX-fp8 GEMM shape");
  if (p.flags & UG_F_GEGLU) UG_REQUIRE(p.N % 128 == 0, "GEGLU GEMM needs N % 128 == 0");
  p.K /= 2; p.C0 /= 2; p.ldw /= 2;          // bytes -> the loaders' fp16 units (see gemm_kernel<MX>)
  p.splitk = 1; p.cfg_p1 = 0;
  if (p.tune_knobs & 2) p.flags |= UG_F_NOXCD;
  if (p.tune_knobs & 128) p.flags |= UG_F_XCDROUND;
  const int force = (p.tune_cfg_p1 - 1);
  // largest tile that still gives the 256 CUs ~one workgroup each (profiles/r02_mx8_per_shape.txt: 4800x1280x5120 on 256x256 tiles = 95
  // workgroups ran below the fp16 kernel)
  const long t256 = (long)cdiv(p.M, 256) * cdiv(p.N, 256), t128n = (long)cdiv(p.M, 256) * cdiv(p.N, 128);
  int pick = t256 >= 200 ? 0 : t128n >= 200 ? 1 : 2;
  {   // 192-row tile (pick 3) when it fills the last round of the persistent grid markedly better than the pick above (M = 19200 / 4800: 500 /
      // 250 workgroups instead of 375 / 190 - the fp16 planner's kCands entry 63 / 64, same arithmetic)
    auto eff = [&](int bm, int bn, int slots, float base) {
      const long tm = cdiv(p.M, bm), tn = cdiv(p.N, bn), tiles = tm * tn;
      return base * ((float)p.M / (tm * bm)) * ((float)p.N / (tn * bn)) * ((float)tiles / (cdiv(tiles, (long)slots) * slots));
    };
    const float cur = pick == 0 ? eff(256, 256, 256, 1.0f) : pick == 1 ? eff(256, 128, 256, 0.95f) : eff(128, 128, 512, 0.78f);
    if (eff(192, 128, 256, 0.88f) > 1.04f * cur) pick = 3;
  }
  if (force >= 100) pick = force - 100;
  {   // grouped tile walk when the fp8 weights exceed the L2 (same rule as pick_group_m)
    const int bm = pick == 2 ? 128 : pick == 3 ? 192 : 256, bn = pick == 0 ? 256 : 128, percu = pick == 2 ? 2 : 1;
    const int ntm = cdiv(p.M, bm), ntn = cdiv(p.N, bn);
    int g = 1;
    if (!(p.tune_knobs & 32) && (double)p.N * p.K * 2.0 > 3.0 * (1 << 20) && ntm >= 2 && ntn >= 2)
      while (g * 2 <= ntm && (double)(g * 2) * (g * 2) <= 32.0 * percu * bn / bm * 1.5) g *= 2;
    p.group_m = g;
  }
  switch (pick) {
    case 0: launch_mx<256, 256, 2, 2, 4>(p, s); break;
    case 1: launch_mx<256, 128, 3, 4, 2>(p, s); break;
    case 3: launch_mx<192, 128, 3, 4, 2>(p, s); break;
    default: launch_mx<128, 128, 2, 2, 2>(p, s); break;
  }
  UG_CHECK(hipGetLastError());
}





// Tile planner.  For every candidate tile the attainable rate is modelled as
//     base(cfg) x (M, N edge-tile utilisation) x (fill of the last round of the persistent grid)
// with base = the rate measured on a large, exactly tiling problem (tools/tune_gemm.py, profiles/r01_gemm_tile_tuning.txt:
// 8192^3 dense, the 512-channel VAE conv for im2col).  Checked against the full sweep of the clip's shapes the model's
// pick is the measured best or within ~5 % of it.  Low-resolution levels (M <= 2048) are latency-bound, not
// throughput-bound: they keep the measured rules below (64x64 tiles / split-K).
struct TileCand { int id, bm, bn, percu; float dense, conv; bool geglu_ok; };   // 35 = loader form of 15; 59 / 54 = producer / consumer 256x128
static const TileCand kCands[] = {
    {15, 256, 256, 1, 1200.f, 1190.f, true},  {35, 256, 256, 1, 1190.f, 1200.f, true},  {59, 256, 128, 1, 1143.f, 1154.f, false},
    {54, 256, 128, 1, 1135.f, 1150.f, true},  {19, 256, 128, 1, 1044.f, 1002.f, false}, {0, 128, 128, 2, 890.f, 1025.f, true},
    {14, 256, 64, 2, 880.f, 883.f, false},    {1, 128, 64, 3, 757.f, 799.f, false},    {12, 64, 64, 5, 456.f, 652.f, false},
    // 192-row producer / consumer tiles (round 2, tools/tune_192.py): ~0.9 / 0.95 of the 256x128 form's rate per tile, but M = 19200 / 4800 /
    // 76800 are whole multiples of 192 and the launches fill the chip (500 / 250 tiles instead of 375 / 190)
    {63, 192, 128, 1, 1030.f, 1120.f, false}, {64, 192, 128, 1, 1020.f, 1110.f, true}};

void gemm_plan(const GemmP& p, int batch, int* cfg_out, int* split_out) {
  const bool geglu = p.flags & UG_F_GEGLU;
  int cfg = 15;
  float best = -1.f;
  for (const TileCand& c : kCands) {
    if (geglu && !c.geglu_ok) continue;
    const long tm = cdiv(p.M, c.bm), tn = cdiv(p.N, c.bn);
    const long tiles = tm * tn * batch, slots = (long)c.percu * 256;
    // knob 4194304 (round 5): this context shares the GPU with another clip in flight - the CUs a thin last round leaves idle are not lost, the other
    // stream's kernels run on them: score the tiles without the last-round fill factor
    const float fill = (p.tune_knobs & 4194304) ? 1.f : (float)tiles / (cdiv(tiles, slots) * slots);
    float sc = (p.conv ? c.conv : c.dense) * ((float)p.M / (tm * c.bm)) * ((float)p.N / (tn * c.bn)) * fill;
    if (c.id == 19 && p.conv && p.K <= 512) sc *= 1.25f;   // its 3-stage ring hides the short K loop's fill (temporal convs, K = 3C); in-situ it loses on dense K = 320
    // in situ the five-step K loops of level 0 (K = 320, operand fresh in the Infinity Cache) run ~8 % better on the smaller tiles than on
    // 256x256 (76800x960x320: 82 vs 90 us); the GEGLU projection (N = 2560) stays on 256x256 (184 vs 201 us)
    if ((c.id == 15 || c.id == 35) && !p.conv && !geglu && p.K <= 384) sc *= 0.9f;
    if (sc > best) { best = sc; cfg = c.id; }
  }
  // in-situ exception (tools/insitu_cfg_sweep.py): the level-0 down-projection (M = 76800, N = 320, K = 1280) re-reads its 197 MB
  // A operand once per 64-column tile; the 256x128 3-stage tile needs 3 instead of 5 passes (122 vs 131 us)
  if (!p.conv && !geglu && p.M >= 50000 && p.N > 256 && p.N <= 384 && p.K >= 1024) cfg = 4;
  // Round 3: in-situ exceptions from a re-run of tools/insitu_cfg_sweep.py with the full config set (profiles/r03_gemm_insitu_cfg_sweep.txt: every GEMM of
  // the clip forced to one tile at a time, operands wherever the producing kernel left them).  The cost model above is within 0 - 3 % of the in-situ
  // best on most shapes; these classes were 4 - 19 % off.  Knob 4096 = off (tools/ab_clip.py insitu).
  if (!(p.tune_knobs & 4096) && batch == 1) {
    const bool bufa = gemm_can_bufa(p, 64, true);
    if (!p.conv && geglu && p.M >= 4096 && p.N >= 4096 && p.K >= 512) cfg = bufa ? 35 : 15;              // 19200x5120x640: 129 vs 137 us; 4800x10240x1280: 113 vs 120
    else if (!p.conv && geglu && p.K <= 384 && p.M < 32768 && p.M >= 4096) cfg = (p.tune_knobs & 16777216) ? 64 : (bufa ? 35 : 15);   // feed-forward tail rows 11264x2560x320: round 3 192x128 (34.6 vs 41.2); round 6, with the lean GEGLU epilogue, the 256x256 tile: 32.7 vs 37.8 in situ (profiles/r06_gemm_insitu_cfg_sweep.txt; knob 16777216 = the old rule)
    else if (!p.conv && !geglu && cfg == 63 && p.K >= 2048 && p.N <= 1280) cfg = 64;                          // 19200x640x2560: 74 vs 79; 4800x1280x5120: 66 vs 69
    else if (!p.conv && !geglu && p.M >= 50000 && p.K <= 384 && p.N >= 640 && p.N < 2048 && bufa) cfg = 59;  // 76800x960x320 (Q|K|V): 77 vs 85
    else if (!p.conv && !geglu && p.N <= 384 && p.K >= 1024 && p.M > 2048 && p.M <= 16384) cfg = 3;          // tail rows 11264x320x1280: 23.8 vs 28.0
    else if (p.conv && p.kt > 1 && p.N > 256 && p.N <= 320 && p.M >= 50000 && !(p.tune_knobs & 524288)) cfg = 14;   // temporal conv 76800x320x960: 80 vs 90 (NOT the VAE decoder's 256- / 128-column ones: 393216x256x768 299 vs 228 on the cost model's tile; knob 524288 = the rule off, A/B)
  }
  int split = 1;
  const long tiles128 = (long)cdiv(p.M, 128) * cdiv(p.N, 128) * batch;
  const int nk = cdiv(p.K, 64);
  const bool plain_epi = !geglu && !p.up_phase && !(p.flags & UG_F_OUT_F32) && batch == 1 && (p.N % 8 == 0) && (p.ldo % 8 == 0) &&
                         (!p.R1 || p.ldr1 % 8 == 0) && (!p.R2 || p.ldr2 % 8 == 0);
  // Latency-bound corner (tools/tune_splitk.py): few tiles, long K loops.  Split-K over the 128x128 tile with the largest
  // factor whose persistent grid (512 / split workgroups) still holds every tile in one round;
  // mid-length K loops do better with two slices of the 3-stage 128x64 tile.
  if (!geglu && p.M <= 2048 && p.N < 2048) {
    cfg = 3;
    if (plain_epi && p.N >= 128) {
      if (nk >= 128) {
        for (int sp = 8; sp >= 2; --sp)
          if ((512 / sp) >= tiles128 && nk / sp >= 16) { split = sp; cfg = 0; break; }
      }
      if (split == 1 && nk >= 48) split = 2;
    }
    // Few-tile corner (a batch-1 StableNormal image: 81 / 324 / 1296 rows; tools/tune_splitk_small.py, profiles/r02_splitk_small_m.txt):
    // the launch has to be spread over the CUs along K.  im2col: 128x128 tiles, ~480 workgroups, >= 7 K steps per slice (conv1280@9x9
    // 62 -> 36 us, conv640@36x36 108 -> 58 us); dense: the 3-stage 128x64 tile, ~256 workgroups, >= 5 K steps per slice.
    if (plain_epi && p.N >= 128) {
      const long t64 = (long)cdiv(p.M, 128) * cdiv(p.N, 64);
      if (p.conv && tiles128 <= 64 && nk >= 64) {
        int sp = (int)std::min<long>(std::min<long>(512 / tiles128, nk / 7), 24);
        while (sp >= 2 && (512 / sp) < tiles128) --sp;     // the persistent grid must hold every tile in one round
        if (sp >= 2) { cfg = 0; split = sp; }
      } else if (!p.conv && t64 <= 64 && nk >= 16) {
        const int sp = (int)std::min<long>(256 / t64, nk / 5);
        if (sp >= 2) { cfg = 3; split = sp; }
      }
    }
    // Round 3 (tools/tune_splitk_small.py ... ws, profiles/r03_tune_sn_small.txt; knob 8192 = off): the batch-1 StableNormal shapes again, now with the
    // producer / consumer tiles in the sweep.  im2col with >= 2 row tiles: the 192x128 producer / consumer tile, K sliced until ~160 workgroups are in
    // flight (conv1280@18x18 37 -> 33 us, conv640@36x36 36 -> 32); a single row tile (M <= 128): 3-stage 128x64 tiles, ~240 workgroups
    // (conv1280@9x9 27 -> 24); dense mid-length K loops with few tiles: two slices (1296x640x2560 23 -> 18).
    if (!(p.tune_knobs & 8192) && plain_epi && p.N >= 128) {
      const long t192 = (long)cdiv(p.M, 192) * cdiv(p.N, 128), t64b = (long)cdiv(p.M, 128) * cdiv(p.N, 64);
      if (p.conv && p.M > 128 && t192 <= 128 && nk >= 16 && gemm_can_bufa(p, 64, true)) {
        cfg = 63; split = (int)std::max<long>(1, std::min<long>(std::min<long>(160 / t192, nk / 8), 8));
      } else if (p.conv && p.M <= 128 && nk >= 64) {
        const int sp = (int)std::min<long>(std::min<long>(240 / t64b, nk / 7), 24);
        if (sp >= 2) { cfg = 3; split = sp; }
      } else if (!p.conv && split == 1 && nk >= 32 && t64b <= 128 && t64b > 64) split = 2;
    }
    // Round 3 (tools/tune_level3.py, profiles/r03_tune_level3.txt): the long-K launches of the lowest level (M = 1200: 3x3 / temporal convs, the
    // 5120 -> 1280 projection) stream 10 - 60 MB of weights from HBM through 36 - 72 K steps per workgroup; the producer / consumer tiles
    // (dedicated fetch waves two K steps ahead in a 3-slot ring) hide that latency better than the symmetric 2-stage 128x128 tile:
    // conv 1280@6x8 62 -> 56 us, conv 2560@6x8 101 -> 87, tconv 37 -> 32, 1200x1280x5120 37 -> 33.  Knob 2048 = off.
    if (!(p.tune_knobs & 2048) && plain_epi && p.M > 1024 && p.N >= 1024 && p.N < 2048 && gemm_can_bufa(p, 64, true)) {
      if (p.conv && nk >= 128) { cfg = 59; split = 4; }
      else if (p.conv && nk >= 48) { cfg = 63; split = 3; }
      else if (!p.conv && nk >= 64) { cfg = 59; split = 3; }
    }
  } else if (!(p.tune_knobs & 512) && plain_epi && p.M <= 8192 && tiles128 <= 256 && p.N >= 128 && p.N < 2048 && (p.conv ? nk >= 32 : nk >= 16)) {
    // Under-filled launch (a 72x72 / 36x36 latent of ONE image: 40 - 250 tiles of 128x128 for 256 CUs): the cost model above prices a
    // round by its tile size only and keeps the 256x128 tile on 63 workgroups; what helps is workgroups - 128x128 (im2col) or 128x64
    // (dense) tiles, K sliced until ~512 are in flight (knob 512 = off; tools/ab_sn.py)
    const long t192u = (long)cdiv(p.M, 192) * cdiv(p.N, 128);
    if (p.conv && !(p.tune_knobs & 8192) && t192u <= 128 && gemm_can_bufa(p, 64, true)) {
      cfg = 63; split = (int)std::max<long>(1, std::min<long>(std::min<long>(160 / t192u, nk / 8), 8));     // conv320@72x72 36 -> 32 us (round 3)
    } else if (p.conv) {
      int sp = (int)std::min<long>(std::min<long>(512 / tiles128, nk / 8), 8);
      while (sp >= 2 && (512 / sp) < tiles128) --sp;
      cfg = 0; split = std::max(1, sp);
    } else {
      const long t64 = (long)cdiv(p.M, 128) * cdiv(p.N, 64);
      cfg = 3;
      const int sp = (int)std::min<long>(std::min<long>(512 / t64, nk / 8), 8);
      split = std::max(1, sp);
      if (!(p.tune_knobs & 8192) && t64 > 128) split = 1;     // round 3: with > 128 tiles the split costs more than it fills (5184x320x1280: 14.7 vs 22.0 us)
    }
  } else if (!geglu && !p.conv && p.M <= 512 && p.N >= 2048) {
    cfg = 3;   // wide projection of a few rows: many small tiles beat a handful of large ones (324x3840x1280: 14.4 vs 19.3 us)
  } else if (!(p.tune_knobs & 8192) && !geglu && !p.conv && batch == 1 && p.M <= 8192 && p.K <= 384 && p.N <= 384) {
    cfg = 3;   // round 3: short, narrow projections of one image (5184x320x320: 7.3 vs 9.6 us)
  }
  // (a former rule - four K slices of the 128x128 tile for the 12x16 level's concatenated 2560-channel convs, 880 TFLOP/s - is gone:
  // the 192-row tile fills the chip there without split-K, 236 vs 324 us = 1200 TFLOP/s)
  if ((p.tune_cfg_p1 - 1) >= 0 && !(geglu && (p.tune_cfg_p1 - 1) != 0 && (p.tune_cfg_p1 - 1) != 4 && (p.tune_cfg_p1 - 1) != 8 && (p.tune_cfg_p1 - 1) != 15 && (p.tune_cfg_p1 - 1) != 35 && (p.tune_cfg_p1 - 1) != 54 && (p.tune_cfg_p1 - 1) != 62 && (p.tune_cfg_p1 - 1) != 64)) cfg = (p.tune_cfg_p1 - 1);
  if ((p.tune_split_p1 - 1) >= 0) split = plain_epi ? std::max(1, (p.tune_split_p1 - 1)) : 1;
  *cfg_out = cfg; *split_out = split;
}

// Tile-walk grouping for dense layers whose weights do not fit the per-XCD L2 (tile_coord): the ~P tiles an XCD processes at a time
// should touch as few operand panels as possible: a x b = P with a * BM ~ b * BN.
static int pick_group_m(const GemmP& p, int cfg, int batch, int split) {
  if ((p.conv && (p.tune_knobs & 64)) || batch > 1 || (p.tune_knobs & 32)) return 1;   // knobs: 32 = row-major walk everywhere, 64 = row-major for im2col (A/B)
  if ((double)p.N * p.K * 2.0 <= 3.0 * (1 << 20)) return 1;          // weights stay L2 resident: share the activation panel instead
  int bm = 256, bn = 128, percu = 1;
  switch (cfg) {
    case 0: bm = 128; bn = 128; percu = 2; break;
    case 1: case 3: bm = 128; bn = 64; percu = cfg == 1 ? 3 : 2; break;
    case 12: bm = 64; bn = 64; percu = 5; break;
    case 14: case 34: bm = 256; bn = 64; percu = 2; break;
    case 15: case 35: bm = 256; bn = 256; break;
    case 60: case 70: bm = 256; bn = 160; break;
    case 61: case 62: case 63: case 64: case 72: bm = 192; bn = 128; break;
    case 73: bm = 192; bn = 160; break;
    default: break;                                                    // 4, 8, 19, 39, 54, 59, 71: 256 x 128
  }
  const int ntm = cdiv(p.M, bm), ntn = cdiv(p.N, bn);
  if (ntm < 2 || ntn < 2) return 1;
  const double P = 32.0 * percu;
  int g = 1;
  while (g * 2 <= ntm && (double)(g * 2) * (g * 2) <= P * bn / bm * 1.5) g *= 2;   // nearest power of two to sqrt(P bn / bm)
  // round 5: a group of g x ntn tiles that fits the XCD's window of P tiles covers whole tile rows anyway - the row-major walk touches the same panels and a
  // run boundary then splits ONE M tile's activation panel between two L2s instead of g (19200 x 640 x 2560, 5 column tiles: g = 4 -> 1).  Knob 1048576 = off.
  if (!(p.tune_knobs & 1048576) && (double)g * ntn <= P) g = 1;
  return g;
}

// Block / wave geometry of the tile configs whose epilogue is tile_epilogue (kernels/gemm_common.h) - ONE table for everything that depends on it: the
// GroupNorm statistics blocks (wave tile rows), the row-partial slots (wave tile columns) and the LayerNorm-fold / per-row bias2 checks.  The fall-back
// kernels of launch_cfg (no buffer addressing) have the geometry of the config they replace.  false: config 60 (20-column lane runs), the halo-staged and
// the streaming kernels (their own epilogues).
static bool cfg_geom(int cfg, int& bm, int& bn, int& wmw, int& wnw) {
  switch (cfg) {
    case 0: bm = 128; bn = 128; wmw = 2; wnw = 2; return true;
    case 1: case 3: bm = 128; bn = 64; wmw = 2; wnw = 2; return true;
    case 4: case 8: case 54: bm = 256; bn = 128; wmw = 4; wnw = 2; return true;
    case 12: bm = 64; bn = 64; wmw = 2; wnw = 2; return true;
    case 14: case 34: bm = 256; bn = 64; wmw = 4; wnw = 2; return true;
    case 15: case 35: bm = 256; bn = 256; wmw = 2; wnw = 4; return true;
    case 19: case 39: case 59: bm = 256; bn = 128; wmw = 2; wnw = 4; return true;
    case 61: case 63: bm = 192; bn = 128; wmw = 2; wnw = 4; return true;
    case 62: case 64: bm = 192; bn = 128; wmw = 4; wnw = 2; return true;
    default: return false;
  }
}

void launch_gemm(const GemmP& p0, int batch, hipStream_t s, int* stat_rb) {
  GemmP p = p0;
  if (stat_rb) *stat_rb = 0;
  if (!stat_rb) p.stat_part = nullptr;
  if (p.tune_knobs & 2) p.flags |= UG_F_NOXCD;
  if (p.tune_knobs & 128) p.flags |= UG_F_XCDROUND;
  if (p.tune_knobs & 16) p.flags |= UG_F_PRIO;
  UG_REQUIRE(p.K % 8 == 0, "GEMM K must be a multiple of 8");
  UG_REQUIRE(p.ldw % 8 == 0, "GEMM ldw must be a multiple of 8");
  UG_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "empty GEMM");
  UG_REQUIRE(p.zero != nullptr, "zero page missing");
  UG_REQUIRE(p.nb_inner >= 1, "nb_inner");
  if (p.flags & UG_F_GEGLU) UG_REQUIRE(p.N % 128 == 0, "GEGLU GEMM needs N % 128 == 0");
  if (p.conv) {
    UG_REQUIRE(p.C0 % 8 == 0 && p.C1 % 8 == 0, "conv channel counts must be multiples of 8");
    UG_REQUIRE(p.K == (p.C0 + p.C1) * p.kt * p.ky * p.kx, "conv K mismatch");
    UG_REQUIRE(p.M == p.T * p.Ho * p.Wo && p.m_off == 0, "conv M mismatch");
    UG_REQUIRE(p.ups == 1 || p.ups == 2, "ups");
  } else {
    UG_REQUIRE(p.C0 % 8 == 0, "dense lda must be a multiple of 8");
  }
  int cfg = p.cfg_p1 - 1, split = p.splitk;
  if (cfg < 0 || split < 1) { int c2, s2; gemm_plan(p, batch, &c2, &s2); if (cfg < 0) cfg = c2; if (split < 1) split = s2; }
  if (p.flags & UG_F_GEGLU) UG_REQUIRE(cfg == 0 || cfg == 4 || cfg == 8 || cfg == 15 || cfg == 35 || cfg == 54 || cfg == 62 || cfg == 64, "GEGLU needs a 64-column wave tile");
  p.splitk = split;
  // Row split for the 3x3 convolutions onto 320 columns (level 0 of the clip): 128-wide tiles waste a sixth of their columns there (320 =
  // 2.5 x 128) and the 256x160 producer / consumer tile (config 60), which wastes none, leaves 88 of its 600 tiles for a third round.
  // So: the rows of the WHOLE rounds on 256x160 tiles, the remaining rows as a second launch on whatever tile the planner picks for
  // them (tools: 76800 rows, 320 -> 320 channels: 127 + 37 us against 185 - 190 us for any single tiling).  Knob 1024 = off.
  bool planned = p0.cfg_p1 == 0;          // the caller took the planner's choice (run_gemm passes it back explicitly) - not a tuning override
  if (!planned && (p.tune_cfg_p1 - 1) < 0) { int c2, s2; gemm_plan(p0, batch, &c2, &s2); planned = (c2 == cfg && s2 == split); }
  const bool halo_on = planned && !(p.tune_knobs & 16384) && split == 1 && (p.tune_cfg_p1 - 1) < 0 && !(p.flags & UG_F_GEGLU);
  // (level 0, in situ - tools/profile_shapes.py: 320 -> 320 171 vs 189 us; the two-source convolutions of the up path 299 vs 297 and 432 vs 412: they keep the row split)
  const bool halo_l0 = halo_on && p.N % 160 == 0 && p.N <= 320 && p.C1 == 0 && !(p.tune_knobs & 32768) && conv_halo_supported(p, batch, 256, 128);
  if (!halo_l0 && planned && p.conv && p.kt == 1 && p.ky == 3 && p.kx == 3 && p.N % 160 == 0 && p.N <= 320 && split == 1 && batch == 1 && !p.up_phase && !(p.flags & (UG_F_OUT_F32 | UG_F_GEGLU)) &&
      (p.tune_cfg_p1 - 1) < 0 && !(p.tune_knobs & 1024) && gemm_can_bufa(p, 64, true)) {
    const long ntn = p.N / 160, tiles = (long)cdiv(p.M, 256) * ntn, whole = tiles / 256 * 256, rem = tiles - whole;
    if (whole > 0 && rem > 0 && rem * 2 <= 256) {
      const int M1 = (int)(whole / ntn) * 256;
      const bool halo = false;                                    // (the halo kernel takes these convolutions as ONE launch on 256 x 128 tiles instead - below; knob 32768 = this row split, A/B)
      GemmP a = p; a.M = M1; a.cfg_p1 = 61;                     // config 60 (+ 1)
      a.stat_part = nullptr;                                      // (two tile geometries for one tensor: no epilogue statistics)
      GemmP b = p0; b.stat_part = nullptr; b.M = p.M - M1; b.m_off = M1; b.Out = (void*)((f16*)p.Out + (long)M1 * p.ldo);
      b.flags |= p.flags & (UG_F_NOXCD | UG_F_XCDROUND);
      if (b.R1) b.R1 += (long)M1 * p.ldr1;
      if (b.R2) b.R2 += (long)M1 * p.ldr2;
      int cb, sb; gemm_plan(b, 1, &cb, &sb);
      if (cb != 59 && cb != 60 && cb != 63 && cb != 64 && cb != 54) cb = 63;
      b.cfg_p1 = cb + 1; b.splitk = 1;
      a.splitk = 1;
      a.group_m = pick_group_m(a, 60, 1, 1); a.tm_T = a.tm_nb = 0;
      launch_cfg(halo && conv_halo_supported(a, 1, 256, 160) ? 70 : 60, a, 1, s);
      if (halo && conv_halo_supported(b, 1, 256, 160)) cb = 70;
      b.group_m = pick_group_m(b, cb, 1, 1); b.tm_T = b.tm_nb = 0;
      launch_cfg(cb, b, 1, s);
      UG_CHECK(hipGetLastError());
      return;
    }
  }
  if (halo_on) {
    // Round 4: every halo-stageable 3x3 convolution goes to the halo kernel (kernels/conv_halo.hip) - of its tiles the one whose tile count fills
    // the 256 CUs best.  tools/ab_halo.py, profiles/r04_conv_halo.txt: 0.81 - 0.92 x the im2col time on levels 1 / 2 of the UNet, 0.72 - 0.95 x on the
    // VAE decoder.  Level 0 (320 columns; tools/ab_halo_l0.py): ONE launch on 256 x 128 tiles - three column tiles, a sixth of the third wasted -
    // runs the 320 -> 320 convolution in 145 us against 177 for the row-split im2col pair, 153 for 256 x 160 halo tiles in one launch and 183
    // for the row split on halo tiles (the thin second launch is what costs).  Bit-identical outputs.  Knob 16384 = halo off, 32768 = level 0 stays
    // on the row-split im2col path (A/B).
    int best = -1; double bfill = 0.0;
    for (int c = 70; c <= 73; ++c) {
      const int bm = c <= 71 ? 256 : 192, bn = (c == 70 || c == 73) ? 160 : 128;
      if (!conv_halo_supported(p, batch, bm, bn)) continue;
      const long tiles = (long)(p.M / bm) * cdiv(p.N, bn);
      const double useful = (double)p.N / (cdiv(p.N, bn) * bn);
      const double fill = (double)tiles / (cdiv(tiles, 256) * 256.0) * useful * (bm == 256 ? 1.0 : 0.97);
      if (fill > bfill) { bfill = fill; best = c; }
    }
    if (halo_l0) best = 71;
    else if (p.N % 160 == 0 && p.N <= 320) best = -1;             // knob 32768: the row split above took it, or it is not stageable
    if (best >= 0) cfg = best;
  }
  if (planned && !(p.tune_knobs & 65536) && (p.tune_cfg_p1 - 1) < 0 && split == 1 && gemm_stream_supported(p, batch)) {
    // Round 4: the short-K projections of the narrow level (K = 320 onto 320 / 960 columns) on the weight-stationary streaming kernel
    // (kernels/gemm_stream.hip); bit-identical.  Knob 65536 = off (A/B).
    cfg = 80;
  }
  if (split > 1) UG_REQUIRE(p.partial != nullptr, "split-K needs a partial buffer");
  p.group_m = pick_group_m(p, cfg, batch, split);
  p.tm_T = p.tm_nb = 0;
  if (p.conv && p.kt > 1 && p.T > 1 && batch == 1 && !(p.tune_knobs & 256)) {   // knob 256: frame-major M walk for temporal convs (A/B)
    int bm = 256;
    switch (cfg) { case 0: case 1: case 3: bm = 128; break; case 12: bm = 64; break; case 61: case 62: case 63: case 64: bm = 192; break; default: break; }
    const long hw = (long)p.Ho * p.Wo;
    if (hw % bm == 0 && hw / bm >= 1) { p.tm_T = p.T; p.tm_nb = (int)(hw / bm); }
  }
  if (p.stat_part) {
    // GroupNorm statistics of the output from the epilogue (GemmP::stat_part): one block per wave tile of WTM rows.  Only where every row of every tile
    // is stored through the epilogue's fp16 vector path and the blocks do not straddle frames; otherwise the caller runs the statistics pass.  Knob 131072 = off.
    int bm = 0, wmw = 0;
    {   // the kernels instantiated with the statistics epilogue: halo tiles of 128 columns; of the tile_epilogue family (geometry: cfg_geom) the buffer-addressed
        // im2col forms of 14 / 19 (symmetric kernel), 35 (loader kernel) and 54 / 59 / 63 / 64 (producer / consumer kernel)
      int gbm, gbn, gwm, gwn;
      const bool sym = (cfg == 14 || cfg == 19) && p.conv && gemm_can_bufa(p, 64, false) && !(p.tune_knobs & 4);
      const bool pk = (cfg == 35 || cfg == 54 || cfg == 59 || cfg == 63 || cfg == 64) && p.conv && gemm_can_bufa(p, 64, true);
      if (cfg == 71) { bm = 256; wmw = 4; }
      else if (cfg == 72) { bm = 192; wmw = 4; }
      else if ((sym || pk) && cfg_geom(cfg, gbm, gbn, gwm, gwn)) { bm = gbm; wmw = gwm; }
    }
    const int wtm = bm ? bm / wmw : 0;
    const bool ok = bm && split == 1 && batch == 1 && !(p.flags & (UG_F_GEGLU | UG_F_OUT_F32 | UG_F_R1_F32)) && !p.up_phase && !(p.tune_knobs & 131072) &&
                    p.M % bm == 0 && p.m_off == 0 && p.stat_hw > 0 && p.stat_hw % wtm == 0 && p.N % 16 == 0 && p.ldo % 8 == 0 &&
                    (!p.R1 || p.ldr1 % 8 == 0) && (!p.R2 || p.ldr2 % 8 == 0);
    static const bool sdbg = getenv("UG_STAT_DEBUG") != nullptr;
    if (sdbg) fprintf(stderr, "[stat] M %d N %d K %d conv %d kt %d hw %d cfg %d split %d -> rb %d\n", p.M, p.N, p.K, p.conv, p.kt, p.stat_hw, cfg, split, ok ? wtm : 0);
    if (ok) *stat_rb = wtm; else p.stat_part = nullptr;
  }
  launch_cfg(cfg, p, batch, s);
  if (split > 1) {
    const long nvec = (long)p.M * (p.N / 8);
    int grid = (int)std::min<long>((nvec + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(splitk_epilogue, dim3(grid), dim3(256), 0, s, p);
  }
  UG_CHECK(hipGetLastError());
}
