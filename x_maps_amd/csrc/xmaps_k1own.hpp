// xmaps_k1own.hpp -- K1 "owner tiles": the column tiles of xmaps_k1cols.hpp for rigs whose (row, time column) -> frame cell
// map is NOT injective -- the reference's own calibration: X_MAP_WIDTH = projector_width = 1080 time columns
// (x_maps_disparity.py:58-59) land on ~300 columns of the rectified frame (cam_proj_calibration.py:299-303), so up to four
// consecutive time columns of a row share one cell and the plain column tiles (one slot per (row, column) pair, one owner
// per cell by injectivity) do not apply.  (gfx950 / MI355X; included by xmaps_hip.hip after xmaps_k1cols.hpp)
//
//   * OWNER of a cell = the FIRST time column of its row that maps to it; delta(row, c) = c - owner column (0..7, found once
//     in xm_create and packed into the top 3 bits of a second copy of the X-map: xp | delta << 13).  Tile T = W time columns
//     [c0, c0 + W) owns the cells whose owner column is one of its own.  It reads the events of its columns AND of a halo of
//     `halo` >= max delta columns behind them: an event of column c belongs to the tile iff c0 <= c - delta < c0 + W.  Halo
//     events are read twice (by their own tile, which drops those whose cell the previous tile owns, and by that one).
//     Every cell has exactly one owner tile: last-writer-wins is resolved in the tile's LDS slots (ds_max on
//     (local index + 1) << 16 | disparity), the flush is a PLAIN 2-byte store, winners and empties alike -- the frame is
//     rewritten completely by every frame: no tag, no clear, no atomics.
//   * slots are indexed by the CELL, not by (row, column): on such rigs the X-map is strongly slanted (ESL: a time column's
//     cell moves 0.37 .. 0.43 columns per row), so a tile's cells form a thin diagonal band.  The u16 disparity frame is
//     SHEARED by whole columns per 8-row group -- cell (x, row) lives at column x + bias + ((row >> 3) * m >> 12), m fitted to
//     the rig's middle time column in xm_create -- which keeps the stores of 64 consecutive rows within a few frame columns; in
//     LDS every (tile, row) has its own first column `base`: the slot array is [nxs columns][rows the LUT can reach] with
//     nxs = the cells a row has in a tile (+ 1), the flush walks it in memory order (lanes = consecutive rows), a precomputed
//     bit mask per (tile, row) says which of the band's cells the tile owns.  K2 reads the same layout (its 16-byte loads take 8 aligned
//     rows of one column: the shear is constant there).
//   * cells outside the band ("extras": where the rectified time map replicates its border the X-map's arg-min jumps by hundreds
//     of columns -- first and last tile of the ESL rig, ~470 cells) get a slot of their own behind the band (index from a second
//     table, read by those events only) and are flushed one by one.
//   * sparse frames (ESL: 140 events per time column, 1320 rows): staging the LUT / X-map bands through LDS costs more than
//     it saves, both gathers go to L2 (tables of 1.2 + 2.9 MB, tiles in XCD-contiguous order), eight in flight per lane.
//   * exactness for any input, as for the column tiles: every event a tile loads is verified against the tile's thresholds;
//     a tile that objects fails the frame through the flag of the sorted-order verification -> automatic redo on the
//     64-bit general path.
// Algorithmic bytes are K1's: 24 B/event.
#pragma once
#include "xmaps_k1cols.hpp"

#ifndef XM_OWN_WAVES
#define XM_OWN_WAVES 1  /* waves per SIMD the 8-events-per-thread kernels are compiled for (experiments: 5, 6) */
#endif

namespace xm {

constexpr int OWN_BW = 4;         // tile widths are multiples of it (K0b finds two boundaries per tile: its first column and the end of its halo)
constexpr int OWN_MAX_DELTA = 7;  // 3 bits in the packed X-map
constexpr int OWN_XP_BITS = 13;   // xp < 8192
constexpr int OWN_MAX_NXS = 16;   // sheared frame columns per tile with per-row ownership (the masks are u16; 32 per 8-row group)
// a tile's band table (own_bm, own_setup): a u32 per row -- or, with ownership per 8-row group, two per group --, padded to 16 bytes
__host__ __device__ inline int own_tab_words(int hrp, bool grouped) { return ((grouped ? (hrp >> 3) * 2 : hrp) + 3) & ~3; }
constexpr int OWN_MAX_COLS = 72;  // own + halo columns of a tile (W <= 64, halo <= 8)
constexpr int OWN_MAX_ROW_PASSES = 4;         // a tile's rows go through its LDS slots in up to this many passes (own_plan)
constexpr size_t OWN_LDS_TARGET = 32 * 1024;  // ... as many as it takes to bring a block's LDS down to this

// EPT = events per thread and event pass: 8 (the column tiles' figure) or 4 -- half the registers per wave and twice the waves for
// the same tile: the kernel is bound by its own instruction issue at three to four waves per SIMD (profiles/r05_own_tiles.md)
template <bool AOS, bool VEC, int EPT>
__device__ __forceinline__ void scatter_own_body(gp_u16 xs, gp_u16 ys, gp_i64 ts, gp_u4 aos, const u32 n_ev, const DevTables& tb,
                                                 gp_state st, XM_GLOBAL uint16_t* frame16, const int W, const int halo,
                                                 const u32 blk, const u32 nblk, const int flags, const u32 frame_in_grid = 0) {
  const bool device_redo = flags & COLS_F_DEVICE_REDO, all_in = flags & COLS_F_ALL_IN_FRAME;
  typedef long long T;
  static_assert(!(AOS && VEC), "AoS records are loaded one per lane");
  static_assert(EPT == 8 || (EPT == 4 && !VEC), "16-byte loads of x / y take eight events");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ u32 s_in, s_oob;
  // the interior thresholds of the tile's columns, thr[c0 + j] at s_thr[3 + j] (j = 1, 5, 9, ... start a 16-byte quad), padded
  // with ~0 ("no event reaches it"): the column of an event = #{j >= 1 : thr[c0 + j] <= a}, four compares per broadcast read
  __shared__ __attribute__((aligned(16))) u32 s_thr[OWN_MAX_COLS + 8];
  const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63;
  const int n = (int)n_ev;
  const int cap = nthreads * EPT;
  const size_t cells16 = frame16_cells(tb);
  gp_i4 bounds = (gp_i4)((const XM_GLOBAL unsigned char*)frame16 + cols_bounds_offset(cells16));
  const XM_GLOBAL u32* thr = (const XM_GLOBAL u32*)((const XM_GLOBAL unsigned char*)frame16 + cols_thr_offset(cells16, tb.xmap_w));
  const int HRp = tb.own_hrp, r_lo = tb.own_r_lo;
  // LDS carve-up (mirrored by own_lds_bytes() on the host): band slots [nxs_max][RP] | extra slots [extra_max] | per row: first
  // frame column of the band, ownership mask.  RP = rows per pass: the tile's rows go through the band slots in ceil(HRp / RP)
  // passes (events and gathers stay in registers in between) -- a block's LDS is what limits the tiles a CU holds at once
  u32* slots = reinterpret_cast<u32*>(smem);
  const int RP = tb.own_rp;
  const int n_band_max = tb.own_nxs_max * RP;
  u32* slots_x = slots + n_band_max;
  u32* s_tab = slots_x + tb.own_extra_max;  // the tile's band table (own_tab_words): band positions and ownership per row / per 8-row group
  const bool grouped = tb.own_grouped != 0;
  const int ent_sh = grouped ? 3 : 0, n_ent = HRp >> ent_sh, tab_words = own_tab_words(HRp, grouped);

  XM_CSTAMP(0);
  const u32 tile = xcd_contiguous_in_frame(blk, nblk, frame_in_grid);
  const int c0 = (int)tile * W;
  const int Wc = min(W, tb.xmap_w - c0);
  const int c_end = min(c0 + W + halo, tb.xmap_w);
  const int ncols = c_end - c0;  // own + halo columns
  // ---- 1. what locates the tile, in one round trip of uniform loads
  const int4 b_lo = bounds[2u * tile], b_hi = bounds[min(2u * tile + 3u, 2u * nblk)];  // (k_cols_bounds with split = halo: boundary 2 t at t W, 2 t + 1 at t W + halo)
  const u32 A_lo = thr[c0], A_hi = thr[c_end];
  const int4 trec = ((const XM_GLOBAL int4*)tb.own_tiles)[tile];  // {band columns, first extra, extras, -}
  // the tile's band positions and ownership masks: 16-byte loads issued with the locating loads, parked in registers until
  // the slots are cleared (a copy loop of 2-byte loads here was a third of the block's life: one round trip per iteration)
  const XM_GLOBAL uint4* gbm = (const XM_GLOBAL uint4*)((const XM_GLOBAL u32*)tb.own_bm + (size_t)tile * (size_t)tab_words);
  const int n_bm4 = tab_words >> 2;
  uint4 bm_q[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) bm_q[q] = gbm[min(tid + q * nthreads, n_bm4 - 1)];
  T t_first, t_last;
  if constexpr (AOS) {
    const uint4 a = aos[0], b = aos[n - 1];
    t_first = (T)(((u64)a.w << 32) | a.z);
    t_last = (T)(((u64)b.w << 32) | b.z);
  } else {
    t_first = ts[0];
    t_last = ts[n - 1];
  }
  const u32 tag = st->tag_b + 1;
  XM_CSTAMP(1);  // (tag_b: the last of the tile's locating loads)
  const int nxs = trec.x, x_first = trec.y, n_extra = trec.z;
  const int nslots = nxs * RP;
  {  // winner slots; the tile's ownership masks (one u16 per row) and band positions (one per 8-row group)
    uint4* l_slots = reinterpret_cast<uint4*>(slots);
    for (int i = tid; i < (XM_CABL(9) ? 0 : (nslots >> 2)); i += nthreads) l_slots[i] = make_uint4(0, 0, 0, 0);  // (HRp % 8 == 0)
    for (int i = tid; i < n_extra; i += nthreads) slots_x[i] = 0;
    uint4* l_bm = reinterpret_cast<uint4*>(s_tab);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (tid + q * nthreads < n_bm4) l_bm[tid + q * nthreads] = bm_q[q];
    for (int i = tid + 2 * nthreads; i < n_bm4; i += nthreads) l_bm[i] = gbm[i];  // (more than 8 rows per thread: small blocks)
  }
  if (tid == 0) {
    s_in = 0;
    s_oob = 0;
  }
  if (tid < OWN_MAX_COLS + 5) s_thr[3 + tid] = tid < ncols ? thr[c0 + tid] : ~0u;
  int lb_s = b_lo.x, lb_e = b_hi.x;
  bool bad = false;
  if (lb_s < 0 || lb_e > n || lb_e < lb_s || (u32)(lb_e - lb_s) > COLS_MAX_TILE_EVENTS) {
    bad = true;
    lb_s = lb_e = 0;
  }
  const int a0 = VEC ? (lb_s & ~(EPT - 1)) : lb_s;
  int n_pass = 0;
  for (int left = lb_e > lb_s ? lb_e - a0 : 0; left > 0; left -= cap) n_pass += 1;
  if (XM_CABL(7)) n_pass = 0;  // (experiments: no per-event work at all)
  u32 xy[EPT];  // x | y << 16
  T tt[EPT];
  const auto load_events = [&](const int pass) {
    if (XM_CABL(6)) {  // (experiments: no event loads)
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        xy[k] = (u32)(tid * 7 + k * 13) % 600u | ((u32)(tid * 3 + k) % 400u) << 16;
        tt[k] = t_first + (T)A_lo + (T)((tid + k * 64) % 64);
      }
      return;
    }
    if constexpr (VEC) {
      const int base_true = a0 + pass * cap + tid * EPT;
      const int last_grp = (n - 1) & ~(EPT - 1);
      const int base = min(base_true, last_grp);
      const uint4 xv = *(gp_u4)(xs + base);
      const uint4 yv = *(gp_u4)(ys + base);
      const u32 xw[4] = {xv.x, xv.y, xv.z, xv.w}, yw[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
      for (int k = 0; k < EPT; ++k)
        xy[k] = (k & 1) ? (xw[k >> 1] >> 16) | (yw[k >> 1] & 0xffff0000u) : (xw[k >> 1] & 0xffffu) | (yw[k >> 1] << 16);
#pragma unroll
      for (int q = 0; q < EPT / 2; ++q) {
        const longlong2 a = *(const XM_GLOBAL longlong2*)(ts + (base + 2 * q < n ? base + 2 * q : base));
        tt[2 * q] = a.x;
        tt[2 * q + 1] = a.y;
      }
    } else {
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int i = lb_s + pass * cap + k * nthreads + tid;
        const int ic = i < lb_e ? i : lb_s;
        if constexpr (AOS) {
          const uint4 r = aos[ic];
          xy[k] = r.x;
          tt[k] = (T)(((u64)r.w << 32) | r.z);
        } else {
          xy[k] = (u32)xs[ic] | ((u32)ys[ic] << 16);
          tt[k] = ts[ic];
        }
      }
    }
  };
  const auto wave_on = [&](const int pass) {  // wave-uniform: does this wave hold any event of the pass?
    const int w0 = VEC ? a0 + pass * cap + (tid & ~63) * EPT : lb_s + pass * cap + (tid & ~63);
    return w0 < lb_e;
  };
  if (n_pass > 0 && wave_on(0)) load_events(0);

  // frame extrema = (t[0], t[n-1]), verified per event below; slot bookkeeping by block 0 (as k_scatter_cols)
  const u32 parity = tag & 1;
  if (t_last < t_first) t_last = t_first;
  if (blk == 0) {
    if (tid == 0) {
      st->tag_a = tag;
      st->mm[parity][0][0] = TimeCodec<T>::enc(t_first);
      st->mm[parity][0][1] = TimeCodec<T>::enc(t_last);
    }
    for (int i = tid; i < MM_SLOTS; i += nthreads) {
      st->mm[parity ^ 1][i][0] = MM_INIT_MIN;
      st->mm[parity ^ 1][i][1] = MM_INIT_MAX;
    }
  }
  __syncthreads();  // slots cleared, masks in place
  XM_CSTAMP(2);

  const u32* lut = tb.lut;
  const uint16_t* xmo_tile = tb.xmap_own + (size_t)c0 * (size_t)tb.xmap_h;  // the tile's first column of the packed X-map
  u32 n_in = 0, n_oob = 0;  // wave-uniform (counted with ballots)
  // per event: its slot (row pass << 24 | index inside the pass's slots; extras: pass 0, behind the band) or -1, and the value
  int code[EPT];
  u32 val[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    code[k] = -1;
    val[k] = 0;
  }
  const bool cached = n_pass <= 1;  // the tile's events fit the block's registers: loaded and looked up once for all row passes
  const int n_rp = (HRp + RP - 1) / RP;
  const u32 rp_inv = n_rp > 1 ? ((1u << 20) + (u32)RP - 1u) / (u32)RP : 0u;  // row / RP = row * rp_inv >> 20 (own_plan has checked it)
  const u32 A_span = A_hi - A_lo;
  // The column of an event inside the tile: #{j >= 1 : thr[c0 + j] <= a}.  Thresholds are (c - 1/2) span / S rounded -- evenly
  // spaced up to a unit --, so E(a) = floor((a - A_lo) ncols / A_span - 1/2) is the column or the one in front of it: one compare
  // against thr[E + 1] settles it.  The wave verifies that on the tile's thresholds themselves (E is monotone: it is enough that
  // E(thr[j]) >= j - 1 and E(thr[j] - 1) <= j - 1 for every j); a tile whose thresholds are not that regular (time stamps that
  // stand still, spans of a few units) counts compares instead.
  const float col_inv = (float)ncols / (float)max(A_span, 1u);
  const auto col_est = [&](const u32 rel) { return __float2uint_rz(fmaxf((float)rel * col_inv - 0.5f, 0.0f)); };
  bool col_lin;
  {
    const int j = lane + 1;  // (ncols <= OWN_MAX_COLS: two rounds of 64 lanes)
    bool ok = true;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int jq = j + 64 * q;
      if (jq <= ncols) {
        const u32 A = jq < ncols ? s_thr[3 + jq] : A_hi;
        if (A < A_lo || A > A_hi) ok = false;
        else {
          if (A > A_lo && col_est(A - 1u - A_lo) > (u32)(jq - 1)) ok = false;
          if (jq < ncols && col_est(A - A_lo) + 1u < (u32)jq) ok = false;
        }
      }
    }
    col_lin = !__any(!ok);
  }
  // The per-event section below keeps ~110 block-uniform values alive at once -- the scalar file holds 102, and what does not fit
  // is parked in lanes of spill VGPRs (v_writelane / v_readlane: 500 of the section's 2 200 instructions, a VALU slot each, in a
  // kernel bound by its instruction issue).  The values that are only ever OPERANDS of per-lane arithmetic are moved to vector
  // registers of their own here (the block has room: 102 of the 128 its occupancy allows) -- XM_OWN_VPIN = 0 keeps them scalar (A/B)
#ifndef XM_OWN_VPIN
#define XM_OWN_VPIN 1
#endif
  u32 v_cam_w = (u32)tb.cam_w, v_cam_h = (u32)tb.cam_h, v_xmap_h = (u32)tb.xmap_h, v_own_hr = (u32)tb.own_hr, v_rp_inv = rp_inv, v_A_lo = A_lo;
  int v_x_offset = tb.x_offset, v_r_lo = r_lo, v_RP = RP, v_Wc = Wc, v_W = W, v_nxs = nxs;
  u32 v_A_span = A_span, v_used_n = (u32)(lb_e - lb_s);
  float v_col_inv = col_inv;
#if XM_OWN_VPIN
  asm volatile("" : "+v"(v_cam_w), "+v"(v_cam_h), "+v"(v_xmap_h), "+v"(v_own_hr), "+v"(v_rp_inv), "+v"(v_A_lo));
  asm volatile("" : "+v"(v_x_offset), "+v"(v_r_lo), "+v"(v_RP), "+v"(v_Wc), "+v"(v_W), "+v"(v_nxs), "+v"(v_A_span), "+v"(v_used_n), "+v"(v_col_inv));
#endif
  const auto col_est_v = [&](const u32 rel) { return __float2uint_rz(fmaxf((float)rel * v_col_inv - 0.5f, 0.0f)); };
  for (int rp = 0; rp < n_rp; ++rp) {
  for (int pass = 0; pass < n_pass; ++pass) {
    const bool on = wave_on(pass);
    if (!on) continue;
    if (rp == 0 || !cached) {
    if (pass > 0 || rp > 0) load_events(pass);
    int e0;
    if constexpr (VEC) e0 = pass * cap + tid * EPT;
    else e0 = pass * cap + tid;
    const u32 used_n = v_used_n;
    const int u0 = VEC ? a0 + e0 - lb_s : e0;
    int tl[EPT];
    u32 rel[EPT];
    bool live[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const u64 a64 = (u64)(tt[k] - t_first);
      const u32 r = (u32)a64 - v_A_lo;
      const bool used = (u32)(u0 + (VEC ? k : k * nthreads)) < used_n;
      const bool in_tile = (u32)(a64 >> 32) == 0u && r < v_A_span;
      bad = bad || (used && !in_tile);
      live[k] = used && in_tile;
      rel[k] = live[k] ? r : 0u;
    }
    if (col_lin) {
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const u32 e = col_est_v(rel[k]);
        tl[k] = (int)e + (rel[k] + v_A_lo >= s_thr[3 + 1 + e] ? 1 : 0);
      }
    } else {
#pragma unroll
      for (int k = 0; k < EPT; ++k) tl[k] = 0;
      for (int j0 = 1; j0 < ncols; j0 += 4) {
        const uint4 A4 = *reinterpret_cast<const uint4*>(&s_thr[3 + j0]);
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
          const u32 av = rel[k] + A_lo;
          tl[k] += (av >= A4.x ? 1 : 0) + (av >= A4.y ? 1 : 0) + (av >= A4.z ? 1 : 0) + (av >= A4.w ? 1 : 0);
        }
      }
    }
    if (rp == 0 && pass == 0) XM_CSTAMP(3);  // events arrived, columns found
    // A1: the rectify LUT from L2, all of the thread's gathers in flight.  x / y outside the camera = map[y, x] IndexError in the
    // reference (calib:279-280): dropped and counted (once: by the tile whose own columns hold the event)
    u32 l[EPT];
    u32 oob_here = 0, in_here = 0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const u32 xk = xy[k] & 0xffffu, yk = xy[k] >> 16;
      const bool inside = xk < v_cam_w && yk < v_cam_h;
      oob_here += (u32)__popcll(__ballot(live[k] && !inside && tl[k] < v_Wc));
      live[k] = live[k] && inside;
      l[k] = XM_CABL(5) ? ((yk * 2u + 100u) << 16) | (xk + 50u) : lut[live[k] ? __umul24(xk, v_cam_h) + yk : 0u];
    }
    if (rp == 0 && pass == 0) XM_CSTAMP(4);  // LUT gathers issued
    // A2: the packed X-map (xp | delta << 13) from L2
    u32 xm[EPT];
    int rr[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int yr = (int)l[k] >> 16;
      rr[k] = yr - v_r_lo;
      live[k] = live[k] && (u32)rr[k] < v_own_hr;  // 0 <= yr < H - 1 (xmd:23): own_hr rows from r_lo on, all of them valid
      xm[k] = XM_CABL(5) ? (u32)(tb.x_offset + 300 + ((c0 + tl[k]) * 3 >> 2) + (yr >> 2)) : xmo_tile[live[k] ? __umul24((u32)tl[k], v_xmap_h) + (u32)yr : 0u];
    }
    const u32 val_base = (u32)(e0 + 1) << 16;
    // (two copies of the loop, one per kind of rig, instead of a branch per event: the kernel is bound by its own instruction issue)
    const auto finish = [&](auto all_in_c) {
      constexpr bool ALL_IN = decltype(all_in_c)::value;
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int xr = (int)(short)(l[k] & 0xffff);
        const int delta = (int)(xm[k] >> OWN_XP_BITS), fu = (int)(xm[k] & ((1u << OWN_XP_BITS) - 1u)) - v_x_offset;
        const int disp = fu - xr;          // (xm_create has checked the range: xmd:27's wrap never triggers)
        bool write = live[k] && disp >= 0;  // xmd:29; an undefined X-map cell is packed as 0: fu = -x_offset < xr_min <= xr
        int fc = fu;
        if constexpr (!ALL_IN) {
          if (fc < 0) fc += tb.rect_w;  // NumPy's negative wrap
          const bool in_frame = (u32)fc < (u32)tb.rect_w && rr[k] + r_lo < tb.rect_h;
          oob_here += (u32)__popcll(__ballot(write && !in_frame && tl[k] < v_Wc));
          write = write && in_frame;
        }
        in_here += (u32)__popcll(__ballot(write && tl[k] < v_Wc));  // counted by the tile whose own columns hold the event
        const int jo = tl[k] - delta;                               // the cell's owner column, relative to c0
        write = write && (u32)jo < (u32)v_W;
        // the cell's column inside its row's band: its frame column - the band's origin (both before the frame's shear: the
        // table holds the origin, the flush adds the row's shear)
        const int row = write ? rr[k] : 0;
        const int sx = fc - (int)(short)(s_tab[row >> ent_sh] & 0xffffu);
        const u32 rpo = __umul24((u32)row, v_rp_inv) >> 20;  // the row's pass
        const int idx = (int)((rpo << 24) | (u32)(__mul24(sx, v_RP) + row - (int)__umul24(rpo, (u32)v_RP)));
        // a cell outside the band (an "extra"): marked with its pair, looked up below in the tiles that have any
        const int extra = (int)(0x80000000u | ((u32)tl[k] << 16) | (u32)row);
        code[k] = write ? ((u32)sx < (u32)v_nxs ? idx : extra) : -1;
        val[k] = (val_base + ((u32)(VEC ? k : k * nthreads) << 16)) | (u32)disp;
      }
    };
    if (all_in) finish(std::true_type{});
    else finish(std::false_type{});
    if (n_extra > 0) {  // tile-uniform (ESL rig: the first and the last tile): the extras' slots come from the second table, at the event's pair
#pragma unroll
      for (int k = 0; k < EPT; ++k)
        if (code[k] < -1) {
          const u32 col = ((u32)code[k] >> 16) & 0x7fffu, row = (u32)code[k] & 0xffffu;
          const u32 e = ((const XM_GLOBAL uint16_t*)tb.xmap_extra)[__umul24((u32)c0 + col, (u32)tb.xmap_h) + row + (u32)r_lo];
          code[k] = e != 0u ? n_band_max + (int)e - 1 : -1;  // (always != 0: xm_create lists every owner cell outside its band)
        }
    }
    if (rp == 0 && pass == 0) XM_CSTAMP(5);  // both gathers arrived, slots worked out
    if (rp == 0) {  // (a tile of several event passes looks its events up once per row pass: counted the first time)
      n_in += in_here;
      n_oob += oob_here;
    }
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k)
      if (code[k] >= 0 && (code[k] >> 24) == rp) atomicMax(&slots[code[k] & 0xffffff], val[k]);
  }
  __syncthreads();
  if (rp == 0) XM_CSTAMP(6);  // pass 0's ds_max done, barrier passed
  // ---- flush of the pass's rows.  Ownership per 8-row group: a thread takes the eight rows of one group in one band column (the
  //      band's position and the frame's shear are constant there): two 16-byte LDS reads, ONE 16-byte store; threads =
  //      consecutive groups of one band column = consecutive 16 bytes of one frame column.  The slots are left cleared.
  const bool more = rp + 1 < n_rp;
  if (grouped) {
    const u32 col_stride = (u32)tb.rect_h;
    const int g_first = (rp * RP) >> 3, g_cnt = min(RP, HRp - rp * RP) >> 3;
    const bool wide = (col_stride & 7u) == 0u;  // (16-byte stores: every frame column then starts 16-byte aligned)
    const u32 n_items = (u32)g_cnt * (u32)nxs;
    const u32 g_magic = (u32)(((1ull << 32) + (u32)g_cnt - 1u) / (u32)max(g_cnt, 1));  // item / g_cnt (exact: items < 2^14, groups < 2^9)
    for (u32 it = (u32)tid; it < n_items; it += (u32)nthreads) {
      const u32 k = g_cnt > 1 ? __umulhi(it, g_magic) : it, g = it - k * (u32)g_cnt, gi = (u32)g_first + g;
      if (!((s_tab[n_ent + gi] >> k) & 1u)) continue;
      uint4* sl = reinterpret_cast<uint4*>(slots + (__umul24(k, (u32)RP) + 8u * g));
      const uint4 a = sl[0], b = sl[1];
      if (more) sl[0] = sl[1] = make_uint4(0, 0, 0, 0);
      XM_GLOBAL uint16_t* p = frame16 + (__umul24((s_tab[gi] >> 16) + k, col_stride) + (u32)r_lo + 8u * gi);
      if (wide) {
        *(XM_GLOBAL uint4*)p = make_uint4((a.x & 0xffffu) | (a.y << 16), (a.z & 0xffffu) | (a.w << 16), (b.x & 0xffffu) | (b.y << 16),
                                          (b.z & 0xffffu) | (b.w << 16));
      } else {
        const u32 v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int r = 0; r < 8; ++r)
          if ((int)(8u * gi) + r_lo + r < tb.rect_h) p[r] = (uint16_t)(v[r] & 0xffffu);  // (the last group's padding rows)
      }
    }
  }
  // ---- flush of the pass's rows: lanes = consecutive rows, each walks its row's band (mask and band position read once per
  //      row; the store of a wave = 64 consecutive rows of one sheared frame column); the slots are left cleared for the next pass
  //      (an event only ever lands on a cell its tile owns: a mask bit)
  if (!grouped) {
    XM_GLOBAL uint16_t* base = frame16 + (size_t)r_lo;
    const u32 col_stride = (u32)tb.rect_h;
    const int r_first = rp * RP, r_cnt = min(RP, HRp - r_first);
    for (int r_l = tid; r_l < r_cnt; r_l += nthreads) {
      const int r_i = r_first + r_l;
      const u32 bm = s_tab[r_i];
      u32 m = bm >> 16;
      const int col = (int)(short)(bm & 0xffffu) + tb.shear_bias + ((((r_i + r_lo) >> 3) * tb.shear_m) >> 12);  // (frame16_col)
      XM_GLOBAL uint16_t* p = base + (__umul24((u32)col, col_stride) + (u32)r_i);
      u32* sl = slots + r_l;
      while (m) {  // eight band columns at a time: the LDS reads go out together, then the stores
        u32 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (m >> k) ? sl[k * RP] : 0u;  // (nothing past the row's last band column: the slots end there)
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if ((m >> k) & 1u) {
            if (!XM_CABL(4) || v[k] == 0x12345678u) p[(u32)k * col_stride] = (uint16_t)(v[k] & 0xffffu);
            if (more) sl[k * RP] = 0;
          }
        m >>= 8;
        p += 8u * col_stride;
        sl += 8 * RP;
      }
    }
  }
  if (more) __syncthreads();
  }
  XM_CSTAMP(7);  // every row pass flushed
  if (XM_CABL(6) || XM_CABL(7)) bad = false;
  if (__ballot(bad) && lane == 0) {
    if (device_redo) {
      st->pad[1] = tag;
    } else {
      __hip_atomic_fetch_add(&st->cnt[parity][blk % CNT_SLOTS][CNT_UNSORTED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&st->unsorted_sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (u32* hf = st->host_flags) host_flag_store(hf, tag);
    }
  }
  if (lane == 0) {
    if (n_in) atomicAdd(&s_in, n_in);
    if (n_oob) atomicAdd(&s_oob, n_oob);
  }
  __syncthreads();

  {  // the extras, one by one
    const XM_GLOBAL u32* xc = (const XM_GLOBAL u32*)tb.own_extra_cells + x_first;
    for (int i = tid; i < n_extra; i += nthreads) frame16[xc[i]] = (uint16_t)(slots_x[i] & 0xffffu);
  }
  XM_CSTAMP(8);
  if (tid == 0) {
    XM_GLOBAL u32* c = st->cnt[parity][blk % CNT_SLOTS];
    if (s_in) __hip_atomic_fetch_add(&c[CNT_INLIER], s_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (s_oob) __hip_atomic_fetch_add(&c[CNT_OOB], s_oob, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <bool AOS, bool VEC, int EPT = COLS_EPT>
__global__ __launch_bounds__(COLS_MAX_THREADS, EPT == 4 ? 8 : XM_OWN_WAVES) void k_scatter_own(
    const uint16_t* __restrict__ xs, const uint16_t* __restrict__ ys, const long long* __restrict__ ts, const uint4* __restrict__ aos,
    u32 n, DevTables tb, SlotState* st, uint16_t* __restrict__ frame16, int W, int halo, int flags) {
  {  // every kernel argument in one scalar round trip (see k_scatter_tiled); never true
    const u64 pp = (u64)xs | (u64)ys | (u64)ts | (u64)aos | (u64)tb.lut | (u64)tb.xmap_own | (u64)tb.own_tiles | (u64)tb.own_bm | (u64)tb.xmap_extra | (u64)tb.own_extra_cells |
                   (u64)st | (u64)frame16;
    const int pi = tb.cam_w | tb.cam_h | tb.xmap_w | tb.xmap_h | tb.x_offset | tb.rect_w | tb.rect_h | W | halo | tb.own_hrp | tb.own_r_lo;
    if ((long long)(pp | (u64)(long long)pi) < 0) return;
  }
  scatter_own_body<AOS, VEC, EPT>((gp_u16)xs, (gp_u16)ys, (gp_i64)ts, (gp_u4)aos, n, tb, (gp_state)st, (XM_GLOBAL uint16_t*)frame16, W,
                                  halo, blockIdx.x, gridDim.x, flags);
}

template <bool AOS, bool VEC, int EPT = COLS_EPT>
__global__ __launch_bounds__(COLS_MAX_THREADS, EPT == 4 ? 8 : XM_OWN_WAVES) void k_scatter_own_batch(const FrameDesc* __restrict__ descs, DevTables tb, int W,
                                                                        int halo, int flags) {
  const FrameDesc d = descs[blockIdx.y];
  if (!d.valid || d.n == 0) return;
  scatter_own_body<AOS, VEC, EPT>((gp_u16)d.x, (gp_u16)d.y, (gp_i64)d.t, (gp_u4)d.aos, (u32)d.n, tb, (gp_state)d.st,
                                  (XM_GLOBAL uint16_t*)d.key_frame, W, halo, blockIdx.x, gridDim.x, flags, blockIdx.y);
}

}  // namespace xm
