// xmaps_k1cols.hpp -- K1 "column tiles": the fused per-event kernel of the verified-sorted projector-view path without
// atomics and without a key.  (gfx950 / MI355X; included by xmaps_hip.hip after xmaps_kernels.hpp)
//
// What the tiled K1 (k_scatter_tiled) pays for at full occupancy is the chip's L2 atomic request rate: its tiles are
// runs of 4096 consecutive EVENTS, so a time column of the X-map is flushed by ~1.4 tiles and the frame cell of a
// (time column, rectified row) pair must be resolved ACROSS tiles -- an atomic max on a packed (order | disparity) key,
// a tag per cell, a clear every 15 frames.  Here a tile is a run of W consecutive X-map TIME COLUMNS instead:
//
//   * cell(row, column) = (X[row, column] - x_offset, row) depends on the pair only (cam_proj_calibration.py:299-303 with
//     xpr = xr + disp = xp - x_offset).  If that map is injective (checked once in xm_create, k_cols_check) every frame cell
//     has exactly ONE (row, column) that can write it, hence exactly one owner tile: last-writer-wins is resolved entirely
//     in the tile's LDS slots (ds_max on (local index + 1) << 16 | disparity, as before) and the flush is a PLAIN STORE.
//   * every tile stores ALL of its live slots, winners and empties (0) alike: the frame is a plain u16 disparity frame
//     [rect_w][rect_h] (column-major) that is completely rewritten by every frame -- no tag, no clear, 2 bytes per cell
//     (K2 reads 4.6 MB instead of 9.3 MB at C-1M).  Cells that no (row, column) pair maps to are never written and stay
//     0 from xm_create.  A slot whose pair can never hold a winner (xp - x_offset < the smallest rectified x of the LUT:
//     the X-map's undefined cells) is "dead" and is skipped by the check and by the flush alike.
//   * the tile's events are the index range [lb(c0), lb(c0 + W)) of the time-sorted stream, lb(c) = first event whose time
//     column is >= c, found by a small kernel of its own (k_cols_bounds: 16 or 32 lanes per boundary, that many probes around the
//     interpolated position, then 4 consecutive events per lane = two dependent round trips for an evenly filled scan).  lb() is a deterministic function
//     of (stream, c), so neighbouring tiles share their boundary whatever the stream looks like; a non-monotone pair marks
//     the frame as failed.
//   * exactness for ANY input: every event a tile loads is checked -- t inside [t[0], t[n-1]] and its column inside the
//     tile's columns.  The ranges of the tiles partition [0, n), so if no tile objects every event went through the slots
//     of the tile that owns its column.  Otherwise the frame is marked as failed through the very flag of the sorted-order
//     verification and redone on the general 64-bit path (K0 -> k_scatter_tiled -> K2), automatically, like an unsorted frame.
//   * the X-map band is exactly the tile's W columns (no slack columns), the slot array W x xmap_h words: at C-1M (W = 2)
//     2640 slots are cleared / scanned per 3125 events instead of 6600 per 4096.
// Algorithmic bytes are those of K1: 24 B/event.
#pragma once
#include "xmaps_kernels.hpp"

namespace xm {

// Pointers of these kernels carry the global address space in their TYPE: a pointer read from a frame descriptor in memory
// is generic to the compiler otherwise (flat loads with 64-bit vector addresses instead of scalar base + 32-bit offset:
// 125 instead of 89 VGPRs in the multi-frame kernel).
#if defined(__HIP_DEVICE_COMPILE__)
#define XM_GLOBAL __attribute__((address_space(1)))
#else
#define XM_GLOBAL  // host pass: the bodies are only parsed (HIP's host-side vector types do not take qualified references)
#endif
typedef const XM_GLOBAL uint16_t* gp_u16;
typedef const XM_GLOBAL long long* gp_i64;
typedef const XM_GLOBAL uint4* gp_u4;
typedef const XM_GLOBAL int4* gp_i4;
typedef XM_GLOBAL SlotState* gp_state;

#ifdef XM_ABLATE  // experiments (tools/cols_timeline.py): s_memtime stamps of thread 0 of the first 64 tiles of frame XM_CSTAMP_FRAME
#ifndef XM_CSTAMP_FRAME
#define XM_CSTAMP_FRAME 30
#endif
#define XM_CSTAMP(ph) do { if (threadIdx.x == 0 && blockIdx.y == XM_CSTAMP_FRAME && blockIdx.x < 64) g_timeline[blockIdx.x][ph] = __builtin_amdgcn_s_memtime(); } while (0)
#define XM_CABL(bit) (g_ablate & (1 << (bit)))  /* bit 4: no flush stores, 5: no band loads, 6: no event loads, 7: no per-event work */
#else
#define XM_CABL(bit) false
#define XM_CSTAMP(ph) do { } while (0)
#endif

constexpr int COLS_EPT = 8;          // events per thread and pass
enum { COLS_F_DEVICE_REDO = 1, COLS_F_ALL_IN_FRAME = 2, COLS_F_EXT_EXTREMA = 4 };
// COLS_F_EXT_EXTREMA (shards: a contiguous piece of a frame's sorted stream): the FRAME's extrema come from a 16-byte device buffer
// {tmin, -tmax} (FrameDesc.p carries its address; the path has no polarity column) instead of the piece's first / last stamp, and a
// piece without events still writes its zeros
constexpr u32 COLS_MAX_TILE_EVENTS = 65535u - 8u;  // the slot value carries (local index + 1) in 16 bits

// ---- xm_create: is cell(row, column) injective over the live pairs?  One block per rectified row. -----------------------
// live  = xp - x_offset >= xr_min (some LUT entry can give disp >= 0; the host has checked that xp - xr - x_offset never
//         leaves the int16 range, so the reference's wrap-around arithmetic is plain arithmetic on this rig)
// cell  = column (xp - x_offset), one negative wrap like NumPy, inside the frame
__device__ inline bool cols_cell(const DevTables& tb, int xp, int r, int xr_min, u32& cell) {
  const int fu = xp - tb.x_offset;
  if (fu < xr_min) return false;
  int fc = (int)(short)fu;
  if (fc < 0) fc += tb.rect_w;
  if (fc < 0 || fc >= tb.rect_w || r >= tb.rect_h) return false;
  cell = __umul24((u32)fc, (u32)tb.rect_h) + (u32)r;  // (24-bit multiplies are full rate, v_mul_lo_u32 is a quarter)
  return true;
}

// n_dup[0]: live pairs that share a cell with another one; n_dup[1]: live pairs whose cell lies outside the frame (an event there
// is an IndexError in the reference: when there are none, K1 need not test the cell of every event)
__global__ __launch_bounds__(BLOCK) void k_cols_check(DevTables tb, int xr_min, u32* __restrict__ n_dup) {
  __shared__ u32 bits[2048];  // rect_w <= 65536 columns
  const int r = blockIdx.x;   // rows 0 .. min(xmap_h - 1, rect_h) - 1 (the last X-map row never holds a winner: xmd:23)
  for (int i = threadIdx.x; i < 2048; i += BLOCK) bits[i] = 0;
  __syncthreads();
  u32 dup = 0, outside = 0;
  for (int c = threadIdx.x; c < tb.xmap_w; c += BLOCK) {
    const int xp = (int)tb.xmap[(u32)c * (u32)tb.xmap_h + (u32)r];
    u32 cell;
    if (!cols_cell(tb, xp, r, xr_min, cell)) {
      outside += xp - tb.x_offset >= xr_min ? 1u : 0u;  // live, but no cell
      continue;
    }
    const u32 fc = cell / (u32)tb.rect_h, bit = 1u << (fc & 31);
    if (atomicOr(&bits[fc >> 5], bit) & bit) dup += 1;
  }
  if (dup) atomicAdd(n_dup, dup);
  if (outside) atomicAdd(n_dup + 1, outside);
}

// ---- the event-range search (G = 16 or 32 lanes per boundary) -----------------------------------------------------------
template <bool AOS>
__device__ __forceinline__ long long cols_t_at(gp_i64 ts, gp_u4 aos, int i) {
  if constexpr (AOS) {
    const uint4 r = aos[i];
    return (long long)(((u64)r.w << 32) | r.z);
  } else {
    return ts[i];
  }
}

// K1 never converts a time stamp: for int64 stamps the X-map column is a step function of a = t - tmin, so the frame's
// columns are described exactly by THRESHOLDS  thr[c] = the smallest a in [0, span + 1] with column(tmin + a) >= c
// (span = tmax - tmin; span + 1 = "no such a"), computed here with the very conversion that is bit-exact with NumPy
// (TimeNorm, xmaps_kernels.hpp).  Then  column(t) = #{c' >= 1 : thr[c'] <= a}  and, for a tile of columns [c0, c1):
// event in tile <=> thr[c0] <= a < thr[c1]; its column = c0 + #{interior c' : thr[c'] <= a}: a few 32-bit compares per event
// instead of the FP64 chain, and half the registers.
__device__ inline u32 cols_threshold(const TimeNorm<long long>& tn, const long long tmin, const u32 span, const int c, const int S) {
  if (c <= 0) return 0u;
  const auto col_at = [&](long long a) { return tn.column(tmin + a); };
  // column c starts where (a / span) * S = c - 0.5
  const double est = ((double)c - 0.5) * (double)span / (double)max(S, 1);
  long long a = (long long)fmin(fmax(est, 0.0), (double)span);
  int guard = 0;
  while (a > 0 && guard < 16 && col_at(a - 1) >= c) { --a; ++guard; }
  while (a <= (long long)span && guard < 32 && col_at(a) < c) { ++a; ++guard; }
  const bool settled = (a == 0 || col_at(a - 1) < c) && (a > (long long)span || col_at(a) >= c);
  if (!settled) {  // (never seen; the conversion is monotone, so a bisection over [0, span + 1] is exact)
    long long lo = -1, hi = (long long)span + 1;  // column(lo) < c (virtual at -1), column(hi) >= c (virtual at span + 1)
    while (hi - lo > 1) {
      const long long mid = lo + ((hi - lo) >> 1);
      if (col_at(mid) >= c) hi = mid; else lo = mid;
    }
    a = hi;
  }
  return (u32)a;
}

// debug / tests: thr[c] for c = 0 .. xmap_w of a frame whose first / last stamps are t_first / t_last (what k_cols_bounds writes
// behind the frame), one thread per column
__global__ __launch_bounds__(BLOCK) void k_debug_cols_thresholds(long long t_first, long long t_last, int S, int xmap_w,
                                                                 u32* __restrict__ out) {
  const int c = blockIdx.x * BLOCK + threadIdx.x;
  if (c > xmap_w) return;
  if (t_last < t_first) t_last = t_first;
  const TimeNorm<long long> tn(t_first, t_last, S);
  out[c] = cols_threshold(tn, t_first, (u32)(t_last - t_first), c, S);
}

template <bool AOS>
__device__ __forceinline__ int cols_x_at(gp_u16 xs, gp_u4 aos, int i) {
  if constexpr (AOS) return (int)(aos[i].x & 0xffff);
  else return (int)xs[i];
}

// ---- K0b's search: G lanes per boundary ------------------------------------------------------------------------
// lb = first event with (u64)(t - tmin) >= A.  State per boundary: event lo is below (or lo == -1), event hi is at or past it (or
// hi == n); the answer is hi once hi - lo == 1.  Round 1: G probes, G * FIN events apart, centred on where an evenly filled scan
// has the boundary (known before anything is loaded).  Last round (<= G * FIN unknown positions): every lane takes FIN consecutive
// events, t and x -- the boundary AND the six x values around it (the camera-column window of K1) come out of one round trip.
// In between (a guess that missed by more than G * FIN * G / 2 events: bursts, unsorted streams): even splits into G probes.
// For a stream that is not sorted the result is still a deterministic function of (stream, A): neighbouring tiles read the same
// boundary, and the per-event verification of K1 catches the rest.
// G = 32 lanes per boundary, or 16 (round 6) where a tile is at most 16 columns wide and has no halo boundary (the column tiles:
// lane l of a group computes the threshold of column j W + l): half the waves and half the probe lines per boundary -- in the
// pipelined step a kernel costs about what it costs alone, K0b's scattered 8-byte probes pull a line each (56 MB per group
// of 32 C-1M frames), and 16 lanes move the step from 0.1964 to 0.1937 ms (profiles/r06_k1_chain.md section 3).  The 16-lane form
// also probes less: round 1 is FOUR probes 1 024 events apart, and the 1 024-event window they leave is entered by interpolation
// (the stamps at its two ends say where in it the threshold falls: G * FIN consecutive events around that place; an evenly filled
// scan misses such a window by ~16 events rms) -- ten lines per boundary instead of twenty-two, K0b 16.9 -> 13.0 us, the step
// 0.1930 -> 0.1908 ms.  At most three interpolated windows, then the even splits: what is read decides, the stamps only guess.
constexpr int COLS_BOUNDS_FIN = 4;
constexpr int cols_bounds_per_block(int G) { return 256 / G; }

// probes of one round, FIN per lane in probing order (lane-major), act[k] / pr[k] = probed / at or past the boundary, q[k] their
// positions (non-decreasing in probing order): narrow (lo, hi).  Executed by the whole wave; gl = the group's first lane.
template <int G>
__device__ __forceinline__ void cols_narrow(const int (&q)[COLS_BOUNDS_FIN], const bool (&act)[COLS_BOUNDS_FIN],
                                            const bool (&pr)[COLS_BOUNDS_FIN], const int gl, int& lo, int& hi) {
  constexpr int FIN = COLS_BOUNDS_FIN;
  int kk = FIN, last_q = -1;  // the lane's first probe at or past the boundary; its last probe
  bool any_act = false;
#pragma unroll
  for (int k = FIN - 1; k >= 0; --k) {
    if (pr[k]) kk = k;
    if (act[k] && last_q < 0) last_q = q[k];
    any_act = any_act || act[k];
  }
  int pos = q[0], prev = -1;  // position of that probe, and of the lane's probe in front of it (-1: it is the lane's first)
#pragma unroll
  for (int k = 1; k < FIN; ++k)
    if (kk == k) {
      pos = q[k];
      prev = q[k - 1];
    }
  const u64 group_mask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
  const u64 gb = (__ballot(kk < FIN) >> gl) & group_mask, ga = (__ballot(any_act) >> gl) & group_mask;
  const int jj = gb ? __builtin_ctzll(gb) : 0, top = ga ? 63 - __builtin_clzll(ga) : 0;
  const int p_hit = __shfl(pos, gl + jj, 64), p_prev = __shfl(prev, gl + jj, 64);
  const int p_before = __shfl(last_q, gl + max(jj - 1, 0), 64), p_top = __shfl(last_q, gl + top, 64);
  if (gb) {
    hi = p_hit;
    if (p_prev >= 0) lo = max(lo, p_prev);
    else if (jj > 0) lo = max(lo, p_before);
  } else if (ga) {
    lo = max(lo, p_top);  // every probe is still below the boundary
  }
}

// ---- K0b: the tile boundaries of one frame, G lanes per boundary ----------------------------------------------------------------
// bounds[j] = {lb(j * W), median x of the three events at / behind it, median x of the three events in front of it, 0} for
// j = 0 .. nb (nb = ceil(xmap_w / W) tiles; bounds[nb].x = n), thr[c] for c = 0 .. xmap_w (see cols_threshold).  Tile j of K1
// owns events [bounds[j].x, bounds[j+1].x) and centres its camera-column window between bounds[j].y and bounds[j+1].z.
// Kept out of K1 on purpose: inside K1 the search is two to three dependent round trips at the head of every tile's chain,
// with the tile's other waves parked at a barrier.  A frame that spans 2^32 us or more gets bounds[j].x = -1: K1 objects.

// where the bounds and the thresholds live: behind the slot's u16 frame (one allocation, one pointer in the frame descriptor)
__host__ __device__ inline size_t cols_bounds_offset(size_t key_cells) { return (key_cells * 2 + 63) & ~(size_t)63; }
__host__ __device__ inline size_t cols_thr_offset(size_t key_cells, int xmap_w) {
  return cols_bounds_offset(key_cells) + sizeof(int4) * ((size_t)xmap_w + 2);
}
__host__ __device__ inline size_t cols_frame_bytes(size_t key_cells, int xmap_w) {
  return cols_thr_offset(key_cells, xmap_w) + sizeof(u32) * ((size_t)xmap_w + 2) + 64;
}

template <bool AOS, int G>
__device__ __forceinline__ void cols_bounds_body(gp_u16 xs, gp_i64 ts, gp_u4 aos, const int n, const DevTables& tb, const int W,
                                                 XM_GLOBAL unsigned char* frame_base, const u32 blk, gp_i64 ext_mm = nullptr,
                                                 const int first = 0,  // first: events [0, first) are filler (shards: alignment)
                                                 const int split = 0) {  // owner tiles: boundaries at t W and t W + split (a tile's
                                                                         // own columns and its halo's end), not every W columns
  typedef long long T;
  const size_t key_cells = frame16_cells(tb);
  XM_GLOBAL int4* bounds = (XM_GLOBAL int4*)(frame_base + cols_bounds_offset(key_cells));
  XM_GLOBAL u32* thr = (XM_GLOBAL u32*)(frame_base + cols_thr_offset(key_cells, tb.xmap_w));
  constexpr int FIN = COLS_BOUNDS_FIN;
  constexpr bool INTERP = G == 16;
  const int lane = threadIdx.x & 63, sl = lane & (G - 1), gl = lane & ~(G - 1);
  const int nb = split ? 2 * ((tb.xmap_w + W - 1) / W) : (tb.xmap_w + W - 1) / W;
  const int j_raw = (int)blk * cols_bounds_per_block(G) + (int)threadIdx.x / G;
  const bool live = j_raw <= nb;  // (a group past the last boundary runs along with its wave and stores nothing)
  if (!__any(live)) return;       // wave-uniform
  const int j = min(j_raw, nb);
  T t_first, t_last;
  if (ext_mm) {  // the frame's extrema as the shards agreed on them ({tmin, -tmax})
    t_first = ext_mm[0];
    t_last = -ext_mm[1];
  } else if constexpr (AOS) {
    const uint4 a = aos[0], b = aos[n - 1];
    t_first = (T)(((u64)a.w << 32) | a.z);
    t_last = (T)(((u64)b.w << 32) | b.z);
  } else {
    t_first = ts[0];
    t_last = ts[n - 1];
  }
  // The first round of probes goes out together with the frame's first / last stamp: an evenly filled scan has the first
  // event of column c near n (c - 1/2) / S (the threshold itself is (c - 1/2) span / S rounded), so where to probe does not
  // depend on anything loaded.  A wave of this kernel is a chain of dependent round trips at loaded-memory latency (its
  // arithmetic hides behind them): stamps -> probes -> probes -> x was four of them, now it is two.
  const int c = min(split ? (j >> 1) * W + (j & 1) * split : j * W, tb.xmap_w);
  const int wj = split ? ((j & 1) ? W - split : split) : W;  // the columns in front of the next boundary (<= G: own_plan)
  const bool search = live && c > 0 && c < tb.xmap_w && n > 0;  // else: boundary 0 is event 0, the last tile takes whatever is left
  int q[FIN];
  bool act[FIN], pr[FIN];
  T tv[FIN];
#pragma unroll
  for (int k = 0; k < FIN; ++k) {
    q[k] = 0;
    act[k] = pr[k] = false;
    tv[k] = 0;
  }
  if (search) {
    const double g = ((double)c - 0.5) / (double)max(tb.t_px_scale, 1) * (double)n;
    // G = 32: 32 probes G * FIN events apart.  G = 16 (INTERP): FOUR probes 1024 events apart (+-2048 events around the estimate,
    // four lines instead of sixteen); the window they leave is entered by interpolating between its ends' stamps (below)
    const int step1 = INTERP ? (sl < 4 ? (2 * sl - 3) * 512 : 0) : (sl - G / 2) * (G * FIN);
    q[0] = min(max((int)fmin(fmax(g, 0.0), (double)(n - 1)) + step1, 0), n - 1);
    if (!INTERP || sl < 4) tv[0] = cols_t_at<AOS>(ts, aos, q[0]);
  }
  if (t_last < t_first) t_last = t_first;  // not sorted at all: keep the arithmetic defined; K1's verification flags the frame
  const u64 span64 = (u64)(t_last - t_first);
  if (span64 >= 0xffffffffull) {  // a - tmin does not fit 32 bits: not this path
    if (sl == 0 && live) bounds[j] = make_int4(-1, 0, 0, 0);
    return;
  }
  const u32 span = (u32)span64;
  const TimeNorm<T> tn(t_first, t_last, tb.t_px_scale);
  // thresholds of this boundary's column and of the interior columns behind it (lane l of the group: column j W + l)
  u32 A = 0;
  if (live && sl < wj && c + sl <= tb.xmap_w && (sl == 0 || j < nb)) {
    A = cols_threshold(tn, t_first, span, c + sl, tb.t_px_scale);
    thr[c + sl] = A;
  }
  A = __shfl(A, gl, 64);
  int lo = -1, hi = n;
  if (c <= 0 || !live || n == 0) hi = 0;
  else if (c >= tb.xmap_w) lo = n - 1;
  if (hi - lo > 1) {  // == search
    act[0] = !INTERP || sl < 4;
    pr[0] = act[0] && (u64)(tv[0] - t_first) >= (u64)A;
  }
  cols_narrow<G>(q, act, pr, gl, lo, hi);
  // INTERP: a = t - tmin at the two ends of (lo, hi), for the interpolation (index -1 <-> 0, index n <-> span + 1; an end that is
  // one of this round's probes: its lane's stamp).  Only ever a GUESS of where to read: what is read decides.
  u32 a_lo = 0, a_hi = span == 0xffffffffu ? span : span + 1u;
  const auto ends_from = [&](const bool on, const int start, const int count) {  // the probes were `count` consecutive events from `start`
    const u64 af = (u64)(__shfl(tv[0], gl, 64) - t_first), al = (u64)(__shfl(tv[FIN - 1], gl + G - 1, 64) - t_first);
    if (on && hi == start) a_hi = (u32)min(af, (u64)0xffffffffu);
    if (on && lo == start + count - 1) a_lo = (u32)min(al, (u64)0xffffffffu);
  };
  if constexpr (INTERP) {
    const u32 a0 = (u32)min((u64)(tv[0] - t_first), (u64)0xffffffffu);
    const u64 gm = (1ull << G) - 1ull;
    const u64 m_lo = (__ballot(act[0] && q[0] == lo) >> gl) & gm, m_hi = (__ballot(act[0] && q[0] == hi) >> gl) & gm;
    const u32 v_lo = __shfl(a0, gl + (m_lo ? __builtin_ctzll(m_lo) : 0), 64), v_hi = __shfl(a0, gl + (m_hi ? __builtin_ctzll(m_hi) : 0), 64);
    if (m_lo) a_lo = v_lo;
    if (m_hi) a_hi = v_hi;
  }
  int x_base = -1, x_cnt = 0, rounds = 0;
  u64 x4 = 0;  // the lane's FIN x values of the last round, 16 bits each
  while (__any(hi - lo > 1)) {
    const bool need = hi - lo > 1;
    const int unknown = hi - lo - 1;          // positions lo + 1 .. hi - 1
    const bool fin = unknown <= G * FIN;      // consecutive events: this round settles the boundary
    const int stride = (unknown + G - 1) / G;
    // INTERP, at most three times (then the even splits: adversarial stamps must not make this a walk): G * FIN consecutive events
    // around where the boundary would lie if the stamps between the window's ends were evenly spread
    const bool dense = fin || (INTERP && rounds < 3);
    int start = lo + 1;
    if (!fin && dense) {
      const double fr = a_hi > a_lo ? fmin(fmax(((double)A - (double)a_lo) / ((double)a_hi - (double)a_lo), 0.0), 1.0) : 0.5;
      start = min(max(lo + 1 + (int)(fr * (double)unknown) - G * FIN / 2, lo + 1), hi - G * FIN);
    }
    rounds += 1;
    u32 xk[FIN];
#pragma unroll
    for (int k = 0; k < FIN; ++k) {
      q[k] = dense ? start + FIN * sl + k : lo + (sl + 1) * stride;
      act[k] = need && q[k] < hi && (dense || k == 0);
      xk[k] = 0;
      if constexpr (AOS) {
        const uint4 r = aos[act[k] ? q[k] : 0];
        tv[k] = (T)(((u64)r.w << 32) | r.z);
        xk[k] = r.x & 0xffffu;
      } else {
        tv[k] = ts[act[k] ? q[k] : 0];
        if (dense) xk[k] = xs[act[k] ? q[k] : 0];
      }
      pr[k] = act[k] && (u64)(tv[k] - t_first) >= (u64)A;
    }
    if (need) {
      x_base = dense ? start : -1;
      x_cnt = dense ? min(G * FIN, hi - start) : 0;
      x4 = 0;
#pragma unroll
      for (int k = 0; k < FIN; ++k) x4 |= (u64)xk[k] << (16 * k);
    }
    cols_narrow<G>(q, act, pr, gl, lo, hi);
    if constexpr (INTERP) ends_from(dense && !fin && need, start, G * FIN);
  }
  const int lb = max(hi, first);
  // median x of the three events at / behind the boundary and of the three in front of it: from the last round's probes where
  // they cover them, else loaded now
  const int i = max(sl < 3 ? min(lb + sl, n - 1) : max(lb - 1 - (sl - 3), 0), 0);
  const int src = i - x_base;
  const bool from_probe = x_base >= 0 && src >= 0 && src < x_cnt;
  const u64 got = __shfl(x4, gl + ((src >> 2) & (G - 1)), 64);
  int xv = (int)((got >> (16 * (src & 3))) & 0xffffull);
  if (sl < 6 && live && !from_probe && n > 0) xv = cols_x_at<AOS>(xs, aos, i);
  const int a0 = __shfl(xv, gl + 0, 64), a1 = __shfl(xv, gl + 1, 64), a2 = __shfl(xv, gl + 2, 64);
  const int e0 = __shfl(xv, gl + 3, 64), e1 = __shfl(xv, gl + 4, 64), e2 = __shfl(xv, gl + 5, 64);
  if (sl == 0 && live)
    bounds[j] = make_int4(lb, max(min(a0, a1), min(max(a0, a1), a2)), max(min(e0, e1), min(max(e0, e1), e2)), 0);
}

template <bool AOS, int G = 32>
__global__ __launch_bounds__(256) void k_cols_bounds(const uint16_t* __restrict__ xs, const long long* __restrict__ ts,
                                                                        const uint4* __restrict__ aos, u32 n, DevTables tb, int W,
                                                                        uint16_t* __restrict__ frame16, int split = 0) {
  cols_bounds_body<AOS, G>((gp_u16)xs, (gp_i64)ts, (gp_u4)aos, (int)n, tb, W, (XM_GLOBAL unsigned char*)frame16, blockIdx.x, nullptr, 0,
                           split);
}

template <bool AOS, int G = 32>
__global__ __launch_bounds__(256) void k_cols_bounds_batch(const FrameDesc* __restrict__ descs, DevTables tb, int W, int flags = 0,
                                                           int split = 0) {
  const FrameDesc d = descs[blockIdx.y];
  const bool ext = flags & COLS_F_EXT_EXTREMA;
  if (!d.valid || (d.n == 0 && !ext)) return;
  cols_bounds_body<AOS, G>((gp_u16)d.x, (gp_i64)d.t, (gp_u4)d.aos, (int)d.n, tb, W, (XM_GLOBAL unsigned char*)d.key_frame, blockIdx.x,
                           ext ? (gp_i64)d.p : nullptr, ext ? (int)d.pad : 0, split);
}

// ---- the kernel body ---------------------------------------------------------------------------------------------------------
// blk / nblk: this block's index among the frame's ceil(xmap_w / W) blocks.  frame16: the slot's plain u16 disparity frame,
// followed by what k_cols_bounds left for this frame (bounds, thresholds).
template <bool AOS, bool VEC>
__device__ __forceinline__ void scatter_cols_body(gp_u16 xs, gp_u16 ys, gp_i64 ts, gp_u4 aos, const u32 n_ev, const DevTables& tb,
                                                  gp_state st, XM_GLOBAL uint16_t* frame16, const int W, const int w_x,
                                                  const int xr_min, const u32 blk, const u32 nblk, const int flags = 0,
                                                  gp_i64 ext_mm = nullptr) {
  // flags: COLS_F_DEVICE_REDO = inside a hipGraph (a failing tile leaves the frame's tag in SlotState.pad[1] for the redo kernels
  // behind this one instead of telling the host); COLS_F_ALL_IN_FRAME = every live (row, column) pair of this rig has its cell
  // inside the frame (xm_create), so an event that passes xmd:29 cannot be an IndexError: no per-event cell test
  const bool device_redo = flags & COLS_F_DEVICE_REDO, all_in = flags & COLS_F_ALL_IN_FRAME;
  typedef long long T;
  static_assert(!(AOS && VEC), "AoS records are loaded one per lane");
  constexpr int EPT = COLS_EPT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ u32 s_in, s_oob, s_sentinel;
  const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63;
  const int n = (int)n_ev;
  const int cap = nthreads * EPT;  // events per pass
  const size_t key_cells = frame16_cells(tb);
  gp_i4 bounds = (gp_i4)((const XM_GLOBAL unsigned char*)frame16 + cols_bounds_offset(key_cells));
  const XM_GLOBAL u32* thr = (const XM_GLOBAL u32*)((const XM_GLOBAL unsigned char*)frame16 + cols_thr_offset(key_cells, tb.xmap_w));
  // LDS carve-up (uint4 units; each band keeps 1 quad of alignment slack in front and a wave of slack behind it for the
  // LDS-direct loads, which write whole waves).  Mirrored by cols_lds_bytes() on the host.
  const int lut_q = ((w_x * tb.cam_h + 3) >> 2) + 1 + 64;
  const int xm_q = ((W * tb.xmap_h + 7) >> 3) + 1 + 64;
  const int slot_q = (W * tb.xmap_h + 3) >> 2;
  u32* lut_base = reinterpret_cast<u32*>(smem);
  int16_t* xm_base = reinterpret_cast<int16_t*>(lut_base + 4 * lut_q);
  u32* slots = reinterpret_cast<u32*>(xm_base + 8 * xm_q);
  uint4* l_lut = reinterpret_cast<uint4*>(lut_base);
  uint4* l_xm = reinterpret_cast<uint4*>(xm_base);
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int dma_q0 = tid & ~63;

  const u32 tile = xcd_contiguous(blk, nblk);
  const int c0 = (int)tile * W;
  const int Wc = min(W, tb.xmap_w - c0);  // >= 1: nblk = ceil(xmap_w / W)
  const int nslots = Wc * tb.xmap_h;
  XM_CSTAMP(0);

  // ---- 1. everything that locates the tile, as uniform loads in one round trip: its event range and camera-column window,
  //         the thresholds of its first and one-past-last column (k_cols_bounds), the frame's first / last time stamp, the tag
  const int4 b_lo = bounds[tile], b_hi = bounds[tile + 1];
  const u32 A_lo = thr[c0], A_hi = thr[c0 + Wc];
  // the first interior thresholds ride along (W = 2 at C-1M: one): fetched inside the event arithmetic they were a scalar
  // round trip in the middle of it
  constexpr int A_PRE = 3;
  u32 A_in[A_PRE];
#pragma unroll
  for (int i = 0; i < A_PRE; ++i) A_in[i] = thr[c0 + min(1 + i, Wc)];
  T t_first, t_last;
  if (ext_mm) {
    t_first = ext_mm[0];
    t_last = -ext_mm[1];
  } else if constexpr (AOS) {
    const uint4 a = aos[0], b = aos[n - 1];
    t_first = (T)(((u64)a.w << 32) | a.z);
    t_last = (T)(((u64)b.w << 32) | b.z);
  } else {
    t_first = ts[0];
    t_last = ts[n - 1];
  }
  const u32 tag = st->tag_b + 1;
  // the tile's X-map band (exactly its columns) needs none of that: L2 -> LDS with LDS-direct loads, issued at once
  const u32 xm_start = (u32)c0 * (u32)tb.xmap_h, xm_shift = xm_start & 7u;  // in int16
  const int16_t* xm_t = xm_base + xm_shift;
  const uint4* g_xm = reinterpret_cast<const uint4*>(tb.xmap + (xm_start - xm_shift));
  const int nq_xm = (int)((xm_shift + (u32)nslots + 7u) >> 3);
  if (!XM_CABL(5))
  for (int q0 = dma_q0; q0 < nq_xm; q0 += nthreads)
    __builtin_amdgcn_global_load_lds((glb_void*)(g_xm + min(q0 + lane, nq_xm - 1)), (lds_void*)(l_xm + q0), 16, 0, 0);
  {  // winner slots
    uint4* l_slots = reinterpret_cast<uint4*>(slots);
    for (int i = tid; i < slot_q; i += nthreads) l_slots[i] = make_uint4(0, 0, 0, 0);
  }
  if (tid == 0) {
    s_in = 0;
    s_oob = 0;
    s_sentinel = 0x80000000u;  // a LUT word with yr = -32768: what an event outside the tile / window "reads" (fails xmd:23)
  }

  int lb_s = b_lo.x, lb_e = b_hi.x;
  bool bad = false;  // this tile objects: the frame is redone on the general path
  if (lb_s < 0 || lb_e > n || lb_e < lb_s || (u32)(lb_e - lb_s) > COLS_MAX_TILE_EVENTS) {
    bad = true;  // boundaries out of order (a stream that is not sorted), a frame of >= 2^32 us, or more events than the
    lb_s = lb_e = 0;  // slots' 16-bit order holds.  Skip the events: the frame's result is discarded anyway
  }
  const int x_lo = min(max(((b_lo.y + b_hi.z) >> 1) - w_x / 2, 0), max(tb.cam_w - w_x, 0));
  XM_CSTAMP(1);

  // ---- 2. the first pass' events (cap = nthreads * EPT per pass).  VEC: 16-byte loads of 8 consecutive events from an
  //         8-aligned start (events in front of lb_s / behind lb_e are masked); otherwise lane-strided loads from lb_s.
  const int a0 = VEC ? (lb_s & ~(EPT - 1)) : lb_s;
  // passes = ceil((lb_e - a0) / cap), by subtraction (<= 16 for the largest tile the slots' order field holds: an integer
  // division is ~25 instructions, several of them quarter rate)
  int n_pass = 0;
  for (int left = lb_e > lb_s ? lb_e - a0 : 0; left > 0; left -= cap) n_pass += 1;
  u32 xw[EPT / 2], yw[EPT / 2];
  T tt[EPT];
  const auto load_events = [&](const int pass) {
    if constexpr (VEC) {
      const int base_true = a0 + pass * cap + tid * EPT;
      const int last_grp = (n - 1) & ~(EPT - 1);
      const int base = min(base_true, last_grp);  // a thread past the end re-reads the last group (masked)
      const uint4 xv = *(gp_u4)(xs + base);
      const uint4 yv = *(gp_u4)(ys + base);
      xw[0] = xv.x; xw[1] = xv.y; xw[2] = xv.z; xw[3] = xv.w;
      yw[0] = yv.x; yw[1] = yv.y; yw[2] = yv.z; yw[3] = yv.w;
#pragma unroll
      for (int q = 0; q < EPT / 2; ++q) {  // a pair that starts past the end is redirected to the group's first pair
        const longlong2 a = *(const XM_GLOBAL longlong2*)(ts + (base + 2 * q < n ? base + 2 * q : base));
        tt[2 * q] = a.x;
        tt[2 * q + 1] = a.y;
      }
    } else {
#pragma unroll
      for (int q = 0; q < EPT / 2; ++q) xw[q] = yw[q] = 0;
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int i = lb_s + pass * cap + k * nthreads + tid;
        const int ic = i < lb_e ? i : lb_s;
        if constexpr (AOS) {
          const uint4 r = aos[ic];
          xw[k >> 1] |= (r.x & 0xffff) << ((k & 1) * 16);
          yw[k >> 1] |= (r.x >> 16) << ((k & 1) * 16);
          tt[k] = (T)(((u64)r.w << 32) | r.z);
        } else {
          xw[k >> 1] |= (u32)xs[ic] << ((k & 1) * 16);
          yw[k >> 1] |= (u32)ys[ic] << ((k & 1) * 16);
          tt[k] = ts[ic];
        }
      }
    }
  };
  // A wave whose first event of a pass lies behind the range has nothing to do in that pass (at C-1M, 3125 events for 512 x 8
  // slots: the tile's last wave always, the one before it mostly): it skips the loads and the per-event work -- an eighth of
  // the kernel's instruction issue -- and only keeps the barriers, the band copies, the clear and the flush company.
  const auto wave_on = [&](const int pass) {  // wave-uniform
    const int w0 = VEC ? a0 + pass * cap + (tid & ~63) * EPT : lb_s + pass * cap + (tid & ~63);
    return w0 < lb_e;
  };
  if (n_pass > 0 && wave_on(0) && !XM_CABL(6)) {
    load_events(0);
    // the compiler waits with vmcnt(0) before the first use of a register loaded BEFORE an LDS-direct load: touch the event
    // registers here, so that the wait sits in front of the LUT band's loads and the band flies during the event arithmetic
#pragma unroll
    for (int q = 0; q < EPT / 2; ++q) asm volatile("" : "+v"(xw[q]), "+v"(yw[q]));
#pragma unroll
    for (int k = 0; k < EPT; ++k) asm volatile("" : "+v"(tt[k]));
  }
  XM_CSTAMP(2);
  // ---- 3. the LUT band (w_x camera columns around the range's x) -> LDS ---------------------------------------------------------
  const int wx_eff = min(w_x, tb.cam_w);
  const u32 lut_start = (u32)x_lo * (u32)tb.cam_h, lut_shift = lut_start & 3u;  // in words
  const u32* lut_t = lut_base + lut_shift;
  const uint4* g_lut = reinterpret_cast<const uint4*>(tb.lut + (lut_start - lut_shift));
  const int nq_lut = (int)((lut_shift + (u32)wx_eff * (u32)tb.cam_h + 3u) >> 2);
  if (n_pass > 0 && !XM_CABL(5))
    for (int q0 = dma_q0; q0 < nq_lut; q0 += nthreads)
      __builtin_amdgcn_global_load_lds((glb_void*)(g_lut + min(q0 + lane, nq_lut - 1)), (lds_void*)(l_lut + q0), 16, 0, 0);

  // ---- 4. frame extrema = (t[0], t[n-1]), verified per event below; slot bookkeeping by block 0 (as k_scatter_tiled does in
  //         its time-sorted mode: K2 reads tag_a and copies it to tag_b)
  const u32 parity = tag & 1;
  if (t_last < t_first) t_last = t_first;  // not sorted at all (k_cols_bounds did the same): the verification flags the frame
  if (blk == 0) {
    if (tid == 0) {
      st->tag_a = tag;
      st->mm[parity][0][0] = TimeCodec<T>::enc(t_first);  // xm_frame_stats.t_min / t_max
      st->mm[parity][0][1] = TimeCodec<T>::enc(t_last);
    }
    for (int i = tid; i < MM_SLOTS; i += nthreads) {
      st->mm[parity ^ 1][i][0] = MM_INIT_MIN;
      st->mm[parity ^ 1][i][1] = MM_INIT_MAX;
    }
  }

  u32 n_in = 0, n_oob = 0;  // per-lane counters (summed over the wave at the end: no ballot + popcount per event)
  for (int pass = 0; pass < n_pass; ++pass) {
    const bool on = wave_on(pass) && !XM_CABL(7);
    if (pass > 0 && on) load_events(pass);
    int e0;  // order of the thread's event 0 inside the tile (event index - a0); event k: + k (VEC) / + k * nthreads
    if constexpr (VEC) e0 = pass * cap + tid * EPT;
    else e0 = pass * cap + tid;
    const u32 used_n = (u32)(lb_e - lb_s), A_span = A_hi - A_lo;  // (thresholds are non-decreasing in the column)
    const int u0 = VEC ? a0 + e0 - lb_s : e0;                      // (index of the thread's event 0) - lb_s
    int tl[EPT], xl[EPT];
    bool fast[EPT];
    u32 smask = 0;  // events outside the LUT window (x noise)
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      tl[k] = 0;
      xl[k] = 0;
      fast[k] = false;
    }
    if (on) {
    // the event's column inside the tile and the verification, in integers: a = t - tmin against the columns' thresholds
    u32 av[EPT];
    bool live[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const u64 a64 = (u64)(tt[k] - t_first);
      av[k] = (u32)a64;
      // one unsigned compare each: (event index - lb_s) < count, (a - thr[c0]) < (thr[c1] - thr[c0]) (+ the high word)
      const bool used = (u32)(u0 + (VEC ? k : k * nthreads)) < used_n;
      const bool in_tile = (u32)(a64 >> 32) == 0u && av[k] - A_lo < A_span;
      bad = bad || (used && !in_tile);
      live[k] = used && in_tile;
      tl[k] = 0;
    }
#pragma unroll
    for (int i = 0; i < A_PRE; ++i)  // interior columns of the tile (W - 1 of them: one at C-1M)
      if (1 + i < Wc) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) tl[k] += av[k] >= A_in[i] ? 1 : 0;
      }
    for (int j = 1 + A_PRE; j < Wc; ++j) {  // wide tiles (sparse frames): the rest, one uniform load each
      const u32 A_j = thr[c0 + j];
#pragma unroll
      for (int k = 0; k < EPT; ++k) tl[k] += av[k] >= A_j ? 1 : 0;
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const u32 xk = (xw[k >> 1] >> ((k & 1) * 16)) & 0xffff, yk = (yw[k >> 1] >> ((k & 1) * 16)) & 0xffff;
      xl[k] = (int)xk - x_lo;
      fast[k] = live[k] && (u32)xl[k] < (u32)wx_eff && yk < (u32)tb.cam_h;
      smask |= live[k] && !fast[k] ? 1u << k : 0u;
    }
    }
    if (pass == 0) {
      XM_CSTAMP(3);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the LDS-direct band loads are tracked by vmcnt
      XM_CSTAMP(4);
      __syncthreads();                                   // bands (and the cleared slots) visible
      XM_CSTAMP(5);
    }
    if (on) {
    // branch-free: A1 + A2 out of the LDS bands; an event that is not in the LUT window reads the sentinel (yr < 0) and drops
    // out at xmd:23.  Three sweeps, so that the eight LDS round trips of each overlap.
    u32 l[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const u32 yk = (yw[k >> 1] >> ((k & 1) * 16)) & 0xffff;
      const u32* src = fast[k] ? lut_t + (__mul24(xl[k], tb.cam_h) + (int)yk) : &s_sentinel;
      l[k] = *src;
    }
    int slot[EPT], fu[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int yr = (int)(short)(l[k] >> 16);
      slot[k] = (u32)yr < (u32)(tb.xmap_h - 1) ? __mul24(tl[k], tb.xmap_h) + yr : -1;  // 0 <= yr < H - 1 (xmd:23); -1: dropped
      fu[k] = (int)xm_t[max(slot[k], 0)] - tb.x_offset;  // the frame column, = xr + disp (calib:300); no int16 wrap on this rig
    }
    if (all_in) {
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int disp = fu[k] - (int)(short)(l[k] & 0xffff);  // (xm_create has checked the range: xmd:27's wrap never triggers)
        const bool write = slot[k] >= 0 && disp >= 0;          // xmd:29
        n_in += write ? 1u : 0u;
        if (write) atomicMax(&slots[slot[k]], ((u32)(e0 + (VEC ? k : k * nthreads) + 1) << 16) | (u32)disp);
      }
    } else {
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int xr = (int)(short)(l[k] & 0xffff), yr = (int)(short)(l[k] >> 16);
        const int disp = fu[k] - xr;
        bool write = slot[k] >= 0 && disp >= 0;
        const bool in_frame = ((u32)fu[k] < (u32)tb.rect_w || (u32)(fu[k] + tb.rect_w) < (u32)tb.rect_w) && yr < tb.rect_h;
        n_oob += write && !in_frame ? 1u : 0u;  // NumPy IndexError (one negative wrap is legal)
        write = write && in_frame;
        n_in += write ? 1u : 0u;
        if (write) atomicMax(&slots[slot[k]], ((u32)(e0 + (VEC ? k : k * nthreads) + 1) << 16) | (u32)disp);
      }
    }
    // events outside the LUT window (x noise), one by one behind the fast path: the LUT entry from global memory, then the same
    // A2 + A3 -- the slot's ds_max orders them against the others whatever the order of processing.  x / y outside the camera
    // = map[y, x] IndexError in the reference (calib:279-280): dropped and counted
    while (__ballot(smask != 0)) {
      const bool act = smask != 0;
      const int ks = act ? __builtin_ctz(smask) : 0;
      smask &= smask - 1;
      u32 exw = xw[0], eyw = yw[0];  // (select chains: a dynamic index would put the arrays into scratch memory)
      int etl = tl[0];
#pragma unroll
      for (int q = 1; q < EPT / 2; ++q) {
        exw = (ks >> 1) == q ? xw[q] : exw;
        eyw = (ks >> 1) == q ? yw[q] : eyw;
      }
#pragma unroll
      for (int kk = 1; kk < EPT; ++kk) etl = ks == kk ? tl[kk] : etl;
      const u32 ex = (exw >> ((ks & 1) * 16)) & 0xffff, ey = (eyw >> ((ks & 1) * 16)) & 0xffff;
      const bool inside = act && ex < (u32)tb.cam_w && ey < (u32)tb.cam_h;
      n_oob += act && !inside ? 1u : 0u;
      const u32 le = inside ? tb.lut[__umul24(ex, (u32)tb.cam_h) + ey] : 0x80000000u;
      const int xr = (int)(short)(le & 0xffff), yr = (int)(short)(le >> 16);
      const bool yok = (u32)yr < (u32)(tb.xmap_h - 1);
      const int sl = yok ? __mul24(etl, tb.xmap_h) + yr : 0;
      const int fue = (int)xm_t[sl] - tb.x_offset, disp = fue - xr;
      bool write = yok && disp >= 0;
      const bool in_frame = ((u32)fue < (u32)tb.rect_w || (u32)(fue + tb.rect_w) < (u32)tb.rect_w) && yr < tb.rect_h;
      n_oob += write && !in_frame ? 1u : 0u;
      write = write && in_frame;
      n_in += write ? 1u : 0u;
      if (write) atomicMax(&slots[sl], ((u32)(e0 + (VEC ? ks : ks * nthreads) + 1) << 16) | (u32)disp);
    }
    }
  }
  XM_CSTAMP(6);
  if (n_pass == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the X-map band is needed by the flush
  }
  if (__ballot(bad) && lane == 0) {
    if (device_redo) {  // inside a hipGraph: the redo kernels behind this one in the graph look at this word (frame_attempt_failed)
      st->pad[1] = tag;
    } else {
      __hip_atomic_fetch_add(&st->cnt[parity][blk % CNT_SLOTS][CNT_UNSORTED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&st->unsorted_sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (u32* hf = st->host_flags) host_flag_store(hf, tag);
    }
  }
  {  // both counters in one word (a lane counts <= 8 events per pass, a tile holds <= 65527 events)
    u32 cnt2 = n_in | (n_oob << 16);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt2 += __shfl_xor(cnt2, o, 64);
    if (lane == 0) {
      if (cnt2 & 0xffffu) atomicAdd(&s_in, cnt2 & 0xffffu);
      if (cnt2 >> 16) atomicAdd(&s_oob, cnt2 >> 16);
    }
  }
  __syncthreads();
  XM_CSTAMP(7);

  // ---- 5. flush: every live slot of the tile -> its frame cell, winners and empties alike (plain 2-byte stores; lanes walk
  //         consecutive rows of one time column = consecutive rows of one frame column where the X-map is smooth).
  //         live: xp - x_offset >= xr_min (cols_cell); with xr_min >= 0 and rect_h >= xmap_h - 1 (the usual rig) neither the
  //         negative wrap nor the row test can trigger: the lean variant.
  {
    const bool lean = xr_min >= 0 && tb.rect_h >= tb.xmap_h - 1;  // uniform
    // (two rows per lane with one 4-byte store where both map to the same frame column was tried: no faster)
    constexpr int FL = 3;  // (2640 slots at C-1M: two sweeps of 512 x 3)
    const int per = tb.xmap_h;
    int dr = nthreads, r_i = tid;  // nthreads % per, tid % per (a few subtractions on small rigs, none on large ones)
    while (dr >= per) dr -= per;
    while (r_i >= per) r_i -= per;
    for (int i0 = tid; i0 < nslots; i0 += FL * nthreads) {
      u32 v[FL];
      int xv[FL], rs[FL];
#pragma unroll
      for (int j = 0; j < FL; ++j) {
        const int i = min(i0 + j * nthreads, nslots - 1);
        v[j] = slots[i];
        xv[j] = (int)xm_t[i];
        rs[j] = r_i;
        r_i += dr;
        if (r_i >= per) r_i -= per;
      }
      if (lean) {
#pragma unroll
        for (int j = 0; j < FL; ++j) {
          const int fu = xv[j] - tb.x_offset;
          if (i0 + j * nthreads < nslots && rs[j] < tb.xmap_h - 1 && fu >= xr_min && fu < tb.rect_w && !XM_CABL(4))
            frame16[__umul24((u32)fu, (u32)tb.rect_h) + (u32)rs[j]] = (uint16_t)(v[j] & 0xffffu);
        }
      } else {
#pragma unroll
        for (int j = 0; j < FL; ++j) {
          u32 cell;
          if (i0 + j * nthreads < nslots && rs[j] < tb.xmap_h - 1 && cols_cell(tb, xv[j], rs[j], xr_min, cell))
            frame16[cell] = (uint16_t)(v[j] & 0xffffu);
        }
      }
    }
  }
  XM_CSTAMP(8);
  if (tid == 0) {
    XM_GLOBAL u32* c = st->cnt[parity][blk % CNT_SLOTS];
    if (s_in) __hip_atomic_fetch_add(&c[CNT_INLIER], s_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (s_oob) __hip_atomic_fetch_add(&c[CNT_OOB], s_oob, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

#ifndef XM_COLS_MAX_THREADS
#define XM_COLS_MAX_THREADS 512
#endif
constexpr int COLS_MAX_THREADS = XM_COLS_MAX_THREADS;
#ifndef XM_COLS_WAVES_PER_EU
#define XM_COLS_WAVES_PER_EU 6
#endif

template <bool AOS, bool VEC>
__global__ __launch_bounds__(COLS_MAX_THREADS, XM_COLS_WAVES_PER_EU) void k_scatter_cols(
    const uint16_t* __restrict__ xs, const uint16_t* __restrict__ ys, const long long* __restrict__ ts, const uint4* __restrict__ aos,
    u32 n, DevTables tb, SlotState* st, uint16_t* __restrict__ frame16, int W, int w_x, int xr_min, int flags) {
  {  // every kernel argument in one scalar round trip (see k_scatter_tiled); never true
    const u64 pp = (u64)xs | (u64)ys | (u64)ts | (u64)aos | (u64)tb.lut | (u64)tb.xmap | (u64)st | (u64)frame16;
    const int pi = tb.cam_w | tb.cam_h | tb.xmap_w | tb.xmap_h | tb.t_px_scale | tb.x_offset | tb.rect_w | tb.rect_h | W | w_x;
    if ((long long)(pp | (u64)(long long)pi) < 0) return;
  }
  scatter_cols_body<AOS, VEC>((gp_u16)xs, (gp_u16)ys, (gp_i64)ts, (gp_u4)aos, n, tb, (gp_state)st, (XM_GLOBAL uint16_t*)frame16, W,
                              w_x, xr_min, blockIdx.x, gridDim.x, flags);
}

// multi-frame launch: grid = (tiles, frames); FrameDesc.key_frame points at the frame's u16 disparity frame (+ bounds)
template <bool AOS, bool VEC>
__global__ __launch_bounds__(COLS_MAX_THREADS, XM_COLS_WAVES_PER_EU) void k_scatter_cols_batch(const FrameDesc* __restrict__ descs,
                                                                                                DevTables tb, int W, int w_x,
                                                                                                int xr_min, int flags) {
  const FrameDesc d = descs[blockIdx.y];  // block-uniform: scalar loads
  const bool ext = flags & COLS_F_EXT_EXTREMA;
  if (!d.valid || (d.n == 0 && !ext)) return;  // (the host sends frames without events down the general path)
  scatter_cols_body<AOS, VEC>((gp_u16)d.x, (gp_u16)d.y, (gp_i64)d.t, (gp_u4)d.aos, (u32)d.n, tb, (gp_state)d.st,
                              (XM_GLOBAL uint16_t*)d.key_frame, W, w_x, xr_min, blockIdx.x, gridDim.x, flags,
                              ext ? (gp_i64)d.p : nullptr);
}

// ---- shards on the column tiles (SURVEY.md 8(e) / BASELINE configs[3]) --------------------------------------------------------
// A shard = a contiguous index range of the frame's time-sorted stream = whole time columns, except that its LAST column may go on
// in the next shard.  Every cell of the u16 frame has one (row, column) pair (injective rigs), so once every column is processed
// by ONE rank the ranks' frames are disjoint and merge by a plain SUM -- 2 bytes per cell on the wire, no packed keys, no
// atomics, no extrema pass.  To get there each rank (but the last) leaves the events of its last column to its successor:
//   k_shard_cols_pack     send buffer <- {t[0], t[m-1], m | the shard's last min(m, cap) events}  (nothing depends on other ranks)
//   (ONE all-gather of the send buffers: it carries the frame's extrema and every predecessor's last events)
//   k_shard_cols_prepare  the frame's extrema from the headers (under the sorted assumption that K1 verifies per event); the
//                         first event of the own last column (256-ary search) = where the own part ends; the events of the
//                         PREDECESSOR's last column out of its buffer into the headroom in front of the own events (padded to
//                         an 8-aligned start with copies of the first event: the 16-byte loads of K1); the piece's FrameDesc
//   k_cols_bounds_batch / k_scatter_cols_batch with COLS_F_EXT_EXTREMA on that piece (tiles of columns the piece does not hold
//   write their zeros), then the SUM all-reduce and K2.
// A piece that cannot be handled this way (a shard without events or inside ONE column, a last column of more than cap events, a
// frame of >= 2^32 us, events out of order) raises the slot's sticky flag: the caller redoes the frame with the packed keys.
struct ShardColsHeader {
  long long t_first, t_last;
  long long m;       // events of the shard
  long long count;   // events in the buffer: the shard's last min(m, cap)
};
static_assert(sizeof(ShardColsHeader) == 32, "ShardColsHeader layout");

__global__ __launch_bounds__(256) void k_shard_cols_pack(const uint16_t* __restrict__ x, const uint16_t* __restrict__ y,
                                                         const long long* __restrict__ t, u64 m, unsigned char* __restrict__ send, u64 cap) {
  ShardColsHeader* hdr = reinterpret_cast<ShardColsHeader*>(send);
  uint16_t* sx = reinterpret_cast<uint16_t*>(send + sizeof(ShardColsHeader));
  uint16_t* sy = sx + cap;
  long long* stt = reinterpret_cast<long long*>(sy + cap);
  const u64 cnt = m < cap ? m : cap, from = m - cnt;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    hdr->t_first = m ? t[0] : 0;
    hdr->t_last = m ? t[m - 1] : 0;
    hdr->m = (long long)m;
    hdr->count = (long long)cnt;
  }
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (u64)gridDim.x * 256) {
    sx[i] = x[from + i];
    sy[i] = y[from + i];
    stt[i] = t[from + i];
  }
}

// first index in [0, n) with (u64)(t[i] - tmin) >= A, n if none: 256-ary search by one block (every thread calls it; a stream that
// is not sorted gets some deterministic answer and K1 objects)
__device__ inline u64 shard_lower_bound(const long long* __restrict__ t, u64 n, long long tmin, u32 A, u64* s_rng, int* s_first) {
  const int tid = threadIdx.x;
  if (n == 0) return 0;
  if (tid == 0) {
    s_rng[0] = 0;  // the answer lies in [lo, hi]; hi == n means "none so far"
    s_rng[1] = n;
  }
  __syncthreads();
  for (int round = 0; round < 9; ++round) {
    const u64 lo = s_rng[0], hi = s_rng[1];
    __syncthreads();
    if (hi <= lo) break;
    const u64 width = hi - lo, stride = (width + 255) / 256;
    const u64 q = lo + (u64)tid * stride;
    const bool ge = q < hi && (u64)(t[q] - tmin) >= (u64)A;
    const u64 bal = __ballot(ge);
    if ((tid & 63) == 0) s_first[tid >> 6] = bal ? (tid & ~63) + (int)__builtin_ctzll(bal) : -1;
    __syncthreads();
    if (tid == 0) {
      int f = -1;
      for (int w = 0; w < 4 && f < 0; ++w) f = s_first[w];
      if (f < 0) {  // every probe lies in front of the boundary
        s_rng[0] = lo + ((width - 1) / stride) * stride + 1;
      } else {
        const u64 qf = lo + (u64)f * stride;
        s_rng[1] = qf;
        if (f > 0) s_rng[0] = qf - stride + 1;
      }
    }
    __syncthreads();
  }
  return s_rng[1];
}

// x / y / t point at the shard's own m events and have >= cap + 8 events of writable headroom in front of them (and 8 behind)
__global__ __launch_bounds__(256) void k_shard_cols_prepare(uint16_t* x, uint16_t* y, long long* t, u64 m,
                                                            const unsigned char* __restrict__ gathered, u64 send_bytes, int rank, int world,
                                                            DevTables tb, u64 cap, long long* mm, uint16_t* frame16, SlotState* st,
                                                            FrameDesc* desc) {
  __shared__ u64 s_rng[2];
  __shared__ int s_first[4];
  __shared__ long long s_mm[2];
  __shared__ int s_fail;
  const int tid = threadIdx.x;
  // the slot's bookkeeping starts over with every frame (what k_reset_slot does; the sticky flag stays)
  if (tid == 0) {
    st->tag_a = 0;
    st->tag_b = 0;
    st->pad[1] = 0;
    long long lo = 0x7fffffffffffffffll, hi = (long long)0x8000000000000000ull;
    int fail = 0;
    for (int r = 0; r < world; ++r) {
      const ShardColsHeader* h = reinterpret_cast<const ShardColsHeader*>(gathered + (u64)r * send_bytes);
      if (h->m <= 0) {
        fail = 1;  // a shard without events breaks the chain of columns
        continue;
      }
      lo = h->t_first < lo ? h->t_first : lo;
      hi = h->t_last > hi ? h->t_last : hi;
    }
    if (hi < lo) {
      lo = hi = 0;
      fail = 1;
    }
    if ((u64)(hi - lo) >= 0xffffffffull) fail = 1;
    s_mm[0] = lo;
    s_mm[1] = hi;
    s_fail = fail;
    mm[0] = lo;
    mm[1] = -hi;
  }
  for (int i = tid; i < 2 * MM_SLOTS; i += 256) {
    st->mm[i / MM_SLOTS][i % MM_SLOTS][0] = MM_INIT_MIN;
    st->mm[i / MM_SLOTS][i % MM_SLOTS][1] = MM_INIT_MAX;
  }
  for (int i = tid; i < 2 * CNT_SLOTS * CNT_STRIDE; i += 256) (&st->cnt[0][0][0])[i] = 0;
  __syncthreads();
  const long long tmin = s_mm[0], tmax = s_mm[1];
  bool fail = s_fail != 0;
  u64 own_n = m, cnt = 0, j0 = 0;
  const ShardColsHeader* ph = rank > 0 ? reinterpret_cast<const ShardColsHeader*>(gathered + (u64)(rank - 1) * send_bytes) : nullptr;
  if (!fail) {
    const TimeNorm<long long> tn(tmin, tmax, tb.t_px_scale);
    const u32 span = (u32)(tmax - tmin);
    if (rank < world - 1) {  // the own last column goes to the successor
      const u32 A = cols_threshold(tn, tmin, span, tn.column(t[m - 1]), tb.t_px_scale);
      own_n = shard_lower_bound(t, m, tmin, A, s_rng, s_first);
      if (own_n == 0 || m - own_n > cap) fail = true;  // the whole shard lies in one column / its last column does not fit the buffer
    }
    if (ph) {  // the predecessor's last column: a suffix of its buffer
      const u64 pc = (u64)ph->count;
      const long long* pt = reinterpret_cast<const long long*>(reinterpret_cast<const unsigned char*>(ph) + sizeof(ShardColsHeader) + 4 * cap);
      const u32 A = cols_threshold(tn, tmin, span, tn.column(ph->t_last), tb.t_px_scale);
      __syncthreads();
      j0 = shard_lower_bound(pt, pc, tmin, A, s_rng, s_first);
      if (j0 == 0) fail = true;  // (its last column may start in front of what the buffer holds)
      cnt = pc - j0;
    }
  }
  if (fail) {
    own_n = m;
    cnt = 0;
  }
  // the piece: [pad: copies of its first event, up to an 8-aligned start | the predecessor's last column | the own part]
  const u64 pad = (8 - (cnt & 7)) & 7;  // (the own events start 8-aligned: cap is a multiple of 8 and so is the headroom)
  if (ph && cnt) {
    const uint16_t* px = reinterpret_cast<const uint16_t*>(reinterpret_cast<const unsigned char*>(ph) + sizeof(ShardColsHeader));
    const uint16_t* py = px + cap;
    const long long* pt = reinterpret_cast<const long long*>(py + cap);
    for (u64 i = tid; i < cnt; i += 256) {
      x[(long long)i - (long long)cnt] = px[j0 + i];
      y[(long long)i - (long long)cnt] = py[j0 + i];
      t[(long long)i - (long long)cnt] = pt[j0 + i];
    }
    if ((u64)tid < pad) {
      x[-(long long)(cnt + pad) + tid] = px[j0];
      y[-(long long)(cnt + pad) + tid] = py[j0];
      t[-(long long)(cnt + pad) + tid] = pt[j0];
    }
  }
  if (tid == 0) {
    if (fail) atomicAdd(&st->unsorted_sticky, 1u);
    desc->x = x - (cnt + pad);
    desc->y = y - (cnt + pad);
    desc->t = t - (cnt + pad);
    desc->p = reinterpret_cast<const int16_t*>(mm);  // COLS_F_EXT_EXTREMA: the frame's {tmin, -tmax}
    desc->aos = nullptr;
    desc->n = pad + cnt + own_n;
    desc->key_frame = reinterpret_cast<u64*>(frame16);
    desc->st = st;
    desc->depth = nullptr;
    desc->bgr = nullptr;
    desc->valid = 1;
    desc->pad = (u32)pad;  // events [0, pad) of the piece are filler: the first tile starts behind them
  }
}

}  // namespace xm
