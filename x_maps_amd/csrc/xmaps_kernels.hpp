// xmaps_kernels.hpp -- gfx950 (MI355X / CDNA4) device code for the X-maps hot path.
//
// Three kernels per frame (no memset, no host round trip):
//   K0 k_minmax   : frame extrema of t (x_maps_disparity.py:12-13)  -- pure streaming reduction
//   K1 k_scatter  : per event: rectify-LUT gather (cam_proj_calibration.py:277-281) -> FP64 time
//                   normalise + rint (x_maps_disparity.py:16-19) -> X-map gather (:25) -> int16
//                   disparity + inlier masks (:23-29) -> last-writer-wins scatter
//                   (cam_proj_calibration.py:299-303 / 312-317) as ONE 64-bit atomic max
//   K2 k_frame_*  : per output pixel: 7x7 max (cv2.dilate) composed with the nearest remap
//                   (disp_to_depth.py:86-95) -> depth (disp_to_depth.py:46-63) -> u8 normalise
//                   (:7-21) -> Turbo BGR + white mask (:24-43)
//
// Last-writer-wins without a clear: every cell of the disparity frame holds a packed key
//   [63]=0 | frame tag:19 | event index:28 | disparity:16
// and K1 does atomic max.  Inside a frame the largest event index wins (= NumPy's fancy-assignment
// order); keys of older frames carry a smaller tag, lose every max and are ignored by K2, so the
// 9-56 MB frame is never zero-filled.  The tag lives in device memory (SlotState) so that the same
// launches can be replayed from a hipGraph.
//
// This is gather/scatter + integer work: no MFMA.  What matters is coalesced event reads, L2-resident
// tables, fire-and-forget atomics and enough waves in flight to hide three dependent memory latencies.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

namespace xm {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int KEY_IDX_SHIFT = 16;
constexpr int KEY_TAG_SHIFT = 44;
constexpr u32 KEY_MAX_TAG = (1u << 19) - 1;
constexpr int MM_SLOTS = 32;   // spread the min/max atomics over 32 addresses: one contended word retires only
                               // ~88 atomics/us on this chip (8 slots measured +3 us on K0)
constexpr int CNT_SLOTS = 64;  // same for the counters
constexpr int BLOCK = 256;

enum { CNT_USED = 0, CNT_INLIER = 1, CNT_OOB = 2, CNT_UNSORTED = 3, CNT_STRIDE = 4 };

// HBM layouts (built once in xm_create).  The scan axis is the SLOW axis of every table an event touches:
// events arrive time-sorted and the projector scans x-slow, so the events of one thread block sit in a band of
// a few camera columns / a few X-map time columns / a few frame columns.  Column-major tables make that band a
// handful of contiguous runs -> coalesced tile loads into LDS and coalesced flushes out of it.
struct DevTables {
  const u32* lut;       // [cam_w][cam_h]   TRANSPOSED  (u16(yr) << 16) | u16(xr)
  const int16_t* xmap;  // [xmap_w][xmap_h] TRANSPOSED  X-map, time column major
  const u32* pmap;      // [proj_h][proj_w] row-major   (u16(my) << 16) | u16(mx)
  const uint2* dlut;    // [65536] per integer disparity: {f32 bits of depth, BGR word} = disparity_pixel(d) (A5-A7)
  const int4* k2_tiles; // [tiles_y][tiles_x] {bx, by, cols, rows_p} of every K2 tile's key-frame patch (cols = 0: none of
                        //   its pixels maps into the frame; cols < 0: patch too large for LDS -> generic path), precomputed
  const u32* k2_pix;    // [proj_h][proj_w] offset of the pixel's 7-tap column run inside its tile's LDS patch, ~0u = the
                        //   pixel maps outside the frame (BORDER_CONSTANT 0)
  const int4* k2_tiles1;  // the same two tables for K2's one-pixel-per-thread geometry (16 x 16 tiles: lone frames, whose
  const u32* k2_pix1;     //   launch is too small to fill the chip with 32 x 16 tiles; see frame_proj_tiled_body)
  int cam_w, cam_h, proj_w, proj_h, rect_w, rect_h, xmap_w, xmap_h;
  int x_offset, t_px_scale;
  double p03;
  float z_near, z_far;
  // owner tiles (xmaps_k1own.hpp; rigs whose (row, time column) -> cell map is not injective): the X-map once more with the
  // distance to the cell's owner column in the top bits; per tile {columns of its cell band, first extra, extras}; per (tile,
  // row) the band's first frame column and the mask of the band cells the tile owns (one u32); cells outside the
  // band ("extras") have a slot index in xmap_extra (at their owner pair) and their frame cell in own_extra_cells.  The rows
  // the rectify LUT can reach: own_hr rows from own_r_lo (a multiple of 8) on, padded to own_hrp (a multiple of 8)
  const uint16_t* xmap_own;     // [xmap_w][xmap_h]  xp | delta << 13, 0 = undefined
  const uint16_t* xmap_extra;   // [xmap_w][xmap_h]  extra slot + 1 at the owner pair of a cell outside its tile's band, else 0
  const int4* own_tiles;        // [tiles] {band columns, first extra, extras, 0}
  const u32* own_bm;            // [tiles][own_tab_words]  the tile's band table (own_setup): band positions and ownership per row / per 8-row group
  const u32* own_extra_cells;   // [extras] cell index in the (sheared) u16 frame
  int own_r_lo, own_hr, own_hrp, own_nxs_max, own_extra_max;
  int own_grouped;  // 1: a tile owns whole 8-row pieces of a frame column (own_plan), 0: any cells of a row
  int own_rp;  // rows per pass of a tile's LDS slots (a multiple of 8; own_hrp = one pass): see scatter_own_body
  // the plain u16 disparity frame of the column / owner tiles is sheared by whole columns per 8-row group: cell (x, row) lives
  // in frame column x + shear_bias + ((row >> 3) * shear_m >> 12); the frame has rect_w + shear_extra columns.  All 0 unless
  // the rig's X-map is slanted (xm_create fits shear_m)
  int shear_m, shear_bias, shear_extra;
};

__host__ __device__ inline size_t frame16_cells(const DevTables& tb) { return (size_t)(tb.rect_w + tb.shear_extra) * (size_t)tb.rect_h; }
// column of cell (x, row) in the u16 frame
__host__ __device__ inline int frame16_col(const DevTables& tb, int x, int row) { return x + tb.shear_bias + (((row >> 3) * tb.shear_m) >> 12); }

// Per-slot device state.  tag_a is written by K0 (block 0) and read by K1/K2; tag_b is written by K1
// (block 0) and read by K0 -- so no kernel reads a word that one of its own blocks is writing.
struct SlotState {
  u32 tag_a;
  u32 tag_b;
  u32 pad[2];
  u64 mm[2][MM_SLOTS][2];               // [parity][slot]{min, max} in order-preserving u64 encoding
  u32 cnt[2][CNT_SLOTS][CNT_STRIDE];    // [parity][slot]{used, inliers, index errors, events outside [t[0], t[n-1]]}
  u32 unsorted_sticky;                  // time-sorted mode: frames whose declaration did not hold (read by xm_sync)
  u32 pad2;
  // XM_FLAG_TRY_SORTED: two words of pinned host memory the kernels report to without a host round trip --
  // [0] = tag of the last frame whose (t[0], t[n-1]) shortcut did NOT hold (written by K1), [1] = tag of the last frame whose
  // K2 has started, i.e. whose K1 verdict is final.  NULL when the mode is off.
  u32* host_flags;
};
static_assert(sizeof(SlotState) % 16 == 0, "SlotState array stride");

// One frame of a MULTI-FRAME launch (grid = frames x tiles), in device memory.  Written by the host (xm_process_batch, the
// hipGraph batch) or by the ingest kernels (device-side frame segmentation: the frame's event range never visits the
// host).  valid == 0: every kernel of the frame exits at once (no frame was cut).  n == 0 with valid != 0: a defined
// empty frame (tags advance, outputs are written empty).
struct FrameDesc {
  const uint16_t* x;
  const uint16_t* y;
  const void* t;
  const int16_t* p;
  const uint4* aos;
  u64 n;
  u64* key_frame;
  SlotState* st;
  float* depth;
  uint8_t* bgr;
  u32 valid;
  u32 pad;
};
static_assert(sizeof(FrameDesc) == 88, "FrameDesc layout");

// Conditional frames of a captured batch (hipGraph): no host is at hand there to redo a frame whose column-tile attempt
// failed (xmaps_k1cols.hpp), so the graph carries BOTH paths and the kernels decide per frame on the device.  A failing
// tile leaves the frame's tag in SlotState.pad[1]; COND = 1 kernels run a frame only if its attempt failed, COND = 2 only
// if it held, COND = 0 always.  (tag_a holds the frame's tag from the attempt's K1 on, and K0 of the redo recomputes the
// very same value from tag_b, which only a K2 advances.)
__device__ inline bool frame_attempt_failed(const SlotState* st) { return st->pad[1] == st->tag_a; }
template <int COND> __device__ inline bool frame_skipped(const SlotState* st) {
  if constexpr (COND == 0) return false;
  else return frame_attempt_failed(st) != (COND == 1);
}

// device -> pinned host memory, visible to the host when the kernel has finished
__device__ inline void host_flag_store(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// ---- order-preserving u64 encodings so that one pair of unsigned atomics serves every t dtype ------
template <typename T> struct TimeCodec;
template <> struct TimeCodec<long long> {
  static __host__ __device__ u64 enc(long long v) { return (u64)v ^ 0x8000000000000000ull; }
  static __host__ __device__ long long dec(u64 u) { return (long long)(u ^ 0x8000000000000000ull); }
};
template <> struct TimeCodec<double> {
  static __host__ __device__ u64 enc(double v) {
    u64 b;
    __builtin_memcpy(&b, &v, 8);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
  }
  static __host__ __device__ double dec(u64 u) {
    u64 b = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u;
    double v;
    __builtin_memcpy(&v, &b, 8);
    return v;
  }
};
template <> struct TimeCodec<float> {  // f32 -> f64 is exact and monotone
  static __host__ __device__ u64 enc(float v) { return TimeCodec<double>::enc((double)v); }
  static __host__ __device__ float dec(u64 u) { return (float)TimeCodec<double>::dec(u); }
};

// Dirty-line flags of the projector-view key frame: one byte per 128-byte line (16 cells).  K1 stores the frame's tag
// byte for every line it writes a key into; K2 only fetches lines whose flag carries the current tag byte (at C-1M only
// 46 % of the lines are dirty, so K2 skips half of its 34 MB read).  Plain idempotent stores, no clearing: a stale flag
// (tags repeat every 255 frames) is only a false positive -- the line is fetched and its keys' full tags decide.
__host__ __device__ inline unsigned char dirty_byte(u32 tag) { return (unsigned char)(tag % 255u + 1u); }

// ---- compact (32-bit) key frame of the verified-sorted projector-view path ------------------------------------------------
//   tag4:4 | tile:16 | disparity:12        (tag4 = tag % 15 + 1; 0 = cleared cell)
// Half the bytes per cell means twice as many winners per 64-byte L2 atomic request and half of K2's key-frame read: K1
// at full occupancy is bound by the chip's L2 atomic rate (~23 G requests/s measured: profiles/r02*_pmc.md), K2 by the
// read.  The order field is the TILE index, not the event index: inside a tile the last writer is resolved exactly in LDS
// (slot value = local index | disparity), across tiles a higher tile = later events.  That is exact as long as EVERY
// event of the tile goes through the LDS slots: an event whose time column falls outside the tile's LDS window cannot, so it
// marks the frame as failed -- the same flag, and the same automatic redo on the 64-bit general path, as an event outside
// [t[0], t[n-1]].  Events outside the LUT window only (x noise) fetch their LUT entry from global memory and then use
// the slots like everybody else.  Preconditions checked by the host (xm_create): projector view, rect_h % 4 == 0, every
// possible disparity < 4096, tiles per frame < 65536; the 4-bit tag is kept unambiguous by clearing the frame at least
// every 15 frames of the slot (9.3 MB memset per 15 frames at C-1M).
// Camera view (VIEW == 1 with KEY32): the cell is the event's own pixel, written by events of ANY tile, so the order field is the
// event itself: key = (event index + 1) << 12 | disparity -- exact for every frame of < 2^20 events whatever their order, strays
// included; no tag: the frame kernel, which reads every pixel of the 1.2 MB frame exactly once, zeroes what it has read.
constexpr u32 KEY32_DISP_BITS = 12, KEY32_TILE_BITS = 16, CAM32_MAX_EVENTS = (1u << 20) - 1u;
__host__ __device__ inline u32 key32_tag(u32 tag) { return (tag % 15u + 1u) << 28; }
__device__ inline uint16_t key_disp32(u32 k, u32 tag4) { return (k & 0xf0000000u) == tag4 ? (uint16_t)(k & 0xfffu) : (uint16_t)0; }

constexpr u64 MM_INIT_MIN = ~0ull;
constexpr u64 MM_INIT_MAX = 0ull;

// ---- t -> X-map column, bit-exact with NumPy (x_maps_disparity.py:16-19) ---------------------------
// int64: (t - tmin) and (tmax - tmin) are exact int64, both converted to f64, IEEE divide, multiply by
// S, round-half-even.  Compiled with -ffp-contract=off so nothing is fused.
template <typename T> struct TimeNorm;
template <> struct TimeNorm<long long> {
  long long tmin;
  double den, scale, rs;
  bool degenerate, fast_frame;
  __device__ TimeNorm(long long lo, long long hi, int S)
      : tmin(lo), den((double)(hi - lo)), scale((double)S), degenerate(hi == lo) {
    // a frame spans microseconds: (hi - lo) < 2^32 always holds in practice; everything else takes the exact path
    fast_frame = (u64)(hi - lo) <= 0xffffffffull && !degenerate;
    rs = (1.0 / den) * scale;
  }
  // Reference: rint(fl(fl(a / den) * S)), a = t - tmin.  Fast value: e = fl(a * fl(fl(1/den) * S)) differs from the
  // reference's product by < 4 ulp (< 2e-12 for columns <= 32767); whenever e is further than 1e-6 from a rounding
  // boundary (x.5) both round to the same integer, so rint(e) IS the reference result.  Closer than that (exact ties such
  // as golden g1d_rint_ties land here) the IEEE divide of column_exact decides.
  // Branch-free and built from full-rate FP64 adds only (K1 as a single launch is bound by this dependent chain; the
  // conversions and v_rndne_f64 are quarter rate): u32 -> double and double -> nearest-even integer both go through the
  // 2^52 trick -- bits(2^52) | a IS 2^52 + a, and the low word of fl(e + 2^52) IS rint(e) for 0 <= e < 2^32.
  // `ok` = the value may be used; callers OR the failures of a batch together and take ONE rare branch.
  __device__ int column_fast(long long t, bool& ok) const {
    constexpr double M = 4503599627370496.0;  // 2^52
    const u64 a = (u64)(t - tmin);
    const double ad = __hiloint2double(0x43300000, (int)(u32)a) - M;  // (double)(u32)a, exact
    const double e = ad * rs;
    const double m = e + M;  // low word = rint(e), ties to even
    const double d = e - (m - M);  // e - rint(e), exact (Sterbenz)
    ok = ((u32)(a >> 32) == 0u) & (fabs(d) < 0.5 - 1e-6);
    return (int)(short)__double2loint(m);
  }
  __device__ int column_exact(long long t) const {
    if (degenerate) return 0;  // 0/0 = NaN -> int16 cast = 0 (what NumPy yields on x86-64)
    const double tn = (double)(t - tmin) / den;
    return (int)(short)(int)rint(tn * scale);
  }
  __device__ int column(long long t) const {
    bool ok;
    const int c = column_fast(t, ok);
    return __builtin_expect(ok && fast_frame, 1) ? c : column_exact(t);
  }
  // N columns at once: straight-line fast values (N independent chains for the scheduler to interleave), one rare branch
  template <int N> __device__ void columns(const long long (&t)[N], int (&col)[N]) const {
    u32 redo = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      bool ok;
      col[k] = column_fast(t[k], ok);
      redo |= ok ? 0u : (1u << k);
    }
    if (!fast_frame) redo = (1u << N) - 1;
    if (__builtin_expect(redo != 0, 0)) {
#pragma unroll
      for (int k = 0; k < N; ++k)
        if ((redo >> k) & 1) col[k] = column_exact(t[k]);
    }
  }
};
template <> struct TimeNorm<double> {
  double tmin, den, scale;
  bool degenerate;
  __device__ TimeNorm(double lo, double hi, int S) : tmin(lo), den(hi - lo), scale((double)S), degenerate(hi == lo) {}
  __device__ int column(double t) const {
    if (degenerate) return 0;
    double tn = (t - tmin) / den;
    return (int)(short)(int)rint(tn * scale);
  }
  template <int N> __device__ void columns(const double (&t)[N], int (&col)[N]) const {
#pragma unroll
    for (int k = 0; k < N; ++k) col[k] = column(t[k]);
  }
};
template <> struct TimeNorm<float> {  // eval caller with an f32 time surface: NumPy stays in f32
  float tmin, den, scale;
  bool degenerate;
  __device__ TimeNorm(float lo, float hi, int S) : tmin(lo), den(hi - lo), scale((float)S), degenerate(hi == lo) {}
  __device__ int column(float t) const {
    if (degenerate) return 0;
    float tn = (t - tmin) / den;
    return (int)(short)(int)rintf(tn * scale);
  }
  template <int N> __device__ void columns(const float (&t)[N], int (&col)[N]) const {
#pragma unroll
    for (int k = 0; k < N; ++k) col[k] = column(t[k]);
  }
};

// blockIdx -> work item such that the blocks that land on one XCD (blockIdx % 8 under round-robin dispatch) own a
// contiguous range of items.  Bijective for any grid size.
constexpr u32 N_XCD = 8;
__device__ inline u32 xcd_contiguous(u32 b, u32 nb) {
  const u32 xcd = b % N_XCD, j = b / N_XCD, q = nb / N_XCD, r = nb % N_XCD;
  return xcd * q + (xcd < r ? xcd : r) + j;
}

// The same for frame `frame` of a (nb, frames) grid: workgroups are dealt to the XCDs in LINEAR order, so the frame's block b sits on
// XCD (frame * nb + b) % 8 -- with nb % 8 != 0 every frame starts on another XCD, and xcd_contiguous(b, nb) would hand every XCD
// every table slice over the frames of a group.  XCD x takes the x-th contiguous run of the frame's items, whatever the phase
// (the runs' lengths move by one item between frames).  Bijective.
__device__ inline u32 xcd_contiguous_in_frame(u32 b, u32 nb, u32 frame) {
  const u32 s = (frame * nb) % N_XCD, bv = b + s, xcd = bv % N_XCD, end = nb + s;  // the frame's blocks: virtual indices [s, end)
  u32 start = 0;
  for (u32 x = 0; x < xcd; ++x) {
    const u32 first = s + ((x + N_XCD - s) % N_XCD);
    start += first < end ? (end - 1 - first) / N_XCD + 1 : 0;
  }
  const u32 first = s + ((xcd + N_XCD - s) % N_XCD);
  return start + (bv - first) / N_XCD;
}

// ---- wave helpers (wave = 64 lanes) ------------------------------------------------------------------
__device__ inline u64 wave_min_u64(u64 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    u64 w = __shfl_xor(v, o, 64);
    v = w < v ? w : v;
  }
  return v;
}
__device__ inline u64 wave_max_u64(u64 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    u64 w = __shfl_xor(v, o, 64);
    v = w > v ? w : v;
  }
  return v;
}

// frame extrema as written by K0: every wave reduces the MM_SLOTS partials itself (128 B, L2-hot) and broadcasts
// the result through SGPRs (readfirstlane), so everything derived from it is wave-uniform
__device__ inline u64 uniform_u64(u64 v) {
  const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
  return ((u64)hi << 32) | lo;
}
__device__ inline void load_frame_minmax(const SlotState* st, u32 parity, u64& lo, u64& hi) {
  const int lane = threadIdx.x & 63;
  u64 a = MM_INIT_MIN, b = MM_INIT_MAX;
  if (lane < MM_SLOTS) {
    a = st->mm[parity][lane][0];
    b = st->mm[parity][lane][1];
  }
#pragma unroll
  for (int o = MM_SLOTS / 2; o > 0; o >>= 1) {
    const u64 a2 = __shfl_xor(a, o, 64), b2 = __shfl_xor(b, o, 64);
    a = a2 < a ? a2 : a;
    b = b2 > b ? b2 : b;
  }
  lo = uniform_u64(a);
  hi = uniform_u64(b);
}

// =====================================================================================================
// K0: min / max of t over the frame's events (those with p == 1 when a polarity column is given).
//   SoA : t[n] (+ p[n]);  VEC = events per 16-byte load of int64 t (2) or scalar (1)
//   AoS : EventCD records, one 16-byte load per event
// Also: advances the slot's frame tag (block 0) and counts the used events.
// =====================================================================================================
#ifndef XM_K0_UN
#define XM_K0_UN 8
#endif
constexpr int K0_UN = XM_K0_UN;  // 16-byte loads of t in flight per thread (vector path)
template <typename T, bool AOS, bool HAS_P, int VEC>
__device__ __forceinline__ void minmax_body(const T* __restrict__ t, const int16_t* __restrict__ p,
                                            const uint4* __restrict__ aos, u64 n, SlotState* st, u32 tag_override,
                                            const u32 blk, const u32 nblk) {
  // the frame tag is only needed for the final atomics: its load (kernarg -> st -> tag_b, a dependent scalar chain) must
  // not sit in front of the event loads
  u64 lo = MM_INIT_MIN, hi = MM_INIT_MAX;
  u32 used = 0;
  const u64 stride = (u64)nblk * BLOCK;
  if constexpr (AOS) {
    for (u64 i = (u64)blk * BLOCK + threadIdx.x; i < n; i += stride) {
      uint4 r = aos[i];
      bool ok = !HAS_P || (short)(r.y & 0xffff) == 1;
      if (ok) {
        u64 e = TimeCodec<long long>::enc((long long)(((u64)r.w << 32) | r.z));
        lo = e < lo ? e : lo;
        hi = e > hi ? e : hi;
        ++used;
      }
    }
  } else if constexpr (VEC == 2) {
    const u64 n2 = n >> 1;
    const longlong2* t2 = reinterpret_cast<const longlong2*>(t);
    const u32* p2 = reinterpret_cast<const u32*>(p);
    // K0_UN independent 16-byte loads per thread per sweep: latency-bound otherwise (8 MB must be in flight at once)
    for (u64 i0 = (u64)blk * BLOCK + threadIdx.x; i0 < n2; i0 += K0_UN * stride) {
      longlong2 v[K0_UN];
      u32 pp[K0_UN];
      bool in[K0_UN];
#pragma unroll
      for (int j = 0; j < K0_UN; ++j) {
        const u64 i = i0 + (u64)j * stride;
        in[j] = i < n2;
        if (in[j]) {
          v[j] = t2[i];
          if constexpr (HAS_P) pp[j] = p2[i];
        }
      }
#pragma unroll
      for (int j = 0; j < K0_UN; ++j) {
        if (!in[j]) continue;
        bool ok0 = true, ok1 = true;
        if constexpr (HAS_P) {
          ok0 = (short)(pp[j] & 0xffff) == 1;
          ok1 = (short)(pp[j] >> 16) == 1;
        }
        if (ok0) {
          u64 e = TimeCodec<T>::enc((T)v[j].x);
          lo = e < lo ? e : lo;
          hi = e > hi ? e : hi;
          ++used;
        }
        if (ok1) {
          u64 e = TimeCodec<T>::enc((T)v[j].y);
          lo = e < lo ? e : lo;
          hi = e > hi ? e : hi;
          ++used;
        }
      }
    }
    if ((n & 1) && blk == 0 && threadIdx.x == 0) {
      u64 i = n - 1;
      if (!HAS_P || p[i] == 1) {
        u64 e = TimeCodec<T>::enc(t[i]);
        lo = e < lo ? e : lo;
        hi = e > hi ? e : hi;
        ++used;
      }
    }
  } else {
    for (u64 i = (u64)blk * BLOCK + threadIdx.x; i < n; i += stride) {
      if (!HAS_P || p[i] == 1) {
        u64 e = TimeCodec<T>::enc(t[i]);
        lo = e < lo ? e : lo;
        hi = e > hi ? e : hi;
        ++used;
      }
    }
  }

  const u32 tag = tag_override ? tag_override : st->tag_b + 1;
  const u32 parity = tag & 1;
  if (blk == 0 && threadIdx.x == 0) st->tag_a = tag;
  // wave -> block -> one pair of fire-and-forget atomics per block, spread over MM_SLOTS addresses.  The three wave
  // reductions advance together: 6 dependent cross-lane steps instead of 18.
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const u64 lo2 = __shfl_xor(lo, o, 64), hi2 = __shfl_xor(hi, o, 64);
    const u32 u2 = __shfl_xor(used, o, 64);
    lo = lo2 < lo ? lo2 : lo;
    hi = hi2 > hi ? hi2 : hi;
    used += u2;
  }
  __shared__ u64 s_lo[BLOCK / 64], s_hi[BLOCK / 64];
  __shared__ u32 s_used[BLOCK / 64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    s_lo[wave] = lo;
    s_hi[wave] = hi;
    s_used[wave] = used;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 u = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) {
      lo = s_lo[w] < lo ? s_lo[w] : lo;
      hi = s_hi[w] > hi ? s_hi[w] : hi;
      u += s_used[w];
    }
    if (u) {
      const int slot = blk % MM_SLOTS;
      __hip_atomic_fetch_min(&st->mm[parity][slot][0], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_max(&st->mm[parity][slot][1], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&st->cnt[parity][blk % CNT_SLOTS][CNT_USED], u, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <typename T, bool AOS, bool HAS_P, int VEC>
__global__ __launch_bounds__(BLOCK) void k_minmax(const T* __restrict__ t, const int16_t* __restrict__ p,
                                                  const uint4* __restrict__ aos, u64 n, SlotState* st,
                                                  u32 tag_override) {
  // every kernel argument in one scalar round trip (see k_scatter_tiled); never true
  if ((long long)((u64)t | (u64)p | (u64)aos | (u64)st | n | (u64)tag_override) < 0) return;
  minmax_body<T, AOS, HAS_P, VEC>(t, p, aos, n, st, tag_override, blockIdx.x, gridDim.x);
}

// multi-frame launch: grid = (blocks per frame, frames); frame f = descs[f]
template <typename T, bool AOS, bool HAS_P, int VEC, int COND = 0>
__global__ __launch_bounds__(BLOCK) void k_minmax_batch(const FrameDesc* __restrict__ descs) {
  const FrameDesc d = descs[blockIdx.y];
  if (!d.valid || frame_skipped<COND>(d.st)) return;
  minmax_body<T, AOS, HAS_P, VEC>((const T*)d.t, d.p, d.aos, d.n, d.st, 0u, blockIdx.x, gridDim.x);
}

// Sharded frames: a shard's extrema as K0 left them (parity of `tag`) -> {tmin, -tmax} in a 16-byte device buffer of a
// reduction-friendly type (int64 for int64 t, f64 for float t: both exact), so that ONE MIN all-reduce of that buffer over
// the ranks yields the frame's extrema without any host round trip.  Empty shard -> {+max, +max} (neutral for MIN).
template <typename T>
__global__ __launch_bounds__(64) void k_minmax_export(const SlotState* __restrict__ st, u32 tag, void* __restrict__ out) {
  u64 lo, hi;
  load_frame_minmax(st, tag & 1, lo, hi);
  if (threadIdx.x != 0) return;
  const bool empty = lo == MM_INIT_MIN && hi == MM_INIT_MAX;
  if constexpr (std::is_same<T, long long>::value) {
    long long* o = static_cast<long long*>(out);
    const long long big = 0x7fffffffffffffffll;
    const long long tmax = TimeCodec<T>::dec(hi);
    o[0] = empty ? big : TimeCodec<T>::dec(lo);
    o[1] = empty || tmax == (-big - 1) ? big : -tmax;
  } else {
    double* o = static_cast<double*>(out);
    o[0] = empty ? __builtin_inf() : (double)TimeCodec<T>::dec(lo);
    o[1] = empty ? __builtin_inf() : -(double)TimeCodec<T>::dec(hi);
  }
}

// =====================================================================================================
// K1: the fused per-event kernel.
// =====================================================================================================
struct EventResult {
  int xr, yr, ts, disp;
  bool inlier;
};

// A1 + A2 for one event whose time column is already known.  Sets oob when NumPy would raise IndexError.
__device__ inline EventResult event_disparity_col(const DevTables& tb, int column, u32 x, u32 y, bool& oob) {
  EventResult r{0, 0, column, 0, false};
  oob = false;
  if (x >= (u32)tb.cam_w || y >= (u32)tb.cam_h) {  // map[y, x] IndexError (calib:279-280)
    oob = true;
    return r;
  }
  const u32 l = tb.lut[x * (u32)tb.cam_h + y];
  r.xr = (int)(short)(l & 0xffff);
  r.yr = (int)(short)(l >> 16);
  const bool y_ok = r.yr >= 0 && r.yr < tb.xmap_h - 1;  // xmd:23 (last X-map row excluded)
  if (!y_ok) return r;
  if ((u32)r.ts >= (u32)tb.xmap_w) {  // only reachable when a caller hands in extrema that do not bound t
    oob = true;
    return r;
  }
  const int xp = (int)tb.xmap[r.ts * tb.xmap_h + r.yr];                // xmd:25
  r.disp = (int)(short)(xp - r.xr - tb.x_offset);                      // int16 wrap-around (xmd:27)
  r.inlier = r.disp >= 0;                                              // xmd:29
  return r;
}

// A1 + A2 for one event.  `used` = belongs to the frame (polarity).
template <typename T>
__device__ inline EventResult event_disparity(const DevTables& tb, const TimeNorm<T>& tn, u32 x, u32 y, T t,
                                              bool used, bool& oob) {
  oob = false;
  if (!used) return EventResult{0, 0, 0, 0, false};
  return event_disparity_col(tb, tn.column(t), x, y, oob);
}

// cell of the disparity frame an inlier event writes; false = NumPy IndexError
template <int VIEW>
__device__ inline bool event_cell(const DevTables& tb, const EventResult& r, u32 x, u32 y, u32& cell) {
  if constexpr (VIEW == 0) {
    int col = (int)(short)(r.xr + r.disp);  // calib:300: int16 add (= xp - x_offset), rint is a no-op
    if (col < 0) col += tb.rect_w;          // NumPy negative index wraps once
    if (col < 0 || col >= tb.rect_w || r.yr >= tb.rect_h) return false;
    cell = (u32)col * (u32)tb.rect_h + (u32)r.yr;  // projector-view key frame is column-major [col][row]
  } else {
    cell = y * (u32)tb.cam_w + x;  // camera-view key frame is row-major; bounds checked by the LUT gather
  }
  return true;
}

// EPT = events per thread: 4 (vector loads: 8 B of x, 8 B of y, 2 x 16 B of t, 8 B of p per thread;
// needs 8/8/16/8-byte aligned columns) or 1 (any alignment).  AOS: one 16-B record per thread.
template <typename T, bool AOS, bool HAS_P, int EPT, int VIEW>
__device__ __forceinline__ void scatter_direct_body(const uint16_t* __restrict__ xs, const uint16_t* __restrict__ ys,
                                                    const T* __restrict__ ts, const int16_t* __restrict__ ps,
                                                    const uint4* __restrict__ aos, u64 n, u64 idx_offset,
                                                    const DevTables& tb, SlotState* st, u32 tag_override, u64 mm_lo,
                                                    u64 mm_hi, const void* __restrict__ mm_ext, u64* __restrict__ frame,
                                                    unsigned char* __restrict__ dirty, const u32 blk, const int sorted_mode = 0) {
  // sorted_mode (never with a polarity column): the caller expects the frame sorted by t -- extrema = t[0], t[n-1], K0 is not
  // launched, and every event is verified against them below exactly as k_scatter_tiled does (a failure marks the frame: it is
  // redone with K0).  The reference's own recordings are frames of this kind: ~150 k sorted events, too sparse for the tiles.
  const bool srt = sorted_mode != 0 && !tag_override && !HAS_P && n > 0;
  const u32 tag = tag_override ? tag_override : (srt ? st->tag_b + 1 : st->tag_a);
  const u32 parity = tag & 1;
  u64 lo, hi;
  if (tag_override) {  // sharded mode: the FRAME's extrema come from the all-reduce of the shards' extrema
    lo = mm_lo;
    hi = mm_hi;
    if (mm_ext) {  // {tmin, -tmax} in device memory (see scatter_tiled_body)
      if constexpr (std::is_same<T, long long>::value) {
        const long long* m = static_cast<const long long*>(mm_ext);
        lo = TimeCodec<T>::enc(m[0]);
        hi = TimeCodec<T>::enc(-m[1]);
      } else {
        const double* m = static_cast<const double*>(mm_ext);
        lo = TimeCodec<T>::enc((T)m[0]);
        hi = TimeCodec<T>::enc((T)(-m[1]));
      }
    }
  } else if (srt) {
    T t_first, t_last;
    if constexpr (AOS) {
      const uint4 a = aos[0], b = aos[n - 1];
      t_first = (T)(long long)(((u64)a.w << 32) | a.z);
      t_last = (T)(long long)(((u64)b.w << 32) | b.z);
    } else {
      t_first = ts[0];
      t_last = ts[n - 1];
    }
    lo = TimeCodec<T>::enc(t_first);
    hi = TimeCodec<T>::enc(t_last);
    if (hi < lo) hi = lo;  // not sorted at all: keep the arithmetic defined; the verification flags the frame
    if (blk == 0) {
      if (threadIdx.x == 0) {
        st->tag_a = tag;            // K2 reads tag_a and copies it to tag_b
        st->mm[parity][0][0] = lo;  // for xm_frame_stats.t_min / t_max
        st->mm[parity][0][1] = hi;
      }
      if (threadIdx.x < MM_SLOTS) {
        st->mm[parity ^ 1][threadIdx.x][0] = MM_INIT_MIN;
        st->mm[parity ^ 1][threadIdx.x][1] = MM_INIT_MAX;
      }
    }
  } else {
    load_frame_minmax(st, parity, lo, hi);
    if (blk == 0) {
      if (threadIdx.x == 0) st->tag_b = tag;
      // re-arm the other parity's min/max slots for the next frame on this slot
      if (threadIdx.x < MM_SLOTS) {
        st->mm[parity ^ 1][threadIdx.x][0] = MM_INIT_MIN;
        st->mm[parity ^ 1][threadIdx.x][1] = MM_INIT_MAX;
      }
    }
  }
  const TimeNorm<T> tn(TimeCodec<T>::dec(lo), TimeCodec<T>::dec(hi), tb.t_px_scale);
  const u64 key_hi = (u64)tag << KEY_TAG_SHIFT;

  u32 x[EPT], y[EPT];
  T t[EPT];
  bool used[EPT];
  const u64 base = ((u64)blk * BLOCK + threadIdx.x) * EPT;
  if constexpr (AOS) {
    static_assert(EPT == 1, "AoS: one record per thread");
    used[0] = base < n;
    if (used[0]) {
      uint4 r = aos[base];
      x[0] = r.x & 0xffff;
      y[0] = r.x >> 16;
      t[0] = (T)(long long)(((u64)r.w << 32) | r.z);
      if (HAS_P) used[0] = (short)(r.y & 0xffff) == 1;
    }
  } else if constexpr (EPT == 4) {
    if (base + 4 <= n) {
      const uint2 xv = *reinterpret_cast<const uint2*>(xs + base);
      const uint2 yv = *reinterpret_cast<const uint2*>(ys + base);
      x[0] = xv.x & 0xffff; x[1] = xv.x >> 16; x[2] = xv.y & 0xffff; x[3] = xv.y >> 16;
      y[0] = yv.x & 0xffff; y[1] = yv.x >> 16; y[2] = yv.y & 0xffff; y[3] = yv.y >> 16;
      if constexpr (sizeof(T) == 8) {
        const longlong2 a = *reinterpret_cast<const longlong2*>(ts + base);
        const longlong2 b = *reinterpret_cast<const longlong2*>(ts + base + 2);
        __builtin_memcpy(&t[0], &a.x, 8); __builtin_memcpy(&t[1], &a.y, 8);
        __builtin_memcpy(&t[2], &b.x, 8); __builtin_memcpy(&t[3], &b.y, 8);
      } else {
        const float4 a = *reinterpret_cast<const float4*>(ts + base);
        __builtin_memcpy(&t[0], &a.x, 4); __builtin_memcpy(&t[1], &a.y, 4);
        __builtin_memcpy(&t[2], &a.z, 4); __builtin_memcpy(&t[3], &a.w, 4);
      }
      used[0] = used[1] = used[2] = used[3] = true;
      if constexpr (HAS_P) {
        const uint2 pv = *reinterpret_cast<const uint2*>(ps + base);
        used[0] = (short)(pv.x & 0xffff) == 1; used[1] = (short)(pv.x >> 16) == 1;
        used[2] = (short)(pv.y & 0xffff) == 1; used[3] = (short)(pv.y >> 16) == 1;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        used[k] = base + k < n;
        if (used[k]) {
          x[k] = xs[base + k];
          y[k] = ys[base + k];
          t[k] = ts[base + k];
          if (HAS_P) used[k] = ps[base + k] == 1;
        }
      }
    }
  } else {
    used[0] = base < n;
    if (used[0]) {
      x[0] = xs[base];
      y[0] = ys[base];
      t[0] = ts[base];
      if (HAS_P) used[0] = ps[base] == 1;
    }
  }

  if (srt) {  // verify the expectation: 2 compares per event
    bool bad = false;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const u64 e = TimeCodec<T>::enc(used[k] ? t[k] : TimeCodec<T>::dec(lo));
      bad = bad || e < lo || e > hi;
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) {
      __hip_atomic_fetch_add(&st->cnt[parity][blk % CNT_SLOTS][CNT_UNSORTED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&st->unsorted_sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (u32* hf = st->host_flags) host_flag_store(hf, tag);
    }
  }
  u32 n_in = 0, n_oob = 0;
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    bool oob;
    const EventResult r = event_disparity<T>(tb, tn, x[k], y[k], t[k], used[k], oob);
    bool write = r.inlier;
    u32 cell = 0;
    if (write && !event_cell<VIEW>(tb, r, x[k], y[k], cell)) {
      write = false;
      oob = true;
    }
    if (write) {
      const u64 key = key_hi | ((idx_offset + base + k) << KEY_IDX_SHIFT) | (u64)(u32)r.disp;
      __hip_atomic_fetch_max(&frame[cell], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (VIEW == 0 && dirty) dirty[cell >> 4] = dirty_byte(tag);
    }
    // wavefront ballots: one popcount per wave instead of per-lane counters
    n_in += __popcll(__ballot(write));
    n_oob += __popcll(__ballot(oob));
  }
  __shared__ u32 s_in, s_oob;
  if (threadIdx.x == 0) {
    s_in = 0;
    s_oob = 0;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    if (n_in) atomicAdd(&s_in, n_in);
    if (n_oob) atomicAdd(&s_oob, n_oob);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    u32* c = st->cnt[parity][blk % CNT_SLOTS];
    if (s_in) __hip_atomic_fetch_add(&c[CNT_INLIER], s_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (s_oob) __hip_atomic_fetch_add(&c[CNT_OOB], s_oob, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <typename T, bool AOS, bool HAS_P, int EPT, int VIEW>
__global__ __launch_bounds__(BLOCK) void k_scatter(const uint16_t* __restrict__ xs, const uint16_t* __restrict__ ys,
                                                   const T* __restrict__ ts, const int16_t* __restrict__ ps,
                                                   const uint4* __restrict__ aos, u64 n, u64 idx_offset,
                                                   DevTables tb, SlotState* st, u32 tag_override, u64 mm_lo,
                                                   u64 mm_hi, const void* __restrict__ mm_ext, u64* __restrict__ frame,
                                                   unsigned char* __restrict__ dirty, int sorted_mode) {
  scatter_direct_body<T, AOS, HAS_P, EPT, VIEW>(xs, ys, ts, ps, aos, n, idx_offset, tb, st, tag_override, mm_lo, mm_hi, mm_ext,
                                                frame, dirty, blockIdx.x, sorted_mode);
}

__device__ inline void scatter_empty_frame(SlotState* st, int sorted_mode);

// one thread per event, frame from a descriptor in device memory (sparse frames of a device-resident stream: ingest);
// grid = (blocks for the largest frame the host allows for, frames); a frame without events still does block 0's bookkeeping
template <typename T, bool AOS, bool HAS_P, int VIEW>
__global__ __launch_bounds__(BLOCK) void k_scatter_direct_batch(const FrameDesc* __restrict__ descs, DevTables tb, int sorted_mode) {
  const FrameDesc d = descs[blockIdx.y];
  if (!d.valid) return;
  if (blockIdx.x != 0 && (u64)blockIdx.x * BLOCK >= d.n) return;
  if (d.n == 0 && sorted_mode) {  // (only block 0 gets here) nothing to take the extrema from: what the tiled kernel does
    scatter_empty_frame(d.st, sorted_mode);
    return;
  }
  scatter_direct_body<T, AOS, HAS_P, 1, VIEW>(d.x, d.y, (const T*)d.t, d.p, d.aos, d.n, 0ull, tb, d.st, 0u, 0ull, 0ull, nullptr,
                                              d.key_frame, nullptr, blockIdx.x, sorted_mode);
}


// =====================================================================================================
// K1 (tiled): the same per-event work, restructured around what the chip charges for.
//
// Measured on MI355X (profiles/r01_ubench_atomics.md): a lane-divergent atomic costs ~39 ps of chip time per
// lane whatever its width/scope, a divergent load ~13 ps, but the same operations coalesced cost 6-10x less:
// the price is per (lane -> distinct cache line) request.  The direct kernel issues 3 such requests per event
// (LUT gather, X-map gather, atomic).  Here one block owns TILE_EVENTS consecutive events = one thin time slice:
//   * its LUT band  (w_x camera columns around the slice's mean x)   -> LDS, coalesced (column-major table)
//   * its X-map band (w_ts time columns around the slice's mean column) -> LDS, coalesced
//   * last-writer-wins is resolved in LDS first: one u32 slot per (time column, rectified row) -- events of
//     the same slot hit the same frame cell because cell = (yr, X[yr, ts]) -- holding max((local idx+1)<<16 | disp)
//   * winners are flushed with lanes walking consecutive rows of one time column; the key frame is
//     column-major, so a wave's atomics fall into a few cache lines instead of 64.
// Events outside the windows (unsorted / raster-ordered input, noise) take the direct global path inside
// the same kernel: always correct, only slower.  Camera view: slot = (row, x - x_lo), frame row-major.
// =====================================================================================================
#ifdef XM_ABLATE
__device__ int g_ablate = 0;  // bit0: no flush atomics, bit1: no LDS slot atomics, bit2: no band loads, bit3: no time divide
#define XM_ABL(bit) (g_ablate & (1 << (bit)))  // bit 2 (band loads) no longer wired
__device__ unsigned long long g_timeline[64][16];  // [block][phase] s_memtime stamps of thread 0 (experiments only)
#define XM_STAMP(ph) do { if ((threadIdx.x == 0) && blockIdx.x < 64) g_timeline[blockIdx.x][ph] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define XM_ABL(bit) 0
#define XM_STAMP(ph) do { } while (0)
#endif
#ifndef XM_TILE_THREADS
#define XM_TILE_THREADS 512
#endif
constexpr int TILE_THREADS = XM_TILE_THREADS;   // 512 x 8 or 1024 x 4 events: same LDS tile, different latency/issue trade
#ifdef XM_TILE_EPT  // experiments: events per thread decoupled from the block size (smaller tiles)
constexpr int TILE_EPT = XM_TILE_EPT;
#else
constexpr int TILE_EPT = 4096 / XM_TILE_THREADS;
#endif
constexpr int TILE_EVENTS = TILE_THREADS * TILE_EPT;  // largest block: 4096 events (the LDS slots hold (local idx + 1) << 16)

// VEC: SoA columns 16-byte aligned -> each thread loads TILE_EPT consecutive events with 8/16-byte loads.  A compile-time
// switch, not a per-block branch: with both load paths in one kernel the compiler's wait-count bookkeeping at the join
// put full vmcnt waits in front of the event loads and of the extrema reduction (seen in the ISA).
#ifdef XM_K1_WAVES_PER_EU  // experiments: cap the VGPRs so that this many waves fit a SIMD (HIP's 2nd launch-bounds argument)
#define XM_K1_BOUNDS __launch_bounds__(TILE_THREADS, XM_K1_WAVES_PER_EU)
#else
#define XM_K1_BOUNDS __launch_bounds__(TILE_THREADS)
#endif
// blk / nblk = this block's index among the frame's blocks / their number (blockIdx.x, gridDim.x of a single-frame launch).
// mm_ext (sharded mode, tag_override != 0): the FRAME's extrema in device memory as {tmin, -tmax} (int64 for int64 t, f64
// for float t) -- the buffer the ranks MIN-all-reduce -- read here so that no host round trip sits between the collective
// and this kernel; NULL: mm_lo / mm_hi carry the encoded extrema.
template <typename T, bool AOS, bool HAS_P, int VIEW, bool VEC, bool KEY32 = false>
__device__ __forceinline__ void scatter_tiled_body(
    const uint16_t* __restrict__ xs, const uint16_t* __restrict__ ys, const T* __restrict__ ts,
    const int16_t* __restrict__ ps, const uint4* __restrict__ aos, u64 n, u64 idx_offset, const DevTables& tb, SlotState* st,
    u32 tag_override, u64 mm_lo, u64 mm_hi, const void* __restrict__ mm_ext, u64* __restrict__ frame,
    unsigned char* __restrict__ dirty, int w_ts, int w_x, int sorted_mode, const u32 blk, const u32 nblk) {
  static_assert(!(AOS && VEC), "AoS records are loaded one per lane");
  constexpr bool PROJ32 = KEY32 && VIEW == 0, CAM32 = KEY32 && VIEW == 1;  // (see KEY32_DISP_BITS)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS carve-up (16-byte aligned pieces; the two bands keep 16 B of slack for their alignment shift).  The LUT band and
  // the winner slots SHARE one region: the band is only read by the first gather of the fast path, the slots only written
  // after it -- two extra barriers buy 26 KB per block, i.e. a third resident block per CU (block residency is what bounds
  // the pipelined frame rate: tools/block_timeline.py).
  const int win_words = VIEW == 0 ? w_ts * tb.xmap_h : w_x * tb.cam_h;
  const int win_q = (win_words + 3) >> 2;  // uint4 count
  // LDS-direct band loads write whole waves (64 x 16 B): each band keeps one wave of slack behind it (k1_lds_bytes())
  const int lut_q = ((w_x * tb.cam_h + 3) >> 2) + 1 + 64;
  u32* win = reinterpret_cast<u32*>(smem);
  u32* lut_base = win;
  int16_t* xm_base = reinterpret_cast<int16_t*>(win + 4 * max(win_q, lut_q));
  __shared__ u32 s_in, s_oob;

  const int tid = threadIdx.x;
  const int nthreads = blockDim.x;               // 64 .. 1024, chosen per frame by the host so that the block's
  const int ev_per_block = nthreads * TILE_EPT;  // time slice fits the LDS window (see launch_scatter)
  XM_STAMP(0);
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Neighbouring tiles share
  // almost all of their LUT band and a column of their X-map band, so XCD k takes the k-th CONTIGUOUS eighth of the
  // frame's tiles: the bands then come out of that XCD's L2 instead of being fetched over the fabric once per block.
  const u32 tile = xcd_contiguous(blk, nblk);
  const u64 block_base = (u64)tile * ev_per_block;  // < n: the host launches ceil(n / ev_per_block) blocks, n > 0

  // ---- 1. Every load that depends on nothing is ISSUED here, small ones first, and nothing is consumed before the
  //         last one is out: vector memory returns in order, so the few bytes that locate the tile (samples, frame
  //         extrema) can be waited for with the 48 KB of events still in flight behind them.
  // 1a. three sampled events (first / middle / last of the block) locate the time slice; t[0] and t[n-1] are the frame
  //     extrema of the time-sorted mode.  Uniform loads.
  int sx[3];
  T st_t[3];
  T t_first, t_last;
  {
    const u64 last = (block_base + ev_per_block <= n ? block_base + ev_per_block : n) - 1;
    const u64 si[3] = {block_base, block_base + ((last - block_base) >> 1), last};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if constexpr (AOS) {
        const uint4 r = aos[si[j]];
        sx[j] = (int)(r.x & 0xffff);
        st_t[j] = (T)(long long)(((u64)r.w << 32) | r.z);
      } else {
        sx[j] = (int)xs[si[j]];
        st_t[j] = ts[si[j]];
      }
    }
    if constexpr (AOS) {
      const uint4 a = aos[0], b = aos[n - 1];
      t_first = (T)(long long)(((u64)a.w << 32) | a.z);
      t_last = (T)(long long)(((u64)b.w << 32) | b.z);
    } else {
      t_first = ts[0];
      t_last = ts[n - 1];
    }
  }
  // 1b. frame extrema as K0 left them: BOTH parities (2 x 16 B in lanes < MM_SLOTS), selected once the tag is known --
  //     loading only the right one would put a scalar load (the tag) in front of this vector load.
  //     Lanes >= MM_SLOTS load a duplicate slot, which a min/max reduction does not notice.
  const ulonglong2 mm_p0 = *reinterpret_cast<const ulonglong2*>(&st->mm[0][tid & (MM_SLOTS - 1)][0]);
  const ulonglong2 mm_p1 = *reinterpret_cast<const ulonglong2*>(&st->mm[1][tid & (MM_SLOTS - 1)][0]);
  // 1c. this thread's events, kept PACKED (two 16-bit coordinates per register) until after the window is known
  u32 xw[TILE_EPT / 2], yw[TILE_EPT / 2], pw[TILE_EPT / 2];
  T tt[TILE_EPT];
  u32 inb = 0;  // bit k: event k of this thread exists (index < n)
#pragma unroll
  for (int q = 0; q < TILE_EPT / 2; ++q) xw[q] = yw[q] = pw[q] = 0;
#pragma unroll
  for (int k = 0; k < TILE_EPT; ++k) tt[k] = (T)0;
  if constexpr (AOS) {  // EventCD records: event k*nthreads + tid, one 16-byte load each, clamped (branch-free)
    {
#pragma unroll
      for (int k = 0; k < TILE_EPT; ++k) {
        const u64 i = block_base + (u32)k * nthreads + tid;
        const bool ok = i < n;
        const uint4 r = aos[ok ? i : block_base];
        inb |= ok ? 1u << k : 0u;
        if (k & 1) {
          xw[k >> 1] |= (r.x & 0xffff) << 16;
          yw[k >> 1] |= r.x & 0xffff0000u;
          pw[k >> 1] |= r.y << 16;
        } else {
          xw[k >> 1] = r.x & 0xffff;
          yw[k >> 1] = r.x >> 16;
          pw[k >> 1] = r.y & 0xffff;
        }
        tt[k] = (T)(long long)(((u64)r.w << 32) | r.z);
      }
    }
  } else if constexpr (VEC) {  // TILE_EPT consecutive events per thread: 8/16-byte loads of x / y / p, 16-byte loads of t
    // Ragged end of the frame, branch-free: a thread past the end re-reads the last group (its events are masked out);
    // the thread that straddles the end loads its whole aligned group -- an aligned 8/16-byte word whose first element
    // is valid cannot cross into another page -- and a t pair that starts past the end is redirected to the first pair.
    // 32-bit element indices (n <= 2^28) and byte offsets: the loads take the scalar-base + 32-bit-offset form
    const u32 n32 = (u32)n;
    const u32 base_true = (u32)block_base + (u32)tid * TILE_EPT;
    const u32 last_grp = (n32 - 1u) & ~(u32)(TILE_EPT - 1);
    const u32 base = base_true < last_grp ? base_true : last_grp;
    const char* xs_b = reinterpret_cast<const char*>(xs);
    const char* ys_b = reinterpret_cast<const char*>(ys);
    const char* ps_b = reinterpret_cast<const char*>(ps);
    const char* ts_b = reinterpret_cast<const char*>(ts);
    if constexpr (TILE_EPT == 8) {
      const uint4 xv = *reinterpret_cast<const uint4*>(xs_b + base * 2u);
      const uint4 yv = *reinterpret_cast<const uint4*>(ys_b + base * 2u);
      xw[0] = xv.x; xw[1] = xv.y; xw[2] = xv.z; xw[3] = xv.w;
      yw[0] = yv.x; yw[1] = yv.y; yw[2] = yv.z; yw[3] = yv.w;
      if constexpr (HAS_P) {
        const uint4 pv = *reinterpret_cast<const uint4*>(ps_b + base * 2u);
        pw[0] = pv.x; pw[1] = pv.y; pw[2] = pv.z; pw[3] = pv.w;
      }
    } else {
      const uint2 xv = *reinterpret_cast<const uint2*>(xs_b + base * 2u);
      const uint2 yv = *reinterpret_cast<const uint2*>(ys_b + base * 2u);
      xw[0] = xv.x; xw[1] = xv.y;
      yw[0] = yv.x; yw[1] = yv.y;
      if constexpr (HAS_P) {
        const uint2 pv = *reinterpret_cast<const uint2*>(ps_b + base * 2u);
        pw[0] = pv.x; pw[1] = pv.y;
      }
    }
    if constexpr (sizeof(T) == 8) {
#pragma unroll
      for (int q = 0; q < TILE_EPT / 2; ++q) {
        const longlong2 a = *reinterpret_cast<const longlong2*>(ts_b + (base + 2 * q < n32 ? base + 2 * q : base) * 8u);
        __builtin_memcpy(&tt[2 * q], &a.x, 8);
        __builtin_memcpy(&tt[2 * q + 1], &a.y, 8);
      }
    } else {
#pragma unroll
      for (int q = 0; q < TILE_EPT / 4; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(ts_b + (base + 4 * q < n32 ? base + 4 * q : base) * 4u);
        __builtin_memcpy(&tt[4 * q], &a.x, 4); __builtin_memcpy(&tt[4 * q + 1], &a.y, 4);
        __builtin_memcpy(&tt[4 * q + 2], &a.z, 4); __builtin_memcpy(&tt[4 * q + 3], &a.w, 4);
      }
    }
#pragma unroll
    for (int k = 0; k < TILE_EPT; ++k) inb |= base_true + k < n32 ? 1u << k : 0u;
  } else {  // any alignment / ragged tail: event k*nthreads + tid, still coalesced across lanes, clamped
#pragma unroll
    for (int k = 0; k < TILE_EPT; ++k) {
      const u64 i = block_base + (u32)k * nthreads + tid;
      const bool ok = i < n;
      const u64 ic = ok ? i : block_base;
      const u32 xv = xs[ic], yv = ys[ic];
      tt[k] = ts[ic];
      inb |= ok ? 1u << k : 0u;
      xw[k >> 1] |= xv << ((k & 1) * 16);
      yw[k >> 1] |= yv << ((k & 1) * 16);
      if constexpr (HAS_P) pw[k >> 1] |= (u32)(uint16_t)ps[ic] << ((k & 1) * 16);
    }
  }
  XM_STAMP(1);

  // ---- 2. frame extrema -> time normalisation.  General mode: written by K0.  Time-sorted mode (the caller declared
  //         the frame sorted by t, true for every frame the trigger finder emits): extrema = t[0], t[n-1], K0 is not
  //         launched at all, and the declaration is VERIFIED below (every event must lie inside [t[0], t[n-1]]).
  const u32 tag = tag_override ? tag_override : (sorted_mode ? st->tag_b + 1 : st->tag_a);
  const u32 parity = tag & 1;
  u64 lo, hi;
  if (tag_override) {  // sharded mode: the FRAME's extrema come from the all-reduce of the shards' extrema
    lo = mm_lo;
    hi = mm_hi;
    if (mm_ext) {  // {tmin, -tmax} left in device memory by the collective (uniform loads)
      if constexpr (std::is_same<T, long long>::value) {
        const long long* m = static_cast<const long long*>(mm_ext);
        lo = TimeCodec<T>::enc(m[0]);
        hi = TimeCodec<T>::enc(-m[1]);
      } else {
        const double* m = static_cast<const double*>(mm_ext);
        lo = TimeCodec<T>::enc((T)m[0]);
        hi = TimeCodec<T>::enc((T)(-m[1]));
      }
    }
  } else {
    if (sorted_mode) {
      lo = TimeCodec<T>::enc(t_first);
      hi = TimeCodec<T>::enc(t_last);
      if (hi < lo) hi = lo;  // not sorted at all: keep the arithmetic defined; the verification flags the frame
    } else {
      u64 a = parity ? mm_p1.x : mm_p0.x, b = parity ? mm_p1.y : mm_p0.y;
#pragma unroll
      for (int o = MM_SLOTS / 2; o > 0; o >>= 1) {
        const u64 a2 = __shfl_xor(a, o, 64), b2 = __shfl_xor(b, o, 64);
        a = a2 < a ? a2 : a;
        b = b2 > b ? b2 : b;
      }
      lo = uniform_u64(a);
      hi = uniform_u64(b);
    }
    if (blk == 0) {
      if (tid == 0) {
        if (sorted_mode) {
          st->tag_a = tag;  // K2 reads tag_a and copies it to tag_b
          st->mm[parity][0][0] = lo;  // for xm_frame_stats.t_min / t_max
          st->mm[parity][0][1] = hi;
        } else {
          st->tag_b = tag;
        }
      }
      for (int i = tid; i < MM_SLOTS; i += nthreads) {  // re-arm the other parity's slots for the next frame on this slot
        st->mm[parity ^ 1][i][0] = MM_INIT_MIN;
        st->mm[parity ^ 1][i][1] = MM_INIT_MAX;
      }
    }
  }
  const TimeNorm<T> tn(TimeCodec<T>::dec(lo), TimeCodec<T>::dec(hi), tb.t_px_scale);
  const u64 key_hi = (u64)tag << KEY_TAG_SHIFT;
  if (tid == 0) {
    s_in = 0;
    s_oob = 0;
  }
  XM_STAMP(2);

  // ---- 3. window = median of the samples.  A wrong guess (unsorted input, a noise event) only sends events down
  //         the direct path.
  int x_lo, ts_lo;
  {
    // the column is monotone in t: the median column is the column of the median time (one conversion, not three)
    const u64 e0 = TimeCodec<T>::enc(st_t[0]), e1 = TimeCodec<T>::enc(st_t[1]), e2 = TimeCodec<T>::enc(st_t[2]);
    const u64 lo01 = e0 < e1 ? e0 : e1, hi01 = e0 < e1 ? e1 : e0;
    const u64 m2 = hi01 < e2 ? hi01 : e2;
    const u64 em = lo01 > m2 ? lo01 : m2;
    const int mc = tn.column(TimeCodec<T>::dec(em));
    const int mx = max(min(sx[0], sx[1]), min(max(sx[0], sx[1]), sx[2]));
    x_lo = min(max(mx - w_x / 2, 0), max(tb.cam_w - w_x, 0));
    ts_lo = min(max(mc - w_ts / 2, 0), max(tb.xmap_w - w_ts, 0));
  }
  XM_STAMP(3);
  // The bands are contiguous runs of the column-major tables: [x_lo, x_lo + w_x) x cam_h words and
  // [ts_lo, ts_lo + w_ts) x xmap_h int16.  Aligned 16-byte loads over ONE index space (LUT quads, then X-map quads), so a
  // full-size block issues 3 (1024 threads) or 6 (512) loads per thread, all in flight at once; the LDS copies keep the global misalignment (a
  // few elements of slack in front).  Branch-free on purpose: loads use a clamped index and out-of-range lanes store into
  // a dummy LDS slot -- any predication here turns into one basic block per load with an s_waitcnt vmcnt(0) behind it
  // (seen in the ISA), i.e. serialized L2 round trips.
  const int wx_eff = min(w_x, tb.cam_w), wts_eff = min(w_ts, tb.xmap_w);
  const u32 lut_start = (u32)x_lo * (u32)tb.cam_h, lut_shift = lut_start & 3u;  // in words
  const u32 xm_start = (u32)ts_lo * (u32)tb.xmap_h, xm_shift = xm_start & 7u;   // in int16
  const u32* lut_t = lut_base + lut_shift;
  const int16_t* xm_t = xm_base + xm_shift;
  const uint4* g_lut = reinterpret_cast<const uint4*>(tb.lut + (lut_start - lut_shift));
  const int nq_lut = (int)((lut_shift + (u32)wx_eff * (u32)tb.cam_h + 3u) >> 2);
  const uint4* g_xm = reinterpret_cast<const uint4*>(tb.xmap + (xm_start - xm_shift));
  const int nq_xm = (int)((xm_shift + (u32)wts_eff * (u32)tb.xmap_h + 7u) >> 3);
  const int nq_all = nq_lut + nq_xm;
  uint4* l_lut = reinterpret_cast<uint4*>(lut_base);
  uint4* l_xm = reinterpret_cast<uint4*>(xm_base);
  // LDS-DIRECT loads (global_load_lds_dwordx4, gfx950): the bands go L2 -> LDS without passing through VGPRs -- no 24
  // registers of band data held across the event arithmetic, no ds_write_b128, nothing to wait for until the gathers.
  // Lane l of a wave writes 16 B at M0 + 16 l, so a wave's 64 quads land contiguously: LDS quad index == band quad index,
  // as before.  Waves entirely past the end of a band skip the load (wave-uniform branch); the last, partial wave of a
  // band re-reads the band's last quad for its surplus lanes and writes it into the wave of slack behind the band.
  // A FIXED number of loads per wave is issued here (enough for the C-1M bands), so that the wait for the thread's own
  // events further down can be a counted one (vmcnt(6)) and the bands stay in flight during the time-column arithmetic;
  // taller tables / smaller blocks fetch the rest after that arithmetic (dynamic trip count = full wait, seen in the ISA).
  // The compiler waits with vmcnt(0) before the first use of a register loaded BEFORE an LDS-direct load (seen in the ISA:
  // it does not count past them), i.e. the time-column arithmetic below would wait for the bands too.  Touch the event
  // registers here instead: the wait lands in front of the band loads, where only the events are outstanding (they were
  // issued ~1 us ago and the samples behind them have already arrived), and the bands then fly during the arithmetic.
#pragma unroll
  for (int q = 0; q < TILE_EPT / 2; ++q) asm volatile("" : "+v"(xw[q]), "+v"(yw[q]), "+v"(pw[q]));
#pragma unroll
  for (int k = 0; k < TILE_EPT; ++k) asm volatile("" : "+v"(tt[k]));
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  constexpr int UL_L = TILE_THREADS >= 1024 ? 2 : 4, UL_X = TILE_THREADS >= 1024 ? 1 : 2;
  const int dma_q0 = tid & ~63;  // first band quad of this wave in pass 0
  const int dma_lane = tid & 63;
  uint4* l_dmy = l_xm + (((w_ts * tb.xmap_h + 7) >> 3) + 1 + 64);  // where waves past the end of a band dump their load
  {
#pragma unroll
    for (int k = 0; k < UL_L; ++k) {
      const int q0 = dma_q0 + k * nthreads;
      __builtin_amdgcn_global_load_lds((glb_void*)(g_lut + min(q0 + dma_lane, nq_lut - 1)),
                                       (lds_void*)(q0 < nq_lut ? l_lut + q0 : l_dmy), 16, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < UL_X; ++k) {
      const int q0 = dma_q0 + k * nthreads;
      __builtin_amdgcn_global_load_lds((glb_void*)(g_xm + min(q0 + dma_lane, nq_xm - 1)),
                                       (lds_void*)(q0 < nq_xm ? l_xm + q0 : l_dmy), 16, 0, 0);
    }
    (void)nq_all;
  }
  XM_STAMP(4);

  // ---- 4. with the bands in flight: unpack the events, their time columns (bit-exact with NumPy, see TimeNorm) ---------
  u32 x[TILE_EPT], y[TILE_EPT], lidx[TILE_EPT];
  bool used[TILE_EPT];
  int col[TILE_EPT];
#pragma unroll
  for (int k = 0; k < TILE_EPT; ++k) {
    x[k] = (xw[k >> 1] >> ((k & 1) * 16)) & 0xffff;
    y[k] = (yw[k >> 1] >> ((k & 1) * 16)) & 0xffff;
    used[k] = (inb >> k) & 1;
    if constexpr (HAS_P) used[k] = used[k] && (short)((pw[k >> 1] >> ((k & 1) * 16)) & 0xffff) == 1;
    lidx[k] = VEC ? (u32)tid * TILE_EPT + k : (u32)k * nthreads + tid;
  }
  tn.columns(tt, col);  // every lane; `used` masks the event below
  if (sorted_mode) {  // verify the time-sorted declaration: 2 compares per event
    bool bad = false;
#pragma unroll
    for (int k = 0; k < TILE_EPT; ++k) {
      const u64 e = TimeCodec<T>::enc(tt[k]);
      bad = bad || (used[k] && (e < lo || e > hi));
    }
    if (__ballot(bad) && (tid & 63) == 0) {
      __hip_atomic_fetch_add(&st->cnt[parity][blk % CNT_SLOTS][CNT_UNSORTED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&st->unsorted_sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (u32* hf = st->host_flags) host_flag_store(hf, tag);
    }
  }
  // Events outside the windows (unsorted / raster-ordered input, a noise event) take the global path -- the very functions
  // of the direct kernel -- in a compact loop: one pass handles every lane's next such event, so a wave with a single
  // stray event (the common case, 0.3 % of the events of a sorted frame but half of its waves) runs ~100 instructions, and
  // a wave without any skips the loop.  The block is issue-bound here (4 waves per SIMD), so the two dependent round
  // trips of a stray event are covered by the other waves' arithmetic; the band loads above are in flight meanwhile.
  int xl[TILE_EPT], tl[TILE_EPT];
  bool fast[TILE_EPT];
  u32 smask = 0;
#pragma unroll
  for (int k = 0; k < TILE_EPT; ++k) {
    xl[k] = (int)x[k] - x_lo;
    tl[k] = col[k] - ts_lo;
    fast[k] = used[k] && (u32)xl[k] < (u32)wx_eff && (u32)tl[k] < (u32)wts_eff && y[k] < (u32)tb.cam_h;
    smask |= used[k] && !fast[k] ? 1u << k : 0u;
  }
  u32 n_in = 0, n_oob = 0;
  u32 ovr = 0;  // PROJ32: bit k = event k's LUT entry was fetched from global memory and sits in xl[k]
  if constexpr (PROJ32) {
    // Every event must be resolved in the LDS slots (the key's order field is only the tile).  An event outside the LUT window
    // (x noise) but inside the time window fetches its LUT entry from global memory here and joins the fast path below; an
    // event outside the TIME window cannot use the slots: the frame is marked as failed and redone on the 64-bit path.
    bool bad = false;
    while (__ballot(smask != 0)) {
      const bool act = smask != 0;
      const int ks = act ? __builtin_ctz(smask) : 0;
      smask &= smask - 1;
      u32 ex = x[0], ey = y[0];
      int et = tl[0];
#pragma unroll
      for (int kk = 1; kk < TILE_EPT; ++kk) {
        const bool sel = ks == kk;
        ex = sel ? x[kk] : ex;
        ey = sel ? y[kk] : ey;
        et = sel ? tl[kk] : et;
      }
      bool oob = false;
      if (act) {
        if (ex >= (u32)tb.cam_w || ey >= (u32)tb.cam_h) {
          oob = true;  // map[y, x] IndexError (calib:279-280): dropped and counted, as on the other paths
        } else if ((u32)et >= (u32)wts_eff) {
          bad = true;
        } else {
          const u32 l = tb.lut[ex * (u32)tb.cam_h + ey];
#pragma unroll
          for (int kk = 0; kk < TILE_EPT; ++kk) xl[kk] = ks == kk ? (int)l : xl[kk];
          ovr |= 1u << ks;
        }
      }
      n_oob += __popcll(__ballot(oob));
    }
    if (__ballot(bad) && (tid & 63) == 0) {
      __hip_atomic_fetch_add(&st->cnt[parity][blk % CNT_SLOTS][CNT_UNSORTED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (u32* hf = st->host_flags) host_flag_store(hf, tag);
    }
  } else
  while (__ballot(smask != 0)) {
    const bool act = smask != 0;
    const int ks = act ? __builtin_ctz(smask) : 0;
    smask &= smask - 1;
    u32 ex = x[0], ey = y[0], el = lidx[0];
    int ec = col[0];
#pragma unroll
    for (int kk = 1; kk < TILE_EPT; ++kk) {
      const bool sel = ks == kk;
      ex = sel ? x[kk] : ex;
      ey = sel ? y[kk] : ey;
      el = sel ? lidx[kk] : el;
      ec = sel ? col[kk] : ec;
    }
    bool oob = false, write = false;
    if (act) {
      const EventResult r = event_disparity_col(tb, ec, ex, ey, oob);
      u32 cell = 0;
      write = r.inlier;
      if (write && !event_cell<VIEW>(tb, r, ex, ey, cell)) {
        write = false;
        oob = true;
      }
      if (write) {
        if constexpr (CAM32) {
          cell = ex * (u32)tb.cam_h + ey;  // (the compact camera frame is column-major; event_cell has checked the pixel)
          __hip_atomic_fetch_max(reinterpret_cast<u32*>(frame) + cell,
                                 ((u32)(idx_offset + block_base + el + 1) << KEY32_DISP_BITS) | ((u32)r.disp & 0xfffu),
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          const u64 key = key_hi | ((idx_offset + block_base + el) << KEY_IDX_SHIFT) | (u64)(u32)r.disp;
          __hip_atomic_fetch_max(&frame[cell], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (VIEW == 0 && dirty) dirty[cell >> 4] = dirty_byte(tag);
        }
      }
    }
    n_in += __popcll(__ballot(write));
    n_oob += __popcll(__ballot(oob));
  }
  XM_STAMP(5);
  for (int q0 = dma_q0 + UL_L * nthreads; q0 < nq_lut; q0 += nthreads)  // taller tables / smaller blocks: the rest
    __builtin_amdgcn_global_load_lds((glb_void*)(g_lut + min(q0 + dma_lane, nq_lut - 1)), (lds_void*)(l_lut + q0), 16, 0, 0);
  for (int q0 = dma_q0 + UL_X * nthreads; q0 < nq_xm; q0 += nthreads)
    __builtin_amdgcn_global_load_lds((glb_void*)(g_xm + min(q0 + dma_lane, nq_xm - 1)), (lds_void*)(l_xm + q0), 16, 0, 0);
  // the LDS-direct loads are tracked by vmcnt like any vector load: all of them landed before the barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  XM_STAMP(11);
  XM_STAMP(12);
  __syncthreads();  // bands visible
  XM_STAMP(6);

  // ---- 5. fast path, BRANCH-FREE so that the events' LDS round trips overlap: A1 + A2 out of the LDS bands with clamped
  //         addresses; then the LUT band's region becomes the slot array and collisions are resolved with ds_max_u32.
  bool wr[TILE_EPT];
  int slot[TILE_EPT];
  u32 val[TILE_EPT];
  {
    u32 l[TILE_EPT];
#pragma unroll
    for (int k = 0; k < TILE_EPT; ++k) {
      const bool o_k = PROJ32 && ((ovr >> k) & 1u);
      l[k] = lut_t[fast[k] ? xl[k] * tb.cam_h + (int)y[k] : 0];
      if constexpr (PROJ32) {
        l[k] = o_k ? (u32)xl[k] : l[k];
        fast[k] = fast[k] || o_k;
      }
    }
    int xr[TILE_EPT], yr[TILE_EPT], xp[TILE_EPT];
    bool yok[TILE_EPT];
#pragma unroll
    for (int k = 0; k < TILE_EPT; ++k) {
      xr[k] = (int)(short)(l[k] & 0xffff);
      yr[k] = (int)(short)(l[k] >> 16);
      yok[k] = fast[k] && yr[k] >= 0 && yr[k] < tb.xmap_h - 1;  // xmd:23
      xp[k] = (int)xm_t[yok[k] ? tl[k] * tb.xmap_h + yr[k] : 0];
    }
#pragma unroll
    for (int k = 0; k < TILE_EPT; ++k) {
      const int disp = (int)(short)(xp[k] - xr[k] - tb.x_offset);  // int16 wrap (xmd:27)
      bool write = yok[k] && disp >= 0;                               // xmd:29
      if constexpr (VIEW == 0) {
        int fc = (int)(short)(xr[k] + disp);  // = xp - x_offset (calib:300)
        if (fc < 0) fc += tb.rect_w;
        const bool in_frame = fc >= 0 && fc < tb.rect_w && yr[k] < tb.rect_h;
        n_oob += __popcll(__ballot(write && !in_frame));  // NumPy IndexError
        write = write && in_frame;
        slot[k] = tl[k] * tb.xmap_h + yr[k];
      } else if constexpr (CAM32) {
        slot[k] = xl[k] * tb.cam_h + (int)y[k];  // [window column][row], as the LUT band: the compact camera frame is column-major
      } else {
        slot[k] = (int)y[k] * w_x + xl[k];
      }
      wr[k] = write;
      val[k] = ((lidx[k] + 1) << 16) | (u32)disp;
      n_in += __popcll(__ballot(write));  // wavefront ballots instead of per-lane counters
    }
  }
  XM_STAMP(13);
  __syncthreads();  // every LUT gather has landed: the region can be reused
  {
    uint4* l_win = reinterpret_cast<uint4*>(win);
    for (int i = tid; i < win_q; i += nthreads) l_win[i] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();  // cleared slots visible
  XM_STAMP(14);
#pragma unroll
  for (int k = 0; k < TILE_EPT; ++k)
    if (wr[k] && !XM_ABL(1)) atomicMax(&win[slot[k]], val[k]);
  if ((tid & 63) == 0) {
    if (n_in) atomicAdd(&s_in, n_in);
    if (n_oob) atomicAdd(&s_oob, n_oob);
  }
  XM_STAMP(7);
  __syncthreads();
  XM_STAMP(8);

  // ---- 6. flush the winners: consecutive lanes -> consecutive slots = consecutive rows of one frame column (VIEW 0) /
  //         consecutive x of one row (VIEW 1).  One pass over the whole window (empty slots cost an LDS read, nothing
  //         else); all LDS reads of a thread are issued before its first atomic.
  {
    constexpr int FL = 4;
    static_assert(KEY_IDX_SHIFT == 16, "slot value ((local idx + 1) << 16 | disp) is added to the key as is");
    // key = tag | (global idx << 16) | disp, and the slot holds ((local idx + 1) << 16) | disp: one 64-bit add
    const u64 key_base = key_hi + ((idx_offset + block_base - 1) << KEY_IDX_SHIFT);
    const int per = VIEW == 0 ? tb.xmap_h : CAM32 ? tb.cam_h : w_x;  // slots per window column (VIEW 0, compact camera frame) / per camera row (VIEW 1)
    // (q, r) = divmod(slot, per), advanced incrementally: slot -> slot + nthreads is (q + dq, r + dr) with one carry
    const int dq = nthreads / per, dr = nthreads - dq * per;
    int q_i, r_i;
    {
      q_i = (int)((float)tid * (1.0f / (float)per));
      r_i = tid - q_i * per;
      if (r_i < 0) { q_i -= 1; r_i += per; }
      if (r_i >= per) { q_i += 1; r_i -= per; }
    }
    for (int i0 = tid; i0 < win_words; i0 += FL * nthreads) {
      u32 v[FL];
      int xv[FL], qs[FL], rs[FL];
#pragma unroll
      for (int j = 0; j < FL; ++j) {
        const int i = min(i0 + j * nthreads, win_words - 1);
        v[j] = win[i];
        xv[j] = VIEW == 0 ? (int)xm_t[i] : 0;
        qs[j] = q_i;
        rs[j] = r_i;
        q_i += dq;
        r_i += dr;
        if (r_i >= per) { r_i -= per; q_i += 1; }
      }
#pragma unroll
      for (int j = 0; j < FL; ++j) {
        const int i = i0 + j * nthreads;
        if (i < win_words && v[j]) {
          const int q = qs[j], r = rs[j];
          const u64 key = key_base + v[j];
          u32 cell;
          if constexpr (VIEW == 0) {  // q = window column, r = rectified row; the frame column comes from the X-map band
            int fc = (int)(short)(xv[j] - tb.x_offset);
            if (fc < 0) fc += tb.rect_w;
            cell = (u32)fc * (u32)tb.rect_h + (u32)r;
          } else if constexpr (CAM32) {  // q = window column, r = camera row: lanes walk consecutive rows of one column of the
            cell = (u32)(x_lo + q) * (u32)tb.cam_h + (u32)r;  // column-major frame (64 lanes = 256 contiguous bytes)
          } else {  // q = camera row, r = x - x_lo
            cell = (u32)q * (u32)tb.cam_w + (u32)(x_lo + r);
          }
          if constexpr (PROJ32) {  // tag4 | tile | disparity into the compact frame (see key32_tag)
            __hip_atomic_fetch_max(reinterpret_cast<u32*>(frame) + cell, key32_tag(tag) | (tile << KEY32_DISP_BITS) | (v[j] & 0xfffu),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else if constexpr (CAM32) {  // (event index + 1) | disparity; the slot holds ((local index + 1) << 16) | disparity
            __hip_atomic_fetch_max(reinterpret_cast<u32*>(frame) + cell,
                                   (((u32)(idx_offset + block_base) + (v[j] >> 16)) << KEY32_DISP_BITS) | (v[j] & 0xfffu),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            if (!XM_ABL(0)) __hip_atomic_fetch_max(&frame[cell], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (VIEW == 0 && dirty) dirty[cell >> 4] = dirty_byte(tag);  // consecutive lanes = consecutive rows
          }
        }
      }
    }
  }
  XM_STAMP(9);
  if (tid == 0) {
    u32* c = st->cnt[parity][blk % CNT_SLOTS];
    if (s_in) __hip_atomic_fetch_add(&c[CNT_INLIER], s_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (s_oob) __hip_atomic_fetch_add(&c[CNT_OOB], s_oob, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  XM_STAMP(10);
}

template <typename T, bool AOS, bool HAS_P, int VIEW, bool VEC, bool KEY32 = false>
__global__ XM_K1_BOUNDS void k_scatter_tiled(
    const uint16_t* __restrict__ xs, const uint16_t* __restrict__ ys, const T* __restrict__ ts,
    const int16_t* __restrict__ ps, const uint4* __restrict__ aos, u64 n, u64 idx_offset, DevTables tb, SlotState* st,
    u32 tag_override, u64 mm_lo, u64 mm_hi, const void* __restrict__ mm_ext, u64* __restrict__ frame,
    unsigned char* __restrict__ dirty, int w_ts, int w_x, int sorted_mode) {
  // All kernel arguments into SGPRs in ONE scalar-load round trip: a test that needs every one of them, placed first.
  // Left alone the compiler fetches them lazily, block by block -- eight dependent s_load -> s_waitcnt pairs along the
  // critical chain of every block (seen in the ISA).  (Inline asm would do it too, but makes every later uniform load a
  // vector load.)  Never true: sizes are non-negative and device addresses have bit 63 clear.
  {
    const u64 pp = (u64)xs | (u64)ys | (u64)ts | (u64)ps | (u64)aos | (u64)tb.lut | (u64)tb.xmap | (u64)st | (u64)frame |
                   (u64)dirty | (u64)mm_ext | n | idx_offset;
    const int pi = tb.cam_w | tb.cam_h | tb.xmap_w | tb.xmap_h | tb.t_px_scale | tb.x_offset | tb.rect_w | tb.rect_h |
                   (int)tag_override | w_ts | w_x | sorted_mode;
    if ((long long)(pp | (u64)(long long)pi) < 0) return;
  }
  scatter_tiled_body<T, AOS, HAS_P, VIEW, VEC, KEY32>(xs, ys, ts, ps, aos, n, idx_offset, tb, st, tag_override, mm_lo, mm_hi, mm_ext,
                                                      frame, dirty, w_ts, w_x, sorted_mode, blockIdx.x, gridDim.x);
}

// A frame without events inside a multi-frame launch: only the slot bookkeeping K1's block 0 does.
__device__ inline void scatter_empty_frame(SlotState* st, int sorted_mode) {
  const u32 tag = sorted_mode ? st->tag_b + 1 : st->tag_a;
  const u32 parity = tag & 1;
  if (threadIdx.x == 0) {
    if (sorted_mode) {
      st->tag_a = tag;
      st->mm[parity][0][0] = MM_INIT_MIN;
      st->mm[parity][0][1] = MM_INIT_MAX;
    } else {
      st->tag_b = tag;
    }
  }
  for (int i = threadIdx.x; i < MM_SLOTS; i += blockDim.x) {
    st->mm[parity ^ 1][i][0] = MM_INIT_MIN;
    st->mm[parity ^ 1][i][1] = MM_INIT_MAX;
  }
}

// Multi-frame launch: grid = (tiles of the largest frame, frames).  One launch exposes frames x tiles blocks to the chip
// (60 frames: 14 700 blocks instead of 245): no per-frame launch ramp, the CUs always have a next block to pick up.  Every
// frame owns a key frame + state (FrameDesc), so blocks of different frames never meet.  The frame's size comes from
// device memory: the same launch serves frames cut out of a device-resident stream (ingest) whose length the host never saw.
template <typename T, bool AOS, bool HAS_P, int VIEW, bool VEC, bool KEY32 = false, int COND = 0>
__global__ XM_K1_BOUNDS void k_scatter_tiled_batch(const FrameDesc* __restrict__ descs, DevTables tb, int w_ts, int w_x,
                                                   int sorted_mode) {
  const FrameDesc d = descs[blockIdx.y];  // block-uniform: scalar loads
  if (!d.valid || frame_skipped<COND>(d.st)) return;
  const u32 evb = blockDim.x * TILE_EPT;
  const u32 nblk = (u32)((d.n + evb - 1) / evb);
  if constexpr (COND == 1) {
    // The redo node of a captured batch: launched with a FEW blocks per frame, which walk the frame's tiles when the frame's
    // attempt failed -- in the usual case (it held) the node costs a handful of blocks that read two words and return, not a
    // block per tile (each with the tiled kernel's LDS to allocate: 10 us per 60-frame replay)
    if (d.n == 0) {
      if (blockIdx.x == 0) scatter_empty_frame(d.st, sorted_mode);
      return;
    }
    for (u32 b = blockIdx.x; b < nblk; b += gridDim.x) {
      scatter_tiled_body<T, AOS, HAS_P, VIEW, VEC, KEY32>(d.x, d.y, (const T*)d.t, d.p, d.aos, d.n, 0ull, tb, d.st, 0u, 0ull, 0ull,
                                                          nullptr, d.key_frame, nullptr, w_ts, w_x, sorted_mode, b, nblk);
      __syncthreads();  // the next tile clears the LDS this one's flush has just read
    }
    return;
  }
  if (blockIdx.x >= nblk) {
    if (d.n == 0 && blockIdx.x == 0) scatter_empty_frame(d.st, sorted_mode);
    return;
  }
  scatter_tiled_body<T, AOS, HAS_P, VIEW, VEC, KEY32>(d.x, d.y, (const T*)d.t, d.p, d.aos, d.n, 0ull, tb, d.st, 0u, 0ull, 0ull,
                                                      nullptr, d.key_frame, nullptr, w_ts, w_x, sorted_mode, blockIdx.x, nblk);
}

// =====================================================================================================
// K2: frame kernels.
// =====================================================================================================
__device__ const u32 kTurbo[256] = {
#include "turbo_lut.inc"
};

struct PixelOut {
  float depth;
  u32 bgr;  // byte0 = B, byte1 = G, byte2 = R
};

// A5 + A6 + A7 for one pixel of the final disparity frame
__device__ inline PixelOut disparity_pixel(float d, double p03, float z_near, float z_far) {
  PixelOut o;
  // disp_to_depth.py:58-61 -- P2 is float64, so the divide is FP64; max(., 1e-9); stored as f32
  o.depth = d == 0.0f ? 0.0f : (float)fmax(p03 / (double)d, 1e-9);
  // disp_to_depth.py:12-20 -- clamp, normalise in f32; `* 255` is f32 x int64 -> f64 under Numba; trunc
  u32 u8 = 0;
  if (o.depth != 0.0f) {
    const float range = z_far - z_near;
    const float c = fmaxf(fminf(o.depth, z_far), z_near);
    const float q = (c - z_near) / range;
    u8 = (u32)(int)((double)q * 255.0) & 0xff;
  }
  // disp_to_depth.py:24-43 -- Turbo, undefined depth (u8 == 0) painted white
  o.bgr = u8 == 0 ? 0x00ffffffu : kTurbo[u8];
  return o;
}

// The fused path only ever sees integer disparities 0..65535 (low 16 bits of a key), and A5-A7 are a pure function of
// the disparity for fixed P2[0,3] / z_near / z_far: tabulate it once per handle with the very same device function
// (bit-identical by construction) -- K2 then replaces an FP64 divide, an f32 divide and the Turbo lookup by one
// 8-byte gather from a table whose live part (disparities < rect_w) sits in L1/L2.
__global__ __launch_bounds__(BLOCK) void k_build_dlut(uint2* __restrict__ dlut, double p03, float z_near, float z_far) {
  const u32 d = blockIdx.x * BLOCK + threadIdx.x;
  if (d < 65536u) {
    const PixelOut o = disparity_pixel((float)d, p03, z_near, z_far);
    dlut[d] = make_uint2(__float_as_uint(o.depth), o.bgr);
  }
}

// cooperative, coalesced store of BLOCK pixels' BGR bytes (3 B each) through LDS
__device__ inline void store_bgr_block(uint8_t* __restrict__ bgr, u64 first_pixel, u64 n_pixels, u32 v) {
  __shared__ __attribute__((aligned(16))) uint8_t s[BLOCK * 3];
  s[threadIdx.x * 3 + 0] = (uint8_t)(v & 0xff);
  s[threadIdx.x * 3 + 1] = (uint8_t)((v >> 8) & 0xff);
  s[threadIdx.x * 3 + 2] = (uint8_t)((v >> 16) & 0xff);
  __syncthreads();
  const u64 remaining = n_pixels - first_pixel;
  uint8_t* dst = bgr + first_pixel * 3;  // BLOCK*3 = 768 B per block -> 4-byte aligned
  if (remaining >= BLOCK) {
    if (threadIdx.x < BLOCK * 3 / 4) reinterpret_cast<u32*>(dst)[threadIdx.x] = reinterpret_cast<u32*>(s)[threadIdx.x];
  } else {
    for (u32 i = threadIdx.x; i < remaining * 3; i += BLOCK) dst[i] = s[i];
  }
}

struct KeyCells {  // cells of the packed-key frame written by K1 (projector view: column-major)
  static constexpr bool keyed = true;
  const u64* f;
  u32 tag;
  __device__ float decode(u64 k) const { return (u32)(k >> KEY_TAG_SHIFT) == tag ? (float)(u32)(k & 0xffff) : 0.0f; }
  __device__ float get(u32 i) const { return decode(f[i]); }
  __device__ float at(const DevTables& tb, int col, int row) const { return decode(f[(u32)col * (u32)tb.rect_h + (u32)row]); }
};
struct F32Cells {  // a plain row-major f32 disparity frame (stage API)
  static constexpr bool keyed = false;
  const float* f;
  __device__ float get(u32 i) const { return f[i]; }
  __device__ float at(const DevTables& tb, int col, int row) const { return f[(u32)row * (u32)tb.rect_w + (u32)col]; }
};

// dilate(7x7) o remap(nearest) composed: out[v,u] = max over the 7x7 window centred on map[v,u] of the
// rectified frame, 0 when the map points outside it; window cells outside the image are ignored.
template <typename Cells>
__device__ inline float dilated_remap(const Cells& cells, const DevTables& tb, u32 pixel) {
  const u32 m = tb.pmap[pixel];
  const int mx = (int)(short)(m & 0xffff), my = (int)(short)(m >> 16);
  if (mx < 0 || mx >= tb.rect_w || my < 0 || my >= tb.rect_h) return 0.0f;  // BORDER_CONSTANT 0
  float best = 0.0f;  // disparities are >= 0, so ignoring the border == zero padding
  const int y0 = max(my - 3, 0), y1 = min(my + 3, tb.rect_h - 1);
  const int x0 = max(mx - 3, 0), x1 = min(mx + 3, tb.rect_w - 1);
  for (int xx = x0; xx <= x1; ++xx) {
#pragma unroll 7
    for (int yy = y0; yy <= y1; ++yy) best = fmaxf(best, cells.at(tb, xx, yy));
  }
  return best;
}

// projector view: one thread per projector pixel.  MODE 0: packed-key frame -> depth + BGR (fused hot
// path); MODE 1: f32 frame -> f32 remapped disparity (stage A4)
template <typename Cells, int MODE>
__global__ __launch_bounds__(BLOCK) void k_frame_proj(Cells cells, DevTables tb, SlotState* st, u32 tag_override,
                                                      float* __restrict__ out_f32, uint8_t* __restrict__ bgr) {
  const u64 n_pixels = (u64)tb.proj_w * tb.proj_h;
  const u64 pixel = (u64)blockIdx.x * BLOCK + threadIdx.x;
  if constexpr (MODE == 0) {
    const u32 tag = tag_override ? tag_override : st->tag_a;
    cells.tag = tag;
    if (!tag_override && blockIdx.x == 0 && threadIdx.x < CNT_SLOTS) {  // re-arm the next frame's counters
      u32* c = st->cnt[(tag & 1) ^ 1][threadIdx.x];
      c[0] = c[1] = c[2] = c[3] = 0;
      if (threadIdx.x == 0) {
        st->tag_b = tag;
        if (u32* hf = st->host_flags) host_flag_store(hf + 1, tag);
      }
    }
  }
  float d = 0.0f;
  if (pixel < n_pixels) d = dilated_remap(cells, tb, (u32)pixel);
  if constexpr (MODE == 1) {
    if (pixel < n_pixels) out_f32[pixel] = d;
  } else {
    const PixelOut o = disparity_pixel(d, tb.p03, tb.z_near, tb.z_far);
    if (out_f32 && pixel < n_pixels) out_f32[pixel] = o.depth;
    if (bgr) store_bgr_block(bgr, (u64)blockIdx.x * BLOCK, n_pixels, o.bgr);
  }
}


// K2 (tiled, projector view, fused path): one block of 16 x 16 threads = a 32 x 16 tile of projector pixels, two per thread
// (K2_PPT: a K2 wave is a chain of dependent round trips -- descriptor, tile record, patch, table -- and what it costs is
// resident waves x lifetime, so each wave carries two pixels' worth of loads through that chain; measured against 1, 3, 4
// pixels per thread and 64x8 / 16x32 tiles: DESIGN.md section 3).  Their map targets span a (32*sx+6) x (16*sy+6) patch of
// the rectified key frame (sx, sy ~ 2.75):
//   0. the patch rectangle of the tile and every pixel's offset into it are static (the maps never change): they come
//      from tables built once in xm_create (k_build_k2_tables), so the loads below start right after one uniform load;
//   1. the patch is loaded ONCE into LDS as u16 disparities (stale tags -> 0, cells outside the frame -> 0); the key
//      frame is column-major, so the patch is `cols` contiguous runs -> paired 16-byte loads, 8 in flight per thread;
//   2. the 7-tap max along rows is taken once per patch cell with 16-byte LDS reads (separable max filter);
//   3. every pixel then needs 7 LDS reads (one per window column) instead of 49.
// PMC on the 49-tap version: SQ_LDS_IDX_ACTIVE 3.1 M cycles / dispatch -- it was LDS-bound.
// Falls back to global reads when the patch does not fit (wild maps).
#ifndef XM_K2_TILE_MAX
#define XM_K2_TILE_MAX 10240
#endif
#ifndef XM_K2_TX
#define XM_K2_TX 16
#define XM_K2_TY 16
#endif
#ifdef XM_ABLATE  // experiments (tools/k2_timeline.py): s_memtime stamps of thread 0 of 64 tiles in the middle of frame 30's K2
#define XM_K2STAMP(ph) do { if (threadIdx.x == 0 && blockIdx.z == 30 && blockIdx.y == 15 && blockIdx.x < 40) g_timeline[blockIdx.x][9 + (ph)] = __builtin_amdgcn_s_memtime(); } while (0)  /* columns 9..15: K1's stamps keep 0..8 */
#else
#define XM_K2STAMP(ph) do { } while (0)
#endif
// pixels per thread (template parameter PPT of the K2 kernels: 2 for launches that fill the chip, 1 for a lone frame): a block's
// tile is K2_TX * PPT x K2_TY pixels, thread (tx, ty) takes columns tx + j * K2_TX
constexpr int K2_TX = XM_K2_TX, K2_TY = XM_K2_TY, K2_TILE_MAX = XM_K2_TILE_MAX;  // at most 20 KB of u16 per block (the rig's
                                                                    // largest patch sizes the dynamic LDS: 10.5 KB at C-1M)

__device__ inline uint16_t key_disp(u64 k, u32 tag) { return (u32)(k >> KEY_TAG_SHIFT) == tag ? (uint16_t)(k & 0xffff) : (uint16_t)0; }

// One-off (xm_create): per K2 tile, the bounding box of its pixels' map targets (+3 cells of dilate margin, rows starting
// on an even row and padded to 8) and, per pixel, where its window starts inside that patch.  The maps are static, so
// K2 no longer decodes the map, reduces a bounding box over the block and synchronises before it can issue its loads.
template <int PPT>
__global__ __launch_bounds__(K2_TX* K2_TY) void k_build_k2_tables(DevTables tb, int4* __restrict__ tiles,
                                                                 u32* __restrict__ pix) {
  constexpr int K2_PPT = PPT, K2_TW = K2_TX * PPT;
  constexpr int NT = K2_TX * K2_TY, NW = NT / 64;
  __shared__ int s_box[NW][4];
  const int tid = threadIdx.x, tx = tid % K2_TX, ty = tid / K2_TX;
  const int v = blockIdx.y * K2_TY + ty;
  int u[K2_PPT], mx[K2_PPT], my[K2_PPT];
  bool in_img[K2_PPT], valid[K2_PPT];
  int x0 = 0x7fffffff, x1 = -0x7fffffff, y0 = 0x7fffffff, y1 = -0x7fffffff;
#pragma unroll
  for (int j = 0; j < K2_PPT; ++j) {
    u[j] = blockIdx.x * K2_TW + tx + j * K2_TX;
    in_img[j] = u[j] < tb.proj_w && v < tb.proj_h;
    mx[j] = my[j] = 0;
    valid[j] = false;
    if (in_img[j]) {
      const u32 m = tb.pmap[(u32)v * (u32)tb.proj_w + (u32)u[j]];
      mx[j] = (int)(short)(m & 0xffff);
      my[j] = (int)(short)(m >> 16);
      valid[j] = mx[j] >= 0 && mx[j] < tb.rect_w && my[j] >= 0 && my[j] < tb.rect_h;
    }
    if (valid[j]) {
      x0 = min(x0, mx[j]);
      x1 = max(x1, mx[j]);
      y0 = min(y0, my[j]);
      y1 = max(y1, my[j]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    x0 = min(x0, __shfl_xor(x0, o, 64));
    x1 = max(x1, __shfl_xor(x1, o, 64));
    y0 = min(y0, __shfl_xor(y0, o, 64));
    y1 = max(y1, __shfl_xor(y1, o, 64));
  }
  if ((tid & 63) == 0) {
    s_box[tid >> 6][0] = x0;
    s_box[tid >> 6][1] = x1;
    s_box[tid >> 6][2] = y0;
    s_box[tid >> 6][3] = y1;
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    x0 = min(x0, s_box[w][0]);
    x1 = max(x1, s_box[w][1]);
    y0 = min(y0, s_box[w][2]);
    y1 = max(y1, s_box[w][3]);
  }
  int4 rec = make_int4(0, 0, 0, 0);
  u32 off[K2_PPT];
#pragma unroll
  for (int j = 0; j < K2_PPT; ++j) off[j] = ~0u;
  if (x1 >= x0) {
    // rows start on a multiple of 8: 16-byte loads of 2 (u64 keys), 4 (u32 keys) or 8 (u16 disparities) rows
    const int bx = x0 - 3, by = (y0 - 3) & ~7;
    const int cols = x1 + 3 - bx + 1, rows = y1 + 3 - by + 1, rows_p = (rows + 7) & ~7;
    const bool fits = cols * rows_p <= K2_TILE_MAX;
    rec = make_int4(bx, by, fits ? cols : -1, rows_p);
#pragma unroll
    for (int j = 0; j < K2_PPT; ++j)
      if (valid[j] && fits) off[j] = (u32)((mx[j] - 3 - bx) * rows_p + (my[j] - 3 - by));
  }
  if (tid == 0) tiles[blockIdx.y * gridDim.x + blockIdx.x] = rec;
#pragma unroll
  for (int j = 0; j < K2_PPT; ++j)
    if (in_img[j]) pix[(u32)v * (u32)tb.proj_w + (u32)u[j]] = off[j];
}

// 7-tap max along 8 consecutive rows of a patch column: inputs e[0..13] = the 16 bytes a (rows r .. r+7) and b (rows r+8 .. r+15),
// outputs o[j] = max(e[j] .. e[j+6]), j = 0..7.  Packed 16-bit arithmetic (v_pk_max_u16): P_k = (e[2k], e[2k+1]) as loaded,
// S_k = (e[2k+1], e[2k+2]) by a 16-bit funnel shift; (o[2j], o[2j+1]) = max(P_j, S_j, P_j+1, S_j+1, P_j+2, S_j+2, P_j+3):
// 24 instructions instead of 72 for unpack + 48 scalar maxima + pack
__device__ __forceinline__ uint4 k2_rowmax8(const uint4 a, const uint4 b) {
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  const auto pk = [](u32 v) { u16x2 r; __builtin_memcpy(&r, &v, 4); return r; };
  const auto up = [](u16x2 v) { u32 r; __builtin_memcpy(&r, &v, 4); return r; };
  const auto mx = [](u16x2 x, u16x2 y) { return __builtin_elementwise_max(x, y); };
  const u32 P[7] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z};
  u16x2 M[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) M[k] = mx(pk(P[k]), pk(__builtin_amdgcn_alignbit(P[k + 1], P[k], 16)));
  uint4 w;
  w.x = up(mx(mx(M[0], M[1]), mx(M[2], pk(P[3]))));
  w.y = up(mx(mx(M[1], M[2]), mx(M[3], pk(P[4]))));
  w.z = up(mx(mx(M[2], M[3]), mx(M[4], pk(P[5]))));
  w.w = up(mx(mx(M[3], M[4]), mx(M[5], pk(P[6]))));
  return w;
}

// blk_lin / grid_x / grid_y = linear block index inside the frame's tile grid and that grid's shape
// FMT: what `keys` points at -- 0: the 64-bit packed-key frame; 1: the compact 32-bit key frame of the verified-sorted path
// (see key32_tag); 2: a plain u16 disparity frame, no tags (sharded frames after reduce-scatter + all-gather, xm_shard_finish_u16)
template <int FMT = 0, int PPT = 2>
__device__ __forceinline__ void frame_proj_tiled_body(const u64* __restrict__ keys, const DevTables& tb, SlotState* st,
                                                      u32 tag_override, const unsigned char* __restrict__ dirty,
                                                      const ulonglong2* __restrict__ zero16, float* __restrict__ depth,
                                                      uint8_t* __restrict__ bgr, int tile_cap, const u32 blk_lin,
                                                      const u32 grid_x, const u32 grid_y, const int4* rec_pre = nullptr) {
  // Dynamic LDS sized to the largest patch of THIS rig (tile_cap cells, a multiple of 8, <= K2_TILE_MAX; set in xm_create):
  // how many blocks fit beside K1's 70 KB blocks on a CU is what bounds the pipelined frame rate, and the static
  // worst case was several times what C-1M's 94 x 56 patches need.
  constexpr bool KEY32 = FMT == 1, U16 = FMT == 2;
  constexpr int K2_PPT = PPT, K2_TW = K2_TX * PPT;
  extern __shared__ __attribute__((aligned(16))) uint16_t k2_lds[];
  uint16_t* tile = k2_lds;  // [tile_cap + 16]  (+16: the last 16-byte read may overrun)
  uint16_t* vmax = tile;  // the row maxima replace the patch IN PLACE: half the LDS per block = more blocks per CU
  constexpr int NT = K2_TX * K2_TY, NW = NT / 64;
  __shared__ __attribute__((aligned(16))) uint8_t s_bgr[K2_TY][K2_TW * 3];
  constexpr int FLAG_LINES = 8, FLAG_COLS = 128;  // patch columns x 128-byte lines per column (rows_p <= 96 -> <= 7 lines)
  __shared__ unsigned char s_live[FLAG_COLS * FLAG_LINES];
  const int tid = threadIdx.x, tx = tid & (K2_TX - 1), ty = tid / K2_TX;
  XM_K2STAMP(0);
  // XCD-aware tile order (see xcd_contiguous): each XCD takes a contiguous run of the tile raster, so the halos that
  // neighbouring tiles share (3 of 22 patch columns each side, boundary cache lines above/below) hit in its own L2.
  const u32 lin_tile = xcd_contiguous(blk_lin, grid_x * grid_y);
  const u32 tile_y = lin_tile / grid_x, tile_x = lin_tile - tile_y * grid_x;
  const u32 tag = tag_override ? tag_override : st->tag_a;  // first needed when the patch is decoded
  const int v = tile_y * K2_TY + ty;
  // the tile's patch rectangle and the pixel's offset into it were computed once in xm_create (k_build_k2_tables)
  const int4 rec = rec_pre ? *rec_pre : (PPT == 1 ? tb.k2_tiles1 : tb.k2_tiles)[lin_tile];  // block-uniform
  const u32* __restrict__ k2_pix = PPT == 1 ? tb.k2_pix1 : tb.k2_pix;
  bool in_img[K2_PPT];
  u32 pix_i[K2_PPT], poff[K2_PPT];
#pragma unroll
  for (int j = 0; j < K2_PPT; ++j) {
    const int u = tile_x * K2_TW + tx + j * K2_TX;
    in_img[j] = u < tb.proj_w && v < tb.proj_h;
    pix_i[j] = __umul24((u32)v, (u32)tb.proj_w) + (u32)u;  // (24-bit multiplies are full rate, v_mul_lo_u32 a quarter)
    poff[j] = in_img[j] ? k2_pix[pix_i[j]] : ~0u;
  }
  // generic path only; the tiled path tests poff where it needs it (after the patch loads are out: testing it here put a
  // full wait for this load in front of them)
  int x0 = 0, x1 = -1;
  if (rec.z < 0) {  // patch too large for LDS (wild map): generic path needs the map entries themselves
    x1 = 0;         // "some pixel maps into the frame": take the branch below, which falls through to the global reads
  } else if (rec.z > 0) {
    x1 = 0;
  }
  float d[K2_PPT];  // generic path (a patch too large for LDS)
  u32 di[K2_PPT];   // tiled path: the integer disparity itself (no int -> float -> int round trip: conversions are quarter rate)
#pragma unroll
  for (int j = 0; j < K2_PPT; ++j) {
    d[j] = 0.0f;
    di[j] = 0;
  }
  if (x1 >= x0) {  // at least one pixel of the tile maps into the frame
    const int bx = rec.x, by = rec.y;                    // patch origin (rows start on an even row: 16-byte aligned pairs)
    const int cols = rec.z, rows_p = rec.w;              // column stride in LDS: 16-byte aligned runs
    if (cols > 0) {
      constexpr int UN = 8;
      // which 128-byte lines of the patch carry keys of THIS frame?  (flag bytes written by K1; all lines when no flags)
      const bool use_flags = dirty != nullptr && cols <= FLAG_COLS && rows_p <= 16 * (FLAG_LINES - 1);
      if (use_flags) {
        const unsigned char want = dirty_byte(tag);
        const u32 n_lines = ((u32)tb.rect_w * (u32)tb.rect_h + 15u) >> 4;
        for (int i = tid; i < cols * FLAG_LINES; i += NT) {
          const int c = i / FLAG_LINES, j = i - c * FLAG_LINES, gx = bx + c;
          unsigned char live = 0;
          if (gx >= 0 && gx < tb.rect_w) {
            const int first_cell = gx * tb.rect_h + max(by, 0);  // first in-frame cell of this patch column
            const u32 line = ((u32)first_cell >> 4) + (u32)j;
            if (line < n_lines) live = dirty[line] == want;
          }
          s_live[i] = live;
        }
        __syncthreads();
      }
      if constexpr (U16) {  // plain disparities: 8-byte loads of 4 rows copied straight into the LDS patch
        const uint16_t* d16 = reinterpret_cast<const uint16_t*>(keys);
        const bool interior = bx >= 0 && by >= 0 && bx + cols <= tb.rect_w && by + rows_p <= tb.rect_h && (tb.rect_h & 3) == 0;
        const int g0 = by >> 3, sh0 = tb.shear_bias + ((g0 * tb.shear_m) >> 12);  // the patch's first 8-row group and its shear
        if (interior && (tb.rect_h & 7) == 0) {
          // 16-byte loads of 8 rows (the patch starts on a multiple of 8 rows and rows_p is one), copied as they are: LDS quad
          // index == patch (column, row octet) index.  A 50 x 56 patch is 350 quads: two loads per thread.
          const int oct = rows_p >> 3, total = cols * oct;
          if (oct <= 8) {
            // patches of <= 64 rows: thread slot s -> (column s >> 3, row octet s & 7) by shift and mask (slots with an octet
            // past the patch idle); K2 is issue bound and the divide by `oct` below is ~12 instructions per load
            const int nslot = cols << 3;
            for (int s0 = tid; s0 < nslot; s0 += 2 * NT) {
              uint4 k[2];
              bool has[2];
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const int sj = s0 + j * NT, c = sj >> 3, ro = sj & 7;
                has[j] = sj < nslot && ro < oct;
                // (the frame is sheared by whole columns per 8-row group -- frame16_col; sh0 / shear_m are 0 on rigs that are not slanted)
                const int cs = bx + c + (((g0 + ro) * tb.shear_m) >> 12);
                k[j] = *reinterpret_cast<const uint4*>(d16 + (has[j] ? __umul24((u32)(cs + tb.shear_bias), (u32)tb.rect_h) + (u32)(by + 8 * ro)
                                                                     : __umul24((u32)(bx + sh0), (u32)tb.rect_h) + (u32)by));
              }
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const int sj = s0 + j * NT;
                if (has[j]) reinterpret_cast<uint4*>(tile)[__mul24(sj >> 3, oct) + (sj & 7)] = k[j];
              }
            }
          } else {
          const float inv_o = __builtin_amdgcn_rcpf((float)oct);  // (approximate: the +-1 fix-ups below absorb it)
          for (int i0 = tid; i0 < total; i0 += 2 * NT) {
            uint4 k[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int i = min(i0 + j * NT, total - 1);
              int c = (int)((float)i * inv_o), ro = i - __mul24(c, oct);
              if (ro < 0) { c -= 1; ro += oct; }
              if (ro >= oct) { c += 1; ro -= oct; }
              k[j] = *reinterpret_cast<const uint4*>(d16 + __umul24((u32)(bx + c + tb.shear_bias + (((g0 + ro) * tb.shear_m) >> 12)), (u32)tb.rect_h) +
                                                     (u32)(by + 8 * ro));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
              if (i0 + j * NT < total) reinterpret_cast<uint4*>(tile)[i0 + j * NT] = k[j];
          }
          }
        } else if (interior) {
          const int quarter = rows_p >> 2, total = cols * quarter;
          const float inv_q = __builtin_amdgcn_rcpf((float)quarter);
          for (int i0 = tid; i0 < total; i0 += 4 * NT) {
            uint2 k[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int i = min(i0 + j * NT, total - 1);
              int c = (int)((float)i * inv_q), rq = i - __mul24(c, quarter);
              if (rq < 0) { c -= 1; rq += quarter; }
              if (rq >= quarter) { c += 1; rq -= quarter; }
              k[j] = *reinterpret_cast<const uint2*>(d16 + __umul24((u32)frame16_col(tb, bx + c, by + 4 * rq), (u32)tb.rect_h) + (u32)(by + 4 * rq));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (i0 + j * NT < total) reinterpret_cast<uint2*>(tile)[i0 + j * NT] = k[j];
          }
        } else {
          const int total = cols * rows_p;
          const float inv_rows = 1.0f / (float)rows_p;
          for (int i = tid; i < total; i += NT) {
            int c = (int)((float)i * inv_rows), r = i - c * rows_p;
            if (r < 0) { c -= 1; r += rows_p; }
            if (r >= rows_p) { c += 1; r -= rows_p; }
            const int gx = bx + c, gy = by + r;
            const bool inside = gx >= 0 && gx < tb.rect_w && gy >= 0 && gy < tb.rect_h;
            const int cx = min(max(gx, 0), tb.rect_w - 1), cy = min(max(gy, 0), tb.rect_h - 1);
            const uint16_t v = d16[(u32)frame16_col(tb, cx, cy) * (u32)tb.rect_h + (u32)cy];
            tile[i] = inside ? v : (uint16_t)0;
          }
        }
      } else if constexpr (KEY32) {
        const u32* keys32 = reinterpret_cast<const u32*>(keys);
        const u32 tag4 = key32_tag(tag);
        const bool interior = bx >= 0 && by >= 0 && bx + cols <= tb.rect_w && by + rows_p <= tb.rect_h;  // rect_h % 4 == 0 (host)
        if (interior) {  // 16-byte loads of 4 consecutive rows, (column, row quad) advanced incrementally
          const int quarter = rows_p >> 2, total = cols * quarter;
          const int dq = NT / quarter, dr = NT - dq * quarter;
          int c_i = (int)((float)tid * (1.0f / (float)quarter)), rq_i = tid - c_i * quarter;
          if (rq_i < 0) { c_i -= 1; rq_i += quarter; }
          if (rq_i >= quarter) { c_i += 1; rq_i -= quarter; }
          u32 cell = (u32)(bx + c_i) * (u32)tb.rect_h + (u32)(by + 4 * rq_i);
          const u32 cell_origin = (u32)bx * (u32)tb.rect_h + (u32)by;
          const u32 dcell = (u32)dq * (u32)tb.rect_h + 4u * (u32)dr, carry = (u32)tb.rect_h - 4u * (u32)quarter;
          auto pass = [&](auto un_tag) {
            constexpr int UL = decltype(un_tag)::value;
            for (int i0 = tid; i0 < total; i0 += UL * NT) {
              uint4 k[UL];
#pragma unroll
              for (int j = 0; j < UL; ++j) {
                k[j] = *reinterpret_cast<const uint4*>(keys32 + (i0 + j * NT < total ? cell : cell_origin));
                cell += dcell;
                rq_i += dr;
                if (rq_i >= quarter) { rq_i -= quarter; cell += carry; }
              }
#pragma unroll
              for (int j = 0; j < UL; ++j) {
                const int i = i0 + j * NT;
                if (i < total)
                  reinterpret_cast<uint2*>(tile)[i] =
                      make_uint2((u32)key_disp32(k[j].x, tag4) | ((u32)key_disp32(k[j].y, tag4) << 16),
                                 (u32)key_disp32(k[j].z, tag4) | ((u32)key_disp32(k[j].w, tag4) << 16));
              }
            }
          };
          const int need = (total + NT - 1) / NT;
          if (need <= 1) pass(std::integral_constant<int, 1>{});
          else if (need <= 2) pass(std::integral_constant<int, 2>{});
          else if (need <= 3) pass(std::integral_constant<int, 3>{});
          else pass(std::integral_constant<int, 4>{});
        } else {  // patches that stick out of the frame (tiles along the border): cell by cell
          const int total = cols * rows_p;
          const float inv_rows = 1.0f / (float)rows_p;
          for (int i = tid; i < total; i += NT) {
            int c = (int)((float)i * inv_rows), r = i - c * rows_p;
            if (r < 0) { c -= 1; r += rows_p; }
            if (r >= rows_p) { c += 1; r -= rows_p; }
            const int gx = bx + c, gy = by + r;
            const bool inside = gx >= 0 && gx < tb.rect_w && gy >= 0 && gy < tb.rect_h;
            const u32 kk = keys32[(u32)min(max(gx, 0), tb.rect_w - 1) * (u32)tb.rect_h + (u32)min(max(gy, 0), tb.rect_h - 1)];
            tile[i] = inside ? key_disp32(kk, tag4) : (uint16_t)0;
          }
        }
      } else if ((tb.rect_h & 1) == 0) {
        const int half = rows_p >> 1, total = cols * half;
        // (c, rp) = divmod(i, half) advanced incrementally: i -> i + NT is (c + dq, rp + dr) with one carry
        const int dq = NT / half, dr = NT - dq * half;
        int c_i = (int)((float)tid * (1.0f / (float)half)), rp_i = tid - c_i * half;  // no integer divide
        if (rp_i < 0) { c_i -= 1; rp_i += half; }
        if (rp_i >= half) { c_i += 1; rp_i -= half; }
        // K2 is bound by instruction issue, and most tiles' patches lie entirely inside the frame: those take a lean
        // loader -- the cell index advances incrementally with the (column, row pair) carry, no clamps, no inside tests --
        // unrolled to what the patch needs (a 46 x 24 patch is 2.2 sixteen-byte loads per thread, not 8).
        const bool interior = !use_flags && bx >= 0 && by >= 0 && bx + cols <= tb.rect_w && by + rows_p <= tb.rect_h;
        if (interior) {
          u32 cell = (u32)(bx + c_i) * (u32)tb.rect_h + (u32)(by + 2 * rp_i);
          const u32 cell_origin = (u32)bx * (u32)tb.rect_h + (u32)by;
          const u32 dcell = (u32)dq * (u32)tb.rect_h + 2u * (u32)dr, carry = (u32)tb.rect_h - 2u * (u32)half;
          auto pass = [&](auto un_tag) {
            constexpr int UL = decltype(un_tag)::value;
            for (int i0 = tid; i0 < total; i0 += UL * NT) {
              ulonglong2 k[UL];
#pragma unroll
              for (int j = 0; j < UL; ++j) {
                k[j] = *reinterpret_cast<const ulonglong2*>(keys + (i0 + j * NT < total ? cell : cell_origin));
                cell += dcell;
                rp_i += dr;
                if (rp_i >= half) { rp_i -= half; cell += carry; }
              }
#pragma unroll
              for (int j = 0; j < UL; ++j) {
                const int i = i0 + j * NT;
                if (i < total)
                  reinterpret_cast<u32*>(tile)[i] = (u32)key_disp(k[j].x, tag) | ((u32)key_disp(k[j].y, tag) << 16);
              }
            }
          };
          const int need = (total + NT - 1) / NT;
          if (need <= 2) pass(std::integral_constant<int, 2>{});
          else if (need <= 3) pass(std::integral_constant<int, 3>{});
          else if (need <= 4) pass(std::integral_constant<int, 4>{});
          else pass(std::integral_constant<int, 8>{});
        } else
        for (int i0 = tid; i0 < total; i0 += UN * NT) {
          ulonglong2 k[UN];
          bool inside[UN];
#pragma unroll
          for (int j = 0; j < UN; ++j) {  // unconditional loads (coordinates clamped into the frame), select afterwards
            const int gx = bx + c_i, gy = by + 2 * rp_i;
            inside[j] = gx >= 0 && gx < tb.rect_w && gy >= 0 && gy < tb.rect_h;
            const int cx = min(max(gx, 0), tb.rect_w - 1), cy = min(max(gy, 0), tb.rect_h - 2);
            const u32 cell = (u32)cx * (u32)tb.rect_h + (u32)cy;
            const ulonglong2* src = reinterpret_cast<const ulonglong2*>(keys + cell);
            if (use_flags) {  // clean line: read the 16-byte zero constant instead (L2-hot, no HBM traffic)
              const int first_cell = cx * tb.rect_h + max(by, 0);
              const int jl = (int)(cell >> 4) - (first_cell >> 4);
              const bool live = inside[j] && (u32)jl < (u32)FLAG_LINES && s_live[min(c_i, FLAG_COLS - 1) * FLAG_LINES + max(min(jl, FLAG_LINES - 1), 0)];
              inside[j] = live;
              src = live ? src : zero16;
            }
            k[j] = *src;
            c_i += dq;
            rp_i += dr;
            if (rp_i >= half) { rp_i -= half; c_i += 1; }
          }
#pragma unroll
          for (int j = 0; j < UN; ++j) {
            const int i = i0 + j * NT;
            if (i < total) {
              const u32 pr = inside[j] ? ((u32)key_disp(k[j].x, tag) | ((u32)key_disp(k[j].y, tag) << 16)) : 0u;
              reinterpret_cast<u32*>(tile)[i] = pr;  // tile[c * rows_p + 2 * rp] (+1): i == c * half + rp
            }
          }
        }
      } else {
        const int total = cols * rows_p;
        const float inv_rows = 1.0f / (float)rows_p;
        for (int i0 = tid; i0 < total; i0 += UN * NT) {
          u64 k[UN];
          bool inside[UN];
#pragma unroll
          for (int j = 0; j < UN; ++j) {
            const int i = min(i0 + j * NT, total - 1);
            int c = (int)((float)i * inv_rows), r = i - c * rows_p;
            if (r < 0) { c -= 1; r += rows_p; }
            if (r >= rows_p) { c += 1; r -= rows_p; }
            const int gx = bx + c, gy = by + r;
            inside[j] = gx >= 0 && gx < tb.rect_w && gy >= 0 && gy < tb.rect_h;
            const int cx = min(max(gx, 0), tb.rect_w - 1), cy = min(max(gy, 0), tb.rect_h - 1);
            k[j] = keys[(u32)cx * (u32)tb.rect_h + (u32)cy];
          }
#pragma unroll
          for (int j = 0; j < UN; ++j) {
            const int i = i0 + j * NT;
            if (i < total) tile[i] = inside[j] ? key_disp(k[j], tag) : (uint16_t)0;
          }
        }
      }
      XM_K2STAMP(1);
      __syncthreads();
      XM_K2STAMP(2);
      {  // 7-tap max along the rows of every patch column: 8 outputs per task from 14 inputs (two 16-byte LDS reads).
        // (Tried: taking them in registers straight from two global loads per thread, no tile buffer and one barrier less --
        // the kernel alone is as fast, the pipelined frame rate 4 % lower: twice the vector-memory requests.)
        // Task t covers tile[t*8 .. t*8+7] (c*rows_p + 8*seg) and reads 6 cells of task t + 1.  In place: a chunk of 4 NT
        // consecutive tasks is computed into registers, a barrier, then stored over its own inputs; the next chunk's inputs
        // lie behind everything this one wrote.
        const int nseg = rows_p >> 3, tasks = cols * nseg;
        constexpr int CH = 4;
        for (int t0 = 0; t0 < tasks; t0 += CH * NT) {  // (block-uniform trip count)
          uint4 w[CH];
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            const int t = t0 + j * NT + tid;
            if (t < tasks)
              w[j] = k2_rowmax8(*reinterpret_cast<const uint4*>(tile + t * 8), *reinterpret_cast<const uint4*>(tile + t * 8 + 8));
          }
          __syncthreads();
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            const int t = t0 + j * NT + tid;
            if (t < tasks) *reinterpret_cast<uint4*>(vmax + t * 8) = w[j];
          }
        }
      }
      XM_K2STAMP(3);
      __syncthreads();
      XM_K2STAMP(4);
#pragma unroll
      for (int q = 0; q < K2_PPT; ++q)
        if (poff[q] != ~0u) {
          const uint16_t* p = vmax + poff[q];
          u32 best = 0;
#pragma unroll
          for (int j = 0; j < 7; ++j) best = max(best, (u32)p[j * rows_p]);
          di[q] = best;
        }
    } else {
      for (int q = 0; q < K2_PPT; ++q) {
        if (!in_img[q]) continue;
        const u32 m = tb.pmap[pix_i[q]];
        const int mx = (int)(short)(m & 0xffff), my = (int)(short)(m >> 16);
        if (!(mx >= 0 && mx < tb.rect_w && my >= 0 && my < tb.rect_h)) continue;  // BORDER_CONSTANT 0
        const int ya = max(my - 3, 0), yb = min(my + 3, tb.rect_h - 1), xa = max(mx - 3, 0), xb = min(mx + 3, tb.rect_w - 1);
        float dq = 0.0f;
        if constexpr (U16) {
          const uint16_t* d16 = reinterpret_cast<const uint16_t*>(keys);
          for (int xx = xa; xx <= xb; ++xx)
            for (int yy = ya; yy <= yb; ++yy) dq = fmaxf(dq, (float)d16[(u32)frame16_col(tb, xx, yy) * (u32)tb.rect_h + (u32)yy]);
        } else if constexpr (KEY32) {
          const u32* keys32 = reinterpret_cast<const u32*>(keys);
          const u32 tag4 = key32_tag(tag);
          for (int xx = xa; xx <= xb; ++xx)
            for (int yy = ya; yy <= yb; ++yy) dq = fmaxf(dq, (float)key_disp32(keys32[(u32)xx * (u32)tb.rect_h + (u32)yy], tag4));
        } else {
          KeyCells cells{keys, tag};
          for (int xx = xa; xx <= xb; ++xx)
            for (int yy = ya; yy <= yb; ++yy) dq = fmaxf(dq, cells.at(tb, xx, yy));
        }
        d[q] = dq;
      }
    }
  }
  XM_K2STAMP(5);
  PixelOut o[K2_PPT];
#pragma unroll
  for (int q = 0; q < K2_PPT; ++q) {
    if (rec.z <= 0) di[q] = (u32)d[q];  // d is an integer disparity here (max of u16 key fields)
    // (byte offset off the table's base: a scalar-base + 32-bit-offset load instead of a 64-bit multiply-add per pixel)
    const uint2 e = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(tb.dlut) + ((di[q] & 0xffffu) << 3));
    o[q].depth = __uint_as_float(e.x);
    o[q].bgr = e.y;
  }
  if (!tag_override && lin_tile == 0 && tid < CNT_SLOTS) {  // re-arm the next frame's counters
    u32* c = st->cnt[(tag & 1) ^ 1][tid];
    c[0] = c[1] = c[2] = c[3] = 0;
    if (tid == 0) {
      st->tag_b = tag;  // time-sorted mode: K1 derived the tag from tag_b without touching it
      if (u32* hf = st->host_flags) host_flag_store(hf + 1, tag);
    }
  }
  if (depth) {
#pragma unroll
    for (int q = 0; q < K2_PPT; ++q)
      if (in_img[q]) depth[pix_i[q]] = o[q].depth;
  }
  if (bgr) {
    const bool full_rows = (tb.proj_w & 3) == 0 && (tile_x + 1) * K2_TW <= tb.proj_w;
    if (full_rows) {  // 3 * K2_TW contiguous bytes per tile row: assemble in LDS, store as dwords
#pragma unroll
      for (int q = 0; q < K2_PPT; ++q) {
        s_bgr[ty][(tx + q * K2_TX) * 3 + 0] = (uint8_t)(o[q].bgr & 0xff);
        s_bgr[ty][(tx + q * K2_TX) * 3 + 1] = (uint8_t)((o[q].bgr >> 8) & 0xff);
        s_bgr[ty][(tx + q * K2_TX) * 3 + 2] = (uint8_t)((o[q].bgr >> 16) & 0xff);
      }
      __syncthreads();
      constexpr int DW = K2_TW * 3 / 4;  // dwords per row
#pragma unroll
      for (int i0 = 0; i0 < K2_TY * DW; i0 += NT) {
        const int i = i0 + tid;
        if (i < K2_TY * DW) {
          const int r = i / DW, q = i - r * DW, vv = tile_y * K2_TY + r;
          if (vv < tb.proj_h)
            reinterpret_cast<u32*>(bgr + (size_t)((__umul24((u32)vv, (u32)tb.proj_w) + tile_x * K2_TW) * 3u))[q] =
                reinterpret_cast<const u32*>(&s_bgr[r][0])[q];
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < K2_PPT; ++q)
        if (in_img[q]) {
          uint8_t* b = bgr + (u64)pix_i[q] * 3;
          b[0] = (uint8_t)(o[q].bgr & 0xff);
          b[1] = (uint8_t)((o[q].bgr >> 8) & 0xff);
          b[2] = (uint8_t)((o[q].bgr >> 16) & 0xff);
        }
    }
  }
  XM_K2STAMP(6);
}

template <int FMT = 0, int PPT = 2>
__global__ __launch_bounds__(K2_TX* K2_TY) void k_frame_proj_tiled(const u64* __restrict__ keys, DevTables tb,
                                                                  SlotState* st, u32 tag_override,
                                                                  const unsigned char* __restrict__ dirty,
                                                                  const ulonglong2* __restrict__ zero16,
                                                                  float* __restrict__ depth, uint8_t* __restrict__ bgr,
                                                                  int tile_cap, int col_lo = 0, int col_hi = 0) {
  // every kernel argument in one scalar round trip (see k_scatter_tiled); never true
  if ((long long)((u64)keys | (u64)tb.k2_tiles | (u64)tb.k2_pix | (u64)tb.k2_tiles1 | (u64)tb.k2_pix1 | (u64)tb.dlut | (u64)tb.pmap | (u64)st | (u64)dirty |
                  (u64)zero16 | (u64)depth | (u64)bgr |
                  (u64)(long long)(tb.proj_w | tb.proj_h | tb.rect_w | tb.rect_h | (int)tag_override)) < 0)
    return;
  if (col_hi > col_lo) {  // band-sharded finish (xm_shard_finish_u16_band): only the tiles whose patch is centred on a frame column
                          // of [col_lo, col_hi) -- the rank's band of the merged frame; tiles without a patch go with column 0
    const int4 rec = (PPT == 1 ? tb.k2_tiles1 : tb.k2_tiles)[xcd_contiguous(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y)];
    const int c = rec.z > 0 ? rec.x + (rec.z >> 1) : 0;
    if (c < col_lo || c >= col_hi) return;
  }
  frame_proj_tiled_body<FMT, PPT>(keys, tb, st, tag_override, dirty, zero16, depth, bgr, tile_cap,
                                  blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, gridDim.y);
}

// multi-frame launch: grid = (tiles_x, tiles_y, frames)
template <int FMT = 0, int COND = 0, int PPT = 2>
__global__ __launch_bounds__(K2_TX* K2_TY) void k_frame_proj_tiled_batch(const FrameDesc* __restrict__ descs, DevTables tb,
                                                                        const ulonglong2* __restrict__ zero16,
                                                                        int tile_cap) {
  // The tile's patch record does not depend on the frame: its load goes out together with the frame descriptor's (a block of
  // K2 is a chain of dependent round trips -- descriptor, record, patch -- and 55 % of its lifetime at full occupancy is spent
  // before the patch has arrived: tools/k2_timeline.py).  The never-true test keeps the compiler from sinking the load
  // behind the branch.
  if constexpr (COND == 1) {  // redo node of a captured batch: a few blocks per frame walk the frame's tiles (see k_scatter_tiled_batch)
    const FrameDesc d = descs[blockIdx.z];
    if (!d.valid || frame_skipped<COND>(d.st)) return;
    const u32 gx = ((u32)tb.proj_w + K2_TX * PPT - 1) / (K2_TX * PPT), gy = ((u32)tb.proj_h + K2_TY - 1) / K2_TY;
    for (u32 b = blockIdx.y * gridDim.x + blockIdx.x; b < gx * gy; b += gridDim.x * gridDim.y) {
      frame_proj_tiled_body<FMT, PPT>(d.key_frame, tb, d.st, 0u, nullptr, zero16, d.depth, d.bgr, tile_cap, b, gx, gy);
      __syncthreads();  // the next tile's patch overwrites the LDS this one's pixels have just read
    }
    return;
  }
  const u32 blk_lin = blockIdx.y * gridDim.x + blockIdx.x;
  const int4 rec = (PPT == 1 ? tb.k2_tiles1 : tb.k2_tiles)[xcd_contiguous(blk_lin, gridDim.x * gridDim.y)];
  const FrameDesc d = descs[blockIdx.z];
  if (!d.valid || rec.w < 0) return;
  if (frame_skipped<COND>(d.st)) return;
  frame_proj_tiled_body<FMT, PPT>(d.key_frame, tb, d.st, 0u, nullptr, zero16, d.depth, d.bgr, tile_cap, blk_lin, gridDim.x,
                                  gridDim.y, &rec);
}

// camera view / plain per-pixel conversion of a frame of n_pixels cells -> depth + BGR
template <typename Cells>
__global__ __launch_bounds__(BLOCK) void k_frame_direct(Cells cells, u64 n_pixels, double p03, float z_near,
                                                        float z_far, SlotState* st, u32 tag_override, int use_tag,
                                                        const uint2* __restrict__ dlut, float* __restrict__ depth,
                                                        uint8_t* __restrict__ bgr) {
  const u64 pixel = (u64)blockIdx.x * BLOCK + threadIdx.x;
  if constexpr (Cells::keyed) {
    if (use_tag) {
      const u32 tag = tag_override ? tag_override : st->tag_a;
      cells.tag = tag;
      if (!tag_override && blockIdx.x == 0 && threadIdx.x < CNT_SLOTS) {
        u32* c = st->cnt[(tag & 1) ^ 1][threadIdx.x];
        c[0] = c[1] = c[2] = c[3] = 0;
        if (threadIdx.x == 0) {
          st->tag_b = tag;
          if (u32* hf = st->host_flags) host_flag_store(hf + 1, tag);
        }
      }
    }
  }
  float d = 0.0f;
  if (pixel < n_pixels) d = cells.get((u32)pixel);
  PixelOut o;
  if (Cells::keyed && dlut) {  // fused path: integer disparity -> tabulated A5-A7 (see k_build_dlut)
    const uint2 e = dlut[(u32)d & 0xffffu];
    o.depth = __uint_as_float(e.x);
    o.bgr = e.y;
  } else {
    o = disparity_pixel(d, p03, z_near, z_far);
  }
  if (depth && pixel < n_pixels) depth[pixel] = o.depth;
  if (bgr) store_bgr_block(bgr, (u64)blockIdx.x * BLOCK, n_pixels, o.bgr);
}

// multi-frame launch of the camera-view frame kernel: grid = (blocks per frame, frames)
__global__ __launch_bounds__(BLOCK) void k_frame_direct_batch(const FrameDesc* __restrict__ descs, u64 n_pixels,
                                                              const uint2* __restrict__ dlut) {
  const FrameDesc d = descs[blockIdx.y];
  if (!d.valid) return;
  SlotState* st = d.st;
  const u32 tag = st->tag_a;
  if (blockIdx.x == 0 && threadIdx.x < CNT_SLOTS) {
    u32* c = st->cnt[(tag & 1) ^ 1][threadIdx.x];
    c[0] = c[1] = c[2] = c[3] = 0;
    if (threadIdx.x == 0) {
      st->tag_b = tag;
      if (u32* hf = st->host_flags) host_flag_store(hf + 1, tag);
    }
  }
  const u64 pixel = (u64)blockIdx.x * BLOCK + threadIdx.x;
  u32 dsp = 0;
  if (pixel < n_pixels) dsp = key_disp(d.key_frame[pixel], tag);
  const uint2 e = dlut[dsp];
  if (d.depth && pixel < n_pixels) d.depth[pixel] = __uint_as_float(e.x);
  if (d.bgr) store_bgr_block(d.bgr, (u64)blockIdx.x * BLOCK, n_pixels, e.y);
}

// camera view on the compact key frame ((event index + 1) << 12 | disparity, 0 = no event): the pixel is zeroed once read, so
// the next frame of the slot starts from an empty frame without a clear of its own.  The frame is COLUMN-major
// (u32[cam_w][cam_h]: K1's flush walks consecutive rows of one window column -- 64 lanes = 256 contiguous bytes per atomic
// instruction instead of four 64-byte row pieces), the outputs are row-major: a block takes a 32 x 32-pixel tile, reads it
// along the columns, hands the 12-bit disparities over through LDS and writes rows (128 bytes of depth, 96 of BGR per row).
constexpr int CAM32_T = 32;
__device__ __forceinline__ void frame_cam32_body(u32* __restrict__ frame32, const int cam_w, const int cam_h, SlotState* st,
                                                 const uint2* __restrict__ dlut, float* __restrict__ depth, uint8_t* __restrict__ bgr,
                                                 const u32 tile_x, const u32 tile_y) {
  __shared__ uint16_t s_d[CAM32_T][CAM32_T + 2];
  __shared__ __attribute__((aligned(16))) uint8_t s_b[CAM32_T][CAM32_T * 3];
  const int tid = threadIdx.x, lo = tid & (CAM32_T - 1), hi = tid / CAM32_T;  // BLOCK / 32 = 8 columns (rows) per pass
  if (tile_x == 0 && tile_y == 0 && tid < CNT_SLOTS) {
    const u32 tag = st->tag_a;
    u32* c = st->cnt[(tag & 1) ^ 1][tid];
    c[0] = c[1] = c[2] = c[3] = 0;
    if (tid == 0) {
      st->tag_b = tag;
      if (u32* hf = st->host_flags) host_flag_store(hf + 1, tag);
    }
  }
  const int x0 = (int)tile_x * CAM32_T, y0 = (int)tile_y * CAM32_T;
#pragma unroll
  for (int i = 0; i < CAM32_T * CAM32_T / BLOCK; ++i) {  // lanes = consecutive rows of one frame column
    const int c = hi + i * (BLOCK / CAM32_T), x = x0 + c, y = y0 + lo;
    u32 k = 0;
    if (x < cam_w && y < cam_h) {
      u32* p = frame32 + (u32)x * (u32)cam_h + (u32)y;
      k = *p;
      if (k) *p = 0u;
    }
    s_d[c][lo] = (uint16_t)(k & 0xfffu);
  }
  __syncthreads();
  const bool dw_rows = bgr && (cam_w & 3) == 0 && x0 + CAM32_T <= cam_w && ((size_t)bgr & 3) == 0;  // whole 96-byte rows, dword aligned
#pragma unroll
  for (int i = 0; i < CAM32_T * CAM32_T / BLOCK; ++i) {  // lanes = consecutive pixels of one output row
    const int r = hi + i * (BLOCK / CAM32_T), x = x0 + lo, y = y0 + r;
    const uint2 e = dlut[s_d[lo][r]];
    if (x < cam_w && y < cam_h) {
      const u32 pixel = (u32)y * (u32)cam_w + (u32)x;
      if (depth) depth[pixel] = __uint_as_float(e.x);
      if (bgr && !dw_rows) {
        bgr[(size_t)pixel * 3 + 0] = (uint8_t)(e.y & 0xff);
        bgr[(size_t)pixel * 3 + 1] = (uint8_t)((e.y >> 8) & 0xff);
        bgr[(size_t)pixel * 3 + 2] = (uint8_t)((e.y >> 16) & 0xff);
      }
    }
    if (dw_rows) {
      s_b[r][lo * 3 + 0] = (uint8_t)(e.y & 0xff);
      s_b[r][lo * 3 + 1] = (uint8_t)((e.y >> 8) & 0xff);
      s_b[r][lo * 3 + 2] = (uint8_t)((e.y >> 16) & 0xff);
    }
  }
  if (dw_rows) {
    __syncthreads();
    constexpr int DW = CAM32_T * 3 / 4;  // dwords per staged row
    for (int i = tid; i < CAM32_T * DW; i += BLOCK) {
      const int r = i / DW, q = i - r * DW, y = y0 + r;
      if (y < cam_h) reinterpret_cast<u32*>(bgr + ((size_t)y * (size_t)cam_w + (size_t)x0) * 3)[q] = reinterpret_cast<const u32*>(&s_b[r][0])[q];
    }
  }
}

__global__ __launch_bounds__(BLOCK) void k_frame_cam32(u32* __restrict__ frame32, int cam_w, int cam_h, SlotState* st,
                                                       const uint2* __restrict__ dlut, float* __restrict__ depth,
                                                       uint8_t* __restrict__ bgr) {
  frame_cam32_body(frame32, cam_w, cam_h, st, dlut, depth, bgr, blockIdx.x, blockIdx.y);
}

__global__ __launch_bounds__(BLOCK) void k_frame_cam32_batch(const FrameDesc* __restrict__ descs, int cam_w, int cam_h,
                                                             const uint2* __restrict__ dlut) {
  const FrameDesc d = descs[blockIdx.z];
  if (!d.valid) return;
  frame_cam32_body(reinterpret_cast<u32*>(d.key_frame), cam_w, cam_h, d.st, dlut, d.depth, d.bgr, blockIdx.x, blockIdx.y);
}

// Sharded frames: a chunk of the (reduced) packed-key frame -> u16 disparities (0 where the tag differs): 2 instead of 8
// bytes per cell for the all-gather that follows the reduce-scatter
__global__ __launch_bounds__(BLOCK) void k_decode_keys_u16(const u64* __restrict__ f, u64 n_cells, u32 tag, uint16_t* __restrict__ out) {
  const u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x;
  if (i < n_cells) out[i] = key_disp(f[i], tag);
}

// camera view on a plain u16 disparity frame
__global__ __launch_bounds__(BLOCK) void k_frame_direct_u16(const uint16_t* __restrict__ disp, u64 n_pixels,
                                                            const uint2* __restrict__ dlut, float* __restrict__ depth,
                                                            uint8_t* __restrict__ bgr) {
  const u64 pixel = (u64)blockIdx.x * BLOCK + threadIdx.x;
  const uint2 e = dlut[pixel < n_pixels ? (u32)disp[pixel] : 0u];
  if (depth && pixel < n_pixels) depth[pixel] = __uint_as_float(e.x);
  if (bgr) store_bgr_block(bgr, (u64)blockIdx.x * BLOCK, n_pixels, e.y);
}

// =====================================================================================================
// stage / debug kernels (reference stage signatures; not on the fused path)
// =====================================================================================================
__global__ __launch_bounds__(BLOCK) void k_stage_rectify(const uint16_t* __restrict__ xs, const uint16_t* __restrict__ ys,
                                                         u64 n, DevTables tb, int16_t* __restrict__ xr,
                                                         int16_t* __restrict__ yr, u32* __restrict__ oob_count) {
  const u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const u32 x = xs[i], y = ys[i];
  if (x >= (u32)tb.cam_w || y >= (u32)tb.cam_h) {
    atomicAdd(oob_count, 1u);
    xr[i] = 0;
    yr[i] = 0;
    return;
  }
  const u32 l = tb.lut[x * (u32)tb.cam_h + y];
  xr[i] = (int16_t)(l & 0xffff);
  yr[i] = (int16_t)(l >> 16);
}

// CamProjMaps.rectify_cam_coords_f32 (cam_proj_calibration.py:272-275): gather from the caller's float rectify maps
// (row-major [cam_h][cam_w]); used by the offline evaluation caller (eval/compute_depth_x_maps.py:99) for the point cloud.
__global__ __launch_bounds__(BLOCK) void k_stage_rectify_f32(const uint16_t* __restrict__ xs, const uint16_t* __restrict__ ys,
                                                             u64 n, int cam_w, int cam_h, const float* __restrict__ mapx,
                                                             const float* __restrict__ mapy, float* __restrict__ xr,
                                                             float* __restrict__ yr, u32* __restrict__ oob_count) {
  const u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const u32 x = xs[i], y = ys[i];
  if (x >= (u32)cam_w || y >= (u32)cam_h) {
    atomicAdd(oob_count, 1u);
    xr[i] = 0.f;
    yr[i] = 0.f;
    return;
  }
  const u32 o = y * (u32)cam_w + x;
  xr[i] = mapx[o];
  yr[i] = mapy[o];
}

// CamProjMaps.construct_point_cloud (cam_proj_calibration.py:319-331): [x+d, y, -d, 1] through Q in float32,
// perspective divide, y and z negated.  d == 0 gives the same inf/nan the NumPy code produces.
struct Mat4f {
  float m[16];
};
__global__ __launch_bounds__(BLOCK) void k_point_cloud(const float* __restrict__ xpr, const float* __restrict__ ypr,
                                                       const float* __restrict__ disp, u64 n, Mat4f Q,
                                                       float* __restrict__ cloud) {
  const u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const float d = disp[i];
  const float p0 = xpr[i] + d, p1 = ypr[i], p2 = -d;
  float r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float acc = Q.m[4 * k] * p0;
    acc = fmaf(Q.m[4 * k + 1], p1, acc);
    acc = fmaf(Q.m[4 * k + 2], p2, acc);
    acc = acc + Q.m[4 * k + 3];
    r[k] = acc;
  }
  cloud[3 * i + 0] = r[0] / r[3];
  cloud[3 * i + 1] = -(r[1] / r[3]);
  cloud[3 * i + 2] = -(r[2] / r[3]);
}

// A2 on caller-supplied rectified coordinates
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_stage_event_disparity(const int16_t* __restrict__ xr,
                                                                 const int16_t* __restrict__ yr, const T* __restrict__ ts,
                                                                 u64 n, DevTables tb, const SlotState* st, u32 tag,
                                                                 int16_t* __restrict__ disp, uint8_t* __restrict__ mask) {
  u64 lo, hi;
  load_frame_minmax(st, tag & 1, lo, hi);
  const TimeNorm<T> tn(TimeCodec<T>::dec(lo), TimeCodec<T>::dec(hi), tb.t_px_scale);
  const u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const int x = xr[i], y = yr[i];
  int d = 0;
  bool ok = y >= 0 && y < tb.xmap_h - 1;
  if (ok) {
    const int col = tn.column(ts[i]);
    const int xp = (int)tb.xmap[col * tb.xmap_h + y];
    d = (int)(short)(xp - x - tb.x_offset);
    ok = d >= 0;
  }
  disp[i] = (int16_t)(ok ? d : 0);
  mask[i] = ok ? 1 : 0;
}

// A3 / A3' on caller-supplied per-event arrays (full length + mask)
template <int VIEW>
__global__ __launch_bounds__(BLOCK) void k_stage_scatter(const int16_t* __restrict__ xr, const int16_t* __restrict__ yr,
                                                         const uint16_t* __restrict__ xs, const uint16_t* __restrict__ ys,
                                                         const int16_t* __restrict__ disp, const uint8_t* __restrict__ mask,
                                                         u64 n, DevTables tb, u32 tag, u64* __restrict__ frame,
                                                         u32* __restrict__ oob_count) {
  const u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n || !mask[i]) return;
  u32 cell;
  const int d = disp[i];
  if constexpr (VIEW == 0) {
    int col = (int)(short)(xr[i] + d);
    const int row = yr[i];
    if (col < 0) col += tb.rect_w;
    int r = row;
    if (r < 0) r += tb.rect_h;  // stage API: arbitrary caller arrays, NumPy index rules
    if (col < 0 || col >= tb.rect_w || r < 0 || r >= tb.rect_h) {
      atomicAdd(oob_count, 1u);
      return;
    }
    cell = (u32)r * (u32)tb.rect_w + (u32)col;
  } else {
    const u32 x = xs[i], y = ys[i];
    if (x >= (u32)tb.cam_w || y >= (u32)tb.cam_h) {
      atomicAdd(oob_count, 1u);
      return;
    }
    cell = y * (u32)tb.cam_w + x;
  }
  // the stage frame stores the f32 value of the int16 disparity; negative values never pass the mask
  // in the reference's pipeline, but keep the low 16 bits faithfully and sign-extend on decode
  const u64 key = ((u64)tag << KEY_TAG_SHIFT) | (i << KEY_IDX_SHIFT) | (u64)(u32)(uint16_t)d;
  __hip_atomic_fetch_max(&frame[cell], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(BLOCK) void k_decode_keys_signed(const u64* __restrict__ f, u64 n_cells, u32 tag,
                                                              float* __restrict__ out) {
  const u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x;
  if (i < n_cells) {
    const u64 k = f[i];
    out[i] = (u32)(k >> KEY_TAG_SHIFT) == tag ? (float)(int)(short)(k & 0xffff) : 0.0f;
  }
}

// every intermediate of A1/A2 per event (tests)
template <typename T, bool HAS_P>
__global__ __launch_bounds__(BLOCK) void k_debug_events(const uint16_t* __restrict__ xs, const uint16_t* __restrict__ ys,
                                                        const T* __restrict__ ts, const int16_t* __restrict__ ps, u64 n,
                                                        DevTables tb, const SlotState* st, u32 tag,
                                                        int16_t* __restrict__ xr, int16_t* __restrict__ yr,
                                                        int16_t* __restrict__ tcol, int16_t* __restrict__ disp,
                                                        uint8_t* __restrict__ mask) {
  u64 lo, hi;
  load_frame_minmax(st, tag & 1, lo, hi);
  const TimeNorm<T> tn(TimeCodec<T>::dec(lo), TimeCodec<T>::dec(hi), tb.t_px_scale);
  const u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  bool oob;
  const bool used = !HAS_P || ps[i] == 1;
  const EventResult r = event_disparity<T>(tb, tn, xs[i], ys[i], ts[i], used, oob);
  if (xr) xr[i] = (int16_t)r.xr;
  if (yr) yr[i] = (int16_t)r.yr;
  if (tcol) tcol[i] = (int16_t)r.ts;
  if (disp) disp[i] = (int16_t)r.disp;
  if (mask) mask[i] = r.inlier ? 1 : 0;
}

// =====================================================================================================
// N1 (setup time): X-map construction, reference python/x_map.py:5-55 (Numba prange over rows).
//   x_map[y, c] = X_OFFSET + argmin_x |c / t_px_scale - time_map[y, x]|   over cells with time_map != 0,
//   FIRST minimum wins (strict <), kept only if the minimum is <= 2 / num_scanlines; c == 0 (t == 0) is skipped.
// All arithmetic in FP64 with the f32 map widened (what Numba does).  One block per rectified row: the row is
// staged in LDS as f64 once, every thread owns one time column and scans the row out of LDS (all lanes read the
// same address -> broadcast, conflict-free).  H*W_t*W compares = 1.5 G for the C-1M tables: ~1 ms here.
// =====================================================================================================
__global__ __launch_bounds__(BLOCK) void k_build_x_map(const float* __restrict__ time_map, int height, int width,
                                                       int x_map_width, int t_px_scale, int x_offset,
                                                       double max_t_diff, int16_t* __restrict__ x_map,
                                                       float* __restrict__ t_diffs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* row = reinterpret_cast<double*>(smem);
  const int y = blockIdx.x;
  for (int x = threadIdx.x; x < width; x += BLOCK) row[x] = (double)time_map[(size_t)y * width + x];
  __syncthreads();
  for (int c = threadIdx.x; c < x_map_width; c += BLOCK) {
    int16_t out = 0;
    float out_d = 0.0f;
    const double t = (double)c / (double)t_px_scale;
    if (t != 0.0) {
      double best = __builtin_inf();
      int best_x = -1;
      for (int x = 0; x < width; ++x) {
        const double m = row[x];
        const double d = fabs(t - m);
        const bool take = (m != 0.0) && (d < best);  // zero cells are undefined; strict < keeps the first minimum
        best = take ? d : best;
        best_x = take ? x : best_x;
      }
      if (best_x != -1 && best <= max_t_diff) {
        out = (int16_t)(best_x + x_offset);
        out_d = (float)best;
      }
    }
    x_map[(size_t)y * x_map_width + c] = out;
    if (t_diffs) t_diffs[(size_t)y * x_map_width + c] = out_d;
  }
}

// The same result from a sorted row (round 3): argmin_x |t - m[x]| with the FIRST minimum winning only depends on the row's defined
// values in sorted order, so the block sorts (value, x) pairs once (bitonic sort in LDS on 48-bit keys: the f32 value in an
// order-preserving encoding, then x -- equal values keep their smallest x in front) and every time column takes a few binary
// searches instead of a scan of the whole row: the nearest value at or above t and the nearest below it, each represented by
// its first x; the distances are the very fabs(t - m) of the scan, in FP64.  Ties: equal distances from both sides -> the
// smaller x, as the scan order would have it.  Two DIFFERENT values whose distances round to the same double (values many
// orders of magnitude below t) could hide a smaller x further out: the neighbouring distinct value on either side is checked
// and such a column is scanned like before (so is a row with a NaN).  H W log W + H W_t log W instead of H W_t W.
__device__ __forceinline__ u32 xmap_f32_key(float m) {
  const u32 b = __float_as_uint(m);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ double xmap_key_val(u64 key) {
  const u32 k = (u32)(key >> 16);
  const u32 b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return (double)__uint_as_float(b);
}

__global__ __launch_bounds__(BLOCK) void k_build_x_map_sorted(const float* __restrict__ time_map, int height, int width, int wp,
                                                              int x_map_width, int t_px_scale, int x_offset, double max_t_diff,
                                                              int16_t* __restrict__ x_map, float* __restrict__ t_diffs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64* keys = reinterpret_cast<u64*>(smem);              // [wp] (wp = power of two >= width)
  float* row = reinterpret_cast<float*>(keys + wp);      // [width], in scan order (the fallback)
  __shared__ int s_nan, s_ndef;
  const int y = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) s_nan = s_ndef = 0;
  __syncthreads();
  int my_def = 0;
  bool my_nan = false;
  for (int x = tid; x < wp; x += BLOCK) {
    u64 k = ~0ull;
    if (x < width) {
      const float m = time_map[(size_t)y * width + x];
      row[x] = m;
      my_nan = my_nan || m != m;
      if (m != 0.0f && m == m) {
        k = ((u64)xmap_f32_key(m) << 16) | (u64)(u32)x;
        my_def += 1;
      }
    }
    keys[x] = k;
  }
  if (my_def) atomicAdd(&s_ndef, my_def);
  if (my_nan) s_nan = 1;
  __syncthreads();
  for (int size = 2; size <= wp; size <<= 1)  // bitonic sort, ascending
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < (wp >> 1); i += BLOCK) {
        const int lo = ((i / stride) * stride << 1) + (i % stride), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const u64 a = keys[lo], b = keys[hi];
        if ((a > b) == up) {
          keys[lo] = b;
          keys[hi] = a;
        }
      }
      __syncthreads();
    }
  const int n_def = s_ndef;
  const bool scan_all = s_nan != 0;
  const auto lower = [&](int lo, int hi, double v) {  // first index in [lo, hi) whose value is >= v
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (xmap_key_val(keys[mid]) >= v) hi = mid; else lo = mid + 1;
    }
    return lo;
  };
  const auto upper = [&](int lo, int hi, double v) {  // first index in [lo, hi) whose value is > v
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (xmap_key_val(keys[mid]) > v) hi = mid; else lo = mid + 1;
    }
    return lo;
  };
  for (int c = tid; c < x_map_width; c += BLOCK) {
    int16_t out = 0;
    float out_d = 0.0f;
    const double t = (double)c / (double)t_px_scale;
    if (t != 0.0 && (n_def > 0 || scan_all)) {
      double best = __builtin_inf();
      int best_x = -1;
      bool scan = scan_all;
      if (!scan) {
        const int i = lower(0, n_def, t);
        if (i < n_def) {
          const double m_hi = xmap_key_val(keys[i]);
          best = fabs(t - m_hi);
          best_x = (int)(keys[i] & 0xffffull);
          const int e = upper(i, n_def, m_hi);
          if (e < n_def && fabs(t - xmap_key_val(keys[e])) == best) scan = true;
        }
        if (i > 0) {
          const double m_lo = xmap_key_val(keys[i - 1]);
          const int j = lower(0, i, m_lo);
          const double d_lo = fabs(t - m_lo);
          const int x_lo = (int)(keys[j] & 0xffffull);
          if (j > 0 && fabs(t - xmap_key_val(keys[j - 1])) == d_lo) scan = true;
          if (d_lo < best || (d_lo == best && x_lo < best_x)) {
            best = d_lo;
            best_x = x_lo;
          }
        }
      }
      if (scan) {  // the scan of k_build_x_map
        best = __builtin_inf();
        best_x = -1;
        for (int x = 0; x < width; ++x) {
          const double m = (double)row[x];
          const double d = fabs(t - m);
          const bool take = (m != 0.0) && (d < best);
          best = take ? d : best;
          best_x = take ? x : best_x;
        }
      }
      if (best_x != -1 && best <= max_t_diff) {
        out = (int16_t)(best_x + x_offset);
        out_d = (float)best;
      }
    }
    x_map[(size_t)y * x_map_width + c] = out;
    if (t_diffs) t_diffs[(size_t)y * x_map_width + c] = out_d;
  }
}

// =====================================================================================================
// N3: per-frame de-duplication filters, reference python/frame_event_filter.py:19-128.
//   LastEventPerXY / FirstEventPerXY / MeanFirstLastEventPerXY: one output event per camera pixel that fired, carrying
//   the last / first / mean(first,last) timestamp; FirstEventPerYT: one per (row, x_proj) cell carrying the first event's
//   x and t.  The reference writes `map[y, x] = t` for all events (forward = last writer wins, reversed = first wins) into
//   int32 maps and reads them back in raster order.  Here: (1) per event, atomic max / min of the event index per cell,
//   (2) exclusive scan of the occupancy mask (two-level block scan), (3) emit EventCD records in raster order.
//   Timestamps go through the reference's int32 maps: t_out = (int64)(int32)t, mean = ((int32)a + (int32)b) >> 1.
//   OBSERVED REFERENCE BEHAVIOUR: the "first" maps are filled with `map[y[::-1], x[::-1]] = t[::-1]`; NumPy (1.26 and
//   2.2 checked) normalises the negative strides of all operands together and iterates in memory order, so that
//   statement keeps the LAST event exactly like the forward one.  As the reference actually runs, FirstEventPerXY ==
//   LastEventPerXY == MeanFirstLast and FirstEventPerYT keeps the last event per (y, x_proj).  `intended == 0`
//   reproduces that (it is what the golden vectors captured from the reference contain); `intended != 0` gives
//   the semantics the class names promise.
// =====================================================================================================
enum { FILTER_FIRST_PER_YT = 1, FILTER_FIRST_PER_XY = 2, FILTER_LAST_PER_XY = 3, FILTER_MEAN_PER_XY = 4 };
constexpr int SCAN_BLOCK = 1024;

__global__ __launch_bounds__(BLOCK) void k_filter_scatter(const uint4* __restrict__ aos, const int16_t* __restrict__ xp, u64 n,
                                                          int by_xp, int map_h, int map_w, u32* __restrict__ first_idx,
                                                          u32* __restrict__ last_idx, u32* __restrict__ oob_count) {
  const u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint4 r = aos[i];
  if ((short)(r.y & 0xffff) != 1) return;  // events[events["p"] == 1]
  int col = by_xp ? (int)xp[i] : (int)(r.x & 0xffff);
  const int row = (int)(r.x >> 16);
  if (col < 0) col += map_w;  // NumPy negative index
  if (col < 0 || col >= map_w || row >= map_h) {
    atomicAdd(oob_count, 1u);
    return;
  }
  const u32 cell = (u32)row * (u32)map_w + (u32)col;
  __hip_atomic_fetch_max(&last_idx[cell], (u32)i + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_min(&first_idx[cell], (u32)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// exclusive scan of (last_idx != 0) inside each SCAN_BLOCK-cell block; block totals to sums[]
__global__ __launch_bounds__(SCAN_BLOCK) void k_filter_scan_blocks(const u32* __restrict__ last_idx, u32 n_cells,
                                                                   u32* __restrict__ pos, u32* __restrict__ sums) {
  __shared__ u32 s_wave[SCAN_BLOCK / 64];
  const u32 i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  const bool occ = i < n_cells && last_idx[i] != 0;
  const u64 ballot = __ballot(occ);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32 before = __popcll(ballot & ((1ull << lane) - 1ull));
  if (lane == 0) s_wave[wave] = __popcll(ballot);
  __syncthreads();
  u32 base = 0;
  for (int w = 0; w < wave; ++w) base += s_wave[w];
  if (i < n_cells) pos[i] = base + before;
  if (threadIdx.x == SCAN_BLOCK - 1) sums[blockIdx.x] = base + before + (occ ? 1u : 0u);
}

// exclusive scan of the block totals (single block; n_blocks is a few hundred)
__global__ __launch_bounds__(SCAN_BLOCK) void k_filter_scan_sums(u32* __restrict__ sums, u32 n_blocks, u32* __restrict__ total) {
  __shared__ u32 s[SCAN_BLOCK];
  u32 carry = 0;
  for (u32 b0 = 0; b0 < n_blocks; b0 += SCAN_BLOCK) {
    const u32 i = b0 + threadIdx.x;
    const u32 v = i < n_blocks ? sums[i] : 0u;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < SCAN_BLOCK; o <<= 1) {  // Hillis-Steele inclusive scan
      const u32 add = threadIdx.x >= (u32)o ? s[threadIdx.x - o] : 0u;
      __syncthreads();
      s[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < n_blocks) sums[i] = carry + s[threadIdx.x] - v;
    const u32 blk_total = s[SCAN_BLOCK - 1];
    __syncthreads();
    carry += blk_total;
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_filter_emit(const uint4* __restrict__ aos, const u32* __restrict__ first_idx,
                                                            const u32* __restrict__ last_idx, const u32* __restrict__ pos,
                                                            const u32* __restrict__ sums, u32 n_cells, int map_w, int filter,
                                                            int intended, uint4* __restrict__ out) {
  const u32 i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  if (i >= n_cells) return;
  const u32 li = last_idx[i];
  if (!li) return;
  const u32 o = sums[blockIdx.x] + pos[i];
  const uint4 last = aos[li - 1];
  const uint4 first = intended ? aos[first_idx[i]] : last;
  const int t_first = (int)first.z, t_last = (int)last.z;  // low 32 bits = the reference's int32 maps
  int t32, x = (int)(i % (u32)map_w);
  const int y = (int)(i / (u32)map_w);
  if (filter == FILTER_LAST_PER_XY) t32 = t_last;
  else if (filter == FILTER_MEAN_PER_XY) t32 = (int)((u32)t_last + (u32)t_first) >> 1;  // int32 wrap, floor division by 2
  else t32 = t_first;
  if (filter == FILTER_FIRST_PER_YT) x = (int)(first.x & 0xffff);
  const long long t64 = (long long)t32;
  uint4 rec;
  rec.x = ((u32)(uint16_t)y << 16) | (u32)(uint16_t)x;
  rec.y = 1u;  // p = True
  rec.z = (u32)((u64)t64 & 0xffffffffull);
  rec.w = (u32)((u64)t64 >> 32);
  out[o] = rec;
}

// =====================================================================================================
// N2: pause detection for the frame segmentation, reference python/trigger_finder.py:153-155:
//   frame_paused_ev_idx = np.nonzero(np.diff(evs["t"]) >= frame_paused_thresh_us)[0]
// on a device-resident stream: flag -> two-level exclusive scan (same scan kernels as the filters) -> emit indices.
// Works on SoA t[n] or on EventCD records (t at byte 8 of each 16-byte record).
// =====================================================================================================
__global__ __launch_bounds__(SCAN_BLOCK) void k_pause_flags(const long long* __restrict__ t, const uint4* __restrict__ aos,
                                                            u32 n, long long thresh, u32* __restrict__ flags) {
  const u32 i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  if (i >= n) return;
  u32 f = 0;
  if (i + 1 < n) {
    long long a, b;
    if (aos) {
      const uint4 ra = aos[i], rb = aos[i + 1];
      a = (long long)(((u64)ra.w << 32) | ra.z);
      b = (long long)(((u64)rb.w << 32) | rb.z);
    } else {
      a = t[i];
      b = t[i + 1];
    }
    f = (b - a) >= thresh ? 1u : 0u;
  }
  flags[i] = f;  // k_filter_scan_blocks treats non-zero as "occupied"
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_pause_emit(const u32* __restrict__ flags, const u32* __restrict__ pos,
                                                           const u32* __restrict__ sums, u32 n, u32* __restrict__ out) {
  const u32 i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  if (i < n && flags[i]) out[sums[blockIdx.x] + pos[i]] = i;
}

// =====================================================================================================
// N4: evaluation metrics of the reference's table script (python/eval/create_evaluation_table.py:14-63):
//   load_and_filter : est >= max_depth -> 0, est <= min_depth -> 0, est[gt == 0] = 0
//   evaluation_stats: margin = 0.01 * mean(gt[gt > 0]); fill rate = (#(|gt - est| < margin, with the difference zeroed where
//   gt == 0) - #(gt == 0)) / (H W - #(gt == 0)); RMSE over (gt > 0) & (est > 0); % of pixels whose error exceeds 1 / 5 / 10
//   (error zeroed where gt == 0).  Differences are taken in f32 like NumPy does on f32 maps; sums are accumulated in f64
//   (the reference's f32 pairwise sums agree to ~1e-7 relative).  Two passes: pass 1 the margin's sum / count, pass 2 the rest.
// =====================================================================================================
struct EvalAcc {
  double sum_gt, sum_sq;
  u64 n_gt_pos, n_gt_zero, n_close, n_valid, n1, n5, n10;
};

__device__ inline float eval_filtered(float est, float gt, int filter, float min_d, float max_d) {
  if (filter) {
    if (est >= max_d) est = 0.0f;
    if (est <= min_d) est = 0.0f;
    if (gt == 0.0f) est = 0.0f;
  }
  return est;
}

template <int PASS>
__global__ __launch_bounds__(BLOCK) void k_eval_stats(const float* __restrict__ est, const float* __restrict__ gt, u64 n,
                                                      int filter, float min_d, float max_d, EvalAcc* acc) {
  double s = 0.0;
  u64 c[7] = {0, 0, 0, 0, 0, 0, 0};
  double margin = 0.0;
  if (PASS == 2) margin = 0.01 * acc->sum_gt / (double)acc->n_gt_pos;  // NaN when no gt > 0, like NumPy's 0 / 0
  const u64 stride = (u64)gridDim.x * BLOCK;
  for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    const float g = gt[i];
    if (PASS == 1) {
      if (g > 0.0f) {
        s += (double)g;
        c[0] += 1;
      }
    } else {
      const float e = eval_filtered(est[i], g, filter, min_d, max_d);
      const float d = g - e;
      const float a = g == 0.0f ? 0.0f : fabsf(d);
      c[1] += g == 0.0f;
      c[2] += (double)a < margin;
      if (g > 0.0f && e > 0.0f) {
        c[3] += 1;
        s += (double)(d * d);  // pow(gt - est, 2) on f32 arrays is an f32 product
      }
      c[4] += a > 1.0f;
      c[5] += a > 5.0f;
      c[6] += a > 10.0f;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
#pragma unroll
    for (int k = 0; k < 7; ++k) c[k] += __shfl_xor(c[k], o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    if (PASS == 1) {
      atomicAdd(&acc->sum_gt, s);
      atomicAdd((unsigned long long*)&acc->n_gt_pos, (unsigned long long)c[0]);
    } else {
      atomicAdd(&acc->sum_sq, s);
      atomicAdd((unsigned long long*)&acc->n_gt_zero, (unsigned long long)c[1]);
      atomicAdd((unsigned long long*)&acc->n_close, (unsigned long long)c[2]);
      atomicAdd((unsigned long long*)&acc->n_valid, (unsigned long long)c[3]);
      atomicAdd((unsigned long long*)&acc->n1, (unsigned long long)c[4]);
      atomicAdd((unsigned long long*)&acc->n5, (unsigned long long)c[5]);
      atomicAdd((unsigned long long*)&acc->n10, (unsigned long long)c[6]);
    }
  }
}

// slot (re)initialisation: zero the key frame, arm min/max + counters, tag = 0
__global__ __launch_bounds__(BLOCK) void k_reset_slot(SlotState* st, u64* __restrict__ frame, u64 n_cells,
                                                      unsigned char* __restrict__ dirty) {
  const u64 stride = (u64)gridDim.x * BLOCK;
  for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n_cells; i += stride) frame[i] = 0;
  if (dirty)
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < ((n_cells + 15) >> 4); i += stride) dirty[i] = 0;
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) {
      st->tag_a = 0;
      st->tag_b = 0;
      st->pad[1] = 0;  // (frame_attempt_failed: tags start over, a stale tag of the old numbering must not match a new one)
      // unsorted_sticky is NOT cleared here: this kernel also runs on tag wrap, and a violation recorded since the last
      // xm_sync must still be reported; it starts at 0 (xm_create zeroes the states) and xm_sync clears it
    }
    for (int i = threadIdx.x; i < 2 * MM_SLOTS; i += BLOCK) {
      st->mm[i / MM_SLOTS][i % MM_SLOTS][0] = MM_INIT_MIN;
      st->mm[i / MM_SLOTS][i % MM_SLOTS][1] = MM_INIT_MAX;
    }
    for (int i = threadIdx.x; i < 2 * CNT_SLOTS * CNT_STRIDE; i += BLOCK) (&st->cnt[0][0][0])[i] = 0;
  }
}

// A frame of a captured batch whose column-tile attempt failed: forget what the attempt counted (same tag, same parity) before
// K0 / K1 / K2 of the 64-bit path run on it.  grid = frames.
__global__ __launch_bounds__(64) void k_redo_prepare_batch(const FrameDesc* __restrict__ descs) {
  const FrameDesc d = descs[blockIdx.x];
  if (!d.valid || !frame_attempt_failed(d.st)) return;
  const u32 parity = d.st->tag_a & 1;
  for (int i = threadIdx.x; i < CNT_SLOTS * CNT_STRIDE; i += 64) (&d.st->cnt[parity][0][0])[i] = 0;
}

}  // namespace xm
