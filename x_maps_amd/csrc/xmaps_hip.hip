// xmaps_hip.hip -- host side of libxmaps_hip.so: the C-ABI declared in include/xmaps.h.
// Written for MI355X (gfx950) only: build with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared xmaps_hip.hip -o libxmaps_hip.so
// -ffp-contract=off keeps the time normalisation (divide, multiply, rint) unfused = bit-exact with NumPy.
#include "xmaps_kernels.hpp"
#include "xmaps_k1cols.hpp"
#include "xmaps_k1own.hpp"
#include "xmaps_k2pipe.hpp"
#include "xmaps_ingest.hpp"

#include <hip/hip_ext.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <limits>
#include <memory>
#include <algorithm>
#include <new>
#include <type_traits>
#include <string>
#include <vector>

#include "../../include/xmaps.h"

using namespace xm;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) return fail(XM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                      __FILE__, __LINE__);                                          \
  } while (0)

struct DevBuf {  // grow-only device scratch
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return XM_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    HIP_TRY(hipMalloc(&p, want));
    cap = want;
    return XM_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct EventsView {
  const uint16_t* x = nullptr;
  const uint16_t* y = nullptr;
  const void* t = nullptr;
  const int16_t* p = nullptr;
  const void* aos = nullptr;
  size_t n = 0;
  int t_dtype = XM_T_INT64;
  bool use_p = false;
};

struct Slot {
  hipStream_t stream = nullptr;
  bool owns_stream = true;
  int worker = -1;    // launch worker of this slot's stream (-1: none)
  u32 api_tag = 0;    // tag of the slot's last frame as the API thread counts them (== host_tag once the workers are idle)
  // XM_FLAG_TRY_SORTED: pinned host words the kernels report to ([0] tag of the last frame whose shortcut failed, [1] tag of
  // the last frame whose K2 has started) and what is needed to redo the slot's last asynchronous frame on the general path
  u32* h_flags = nullptr;
  struct Prev {
    bool valid = false;
    EventsView ev;
    float* depth = nullptr;
    uint8_t* bgr = nullptr;
    bool check = false;           // the frame took the try-sorted shortcut: its verdict decides about a redo
    float* host_depth = nullptr;  // XM_MEM_HOST_PINNED: where the outputs are copied to
    uint8_t* host_bgr = nullptr;
    u32 tag = 0;
    hipStream_t stream = nullptr;  // the stream the frame's launches went to (the group's stream for xm_process_batch)
  } prev;
  u64* key_frame = nullptr;
  u32* key32 = nullptr;            // compact key frame of the verified-sorted projector-view path (see key32_tag)
  u32 key32_valid_from = 0;        // tag of the frame before which key32 was last cleared: every key in it has a tag in
                                   // [valid_from, valid_from + 15), so the 4-bit tag field is unambiguous
  bool last_key32 = false;         // the slot's last frame took a compact path (key32 or column tiles): a failure counts against it
  bool last_cols = false;          // ... the column tiles (K0b was launched in K0's place)
  uint16_t* frame16 = nullptr;     // plain u16 disparity frame of the column-tile path (xmaps_k1cols.hpp): rewritten by every frame
  unsigned char* dirty = nullptr;  // projector view: one flag byte per 128-byte line of key_frame
  SlotState* st = nullptr;  // device
  u32 host_tag = 0;         // mirrors st->tag_a after the enqueued work has run
  bool any_frame = false;
  bool last_sorted = false;
  uint64_t last_n = 0;
  int last_t_dtype = XM_T_INT64;  // how xm_last_frame_stats decodes t_min / t_max
  // the slot's last frame ran inside a multi-frame launch on ANOTHER stream: work on the slot's own stream waits for this
  hipEvent_t pending_batch_ev = nullptr;
  hipStream_t pending_batch_stream = nullptr;
  bool eager_dirty = false;  // eager work was enqueued on the slot's own stream since the last synchronisation point
  // staging for XM_MEM_HOST calls
  DevBuf ev_x, ev_y, ev_t, ev_p, ev_aos, out_depth, out_bgr, dbg[5];
};

// ---- launch workers -------------------------------------------------------------------------------------------
// A kernel launch costs the calling thread ~2.7 us in the HIP runtime, three launches per frame; with the GPU at ~12 us per
// frame that single thread is what bounds the asynchronous device-pointer path (tools/only_kernel_eager.sh: 3.2 us per call
// + 2.75 us per launch, whatever the kernels do).  One worker thread per slot stream takes the launches: the API call only
// settles the slot's previous frame, assigns the frame to a slot and posts a job (5 instead of 11.5 us per call).  The frame
// rate does not change -- with the drain fix and own hardware queues the GPU is the bound -- so this is opt-in
// (XM_FLAG_LAUNCH_WORKERS) for hosts whose calling thread has other work to do.
struct Job {
  enum Kind : int { FRAME = 0, STOP = 1 };
  int kind = FRAME;
  int slot = 0;
  EventsView ev;
  float* depth = nullptr;
  uint8_t* bgr = nullptr;
  bool allow_sorted = true;
};

struct Worker {
  static constexpr unsigned CAP = 256;  // jobs in flight per stream (the producer waits when full)
  Job ring[CAP];
  std::atomic<unsigned long long> head{0}, tail{0}, done{0};  // produced / taken / finished
  std::atomic<int> error{0};  // first failing return code of a job (reported by the next xm_sync)
  std::string error_text;
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<bool> sleeping{false};
  std::thread th;
};

}  // namespace

struct xm_handle {
  xm_config cfg{};
  DevTables tb{};
  u32* d_lut = nullptr;
  int16_t* d_xmap = nullptr;
  u32* d_pmap = nullptr;
  uint2* d_dlut = nullptr;
  // K2's static per-tile / per-pixel tables for its two geometries: [0] one pixel per thread (16 x 16 tiles), [1] two (32 x 16)
  // ([2]: four pixels per thread, 64 x 16 tiles -- the pipelined kernel on rigs whose patches are small against the tile)
  int4* d_k2_tiles[3] = {nullptr, nullptr, nullptr};
  u32* d_k2_pix[3] = {nullptr, nullptr, nullptr};
  int k2_tile_cap[3] = {K2_TILE_MAX, K2_TILE_MAX, K2_TILE_MAX};  // cells of the largest K2 patch (multiple of 8)
  bool k2_pipe4 = false;  // the pipelined kernel takes the 64 x 16 geometry
  int k2_patch_cols_max = 0;  // widest patch of the 16 x 16 / 32 x 16 tiles (-1: some patch does not fit LDS)
  int k2_force_ppt = 0;                             // XM_K2_PPT=1/2: experiments
  // pipelined K2 of the group launches (xmaps_k2pipe.hpp): table entries kept in LDS, CUs of the device, switch (XM_K2_PIPE=0: off)
  int k2_pipe_nlds = 0, n_cus = 256;
  bool k2_pipe = true, k2_pipe_rig_ok = false;  // (rig_ok: every tile's patch fits the pipelined loader, rect_h % 8 == 0)
  ulonglong2* d_zero16 = nullptr;  // 16 zero bytes: what K2 reads instead of a clean key-frame line
  SlotState* d_states = nullptr;  // n_slots + 1 (last = aux state for stage / shard calls)
  SlotState* aux_st = nullptr;
  std::vector<Slot> slots;
  int next_slot = 0;
  int last_slot = 0;
  size_t key_cells = 0;   // cells of the fused path's key frame (rect or camera frame)
  int out_w = 0, out_h = 0;
  u64* stage_frame = nullptr;  // lazily allocated scratch for the stage API (max(rect, cam) cells)
  size_t stage_cells = 0;
  hipEvent_t prof_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t fork_ev = nullptr;
  // K1 tiling: LDS windows (time columns / camera columns) and the dynamic LDS they need; 0 = direct kernel
  int w_ts = 0, w_x = 0;
  size_t k1_lds = 0;
  bool k1_direct = false, k2_direct = false;
  bool k2_flags = false;      // XM_K2_FLAGS=1: K1 marks dirty 128-byte lines of the key frame, K2 skips clean ones.
                              // Measured: K2 fetches 37 % fewer bytes but is not faster (it is latency, not bandwidth bound)
  std::vector<hipStream_t> gstreams;  // default-priority streams the hipGraph batches are captured on and launched from
  std::vector<std::unique_ptr<Worker>> workers;  // one per slot stream (empty: launches happen in the calling thread)
  bool key32_ok = false;      // the rig qualifies for the compact key frame (projector view, rect_h % 4 == 0, disparities < 4096)
  // column-tile K1 (xmaps_k1cols.hpp): the rig qualifies (projector view, cell(row, column) injective, no int16 wrap in the
  // disparity arithmetic), smallest rectified x of the LUT, widest tile the LDS budget allows, events a tile should hold
  bool cols_ok = false;
  bool cols_single = false;  // XM_COLS=2: also for single-frame calls (default: groups of frames only -- a single frame's third
                             // launch, the boundary pass, costs the pipelined one-frame-per-call path more than the tiles save)
  int cols_xr_min = 0, cols_w_max = 0, cols_target = 3700;
  int cols_flags = 0;  // COLS_F_ALL_IN_FRAME when no live (row, column) pair of the rig maps outside the frame
  // owner tiles (xmaps_k1own.hpp): the rig's (row, column) -> cell map is not injective (the reference's own calibration), but
  // every cell's columns lie within own_halo columns of its first one: cols_ok with own_mode set; fixed tile width own_w
  bool own_mode = false;
  int own_w = 0, own_halo = 0;
  uint16_t* d_xmap_own = nullptr;
  uint16_t* d_xmap_extra = nullptr;
  int4* d_own_tiles = nullptr;
  int16_t* d_own_base = nullptr;
  uint16_t* d_own_masks = nullptr;
  u32* d_own_extra_cells = nullptr;
  int own_extras = 0;  // owner cells outside their tile's band, over all tiles
  // XM_FLAG_ADAPTIVE_BATCH: asynchronous device-pointer frames are submitted as GROUPS (multi-frame launches) whenever the GPU
  // is still busy with earlier ones: a frame is launched at once while fewer than three groups are in flight (an idle GPU -- the
  // 60 Hz live case -- never waits), otherwise it joins the pending list, which goes out as one group when a group in flight
  // has finished, when it holds ab_max = n_slots / 4 frames, or at the next synchronising call
  struct Deferred {
    EventsView ev;
    float* depth;
    uint8_t* bgr;
  };
  std::vector<Deferred> pending;
  int ab_max = 0;                                  // 0: off
  hipEvent_t ab_inflight[4] = {nullptr, nullptr, nullptr, nullptr};  // end-of-group events of the last four groups submitted this way
  uint64_t ab_groups = 0, ab_frames = 0;
  std::atomic<uint64_t> path_counts[4] = {};  // frames enqueued per K1 variant (xm_path_counts)
  // (atomics: with XM_FLAG_LAUNCH_WORKERS the launch threads and the API thread all pass through enqueue_frame)
  std::atomic<int> key32_score{0};  // raised by frames that failed the compact path, decays with every frame that took it
  std::atomic<int> key32_pause{0};  // frames for which the compact path stays switched off (it kept failing: sparse / noisy stream)
  bool time_sorted = false;   // XM_FLAG_TIME_SORTED
  bool try_sorted = false;    // XM_FLAG_TRY_SORTED
  bool gate_slots = false;    // experiments (XM_GATE_SLOTS=1): asynchronous calls wait (polling a pinned word) until the slot's previous frame has reached K2
  bool capturing = false;     // inside xm_graph_create's stream capture (no host-side redo possible there)
  uint64_t sorted_fallbacks = 0;
  std::vector<hipEvent_t> join_ev;
  // multi-frame launches (xm_process_batch, batched hipGraphs, ingest): frame descriptors.  Eager batches stage them
  // through a ring of pinned host entries -> device entries (one memcpy per batch, stream-ordered before its kernels).
  static constexpr int DESC_RING = 16;
  FrameDesc* h_descs = nullptr;   // pinned  [DESC_RING][n_slots]
  FrameDesc* d_descs = nullptr;   // device  [DESC_RING][n_slots]
  hipEvent_t desc_ev[DESC_RING] = {};  // recorded after the ring entry's upload: the entry may be rewritten once it fired
  bool desc_used[DESC_RING] = {};
  int desc_next = 0;
  uint64_t batch_counter = 0;
  // an event per (stream, ring entry) recorded at the end of a batch: eager work on a slot's own stream waits for it
  std::vector<std::vector<hipEvent_t>> batch_ev;  // [distinct stream][8]
  std::vector<hipStream_t> streams;               // distinct slot streams
  std::vector<int> batch_ev_next;
  hipEvent_t graph_ev[8] = {};  // end-of-replay events (ring), recorded on the graphs' origin stream
  unsigned graph_ev_next = 0;
  // dynamic-LDS caps already raised on this handle's device, per kernel function (launch workers call concurrently)
  std::mutex lds_mu;
  std::vector<std::pair<const void*, size_t>> lds_caps;
  int ensure_lds(const void* fn, size_t bytes);
};

struct xm_graph {
  xm_handle* h = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  std::vector<u32> frames_on_slot;
  int n_frames = 0;
  std::vector<FrameDesc> h_descs;  // batched capture: the frames' descriptors (static for the graph's lifetime)
  FrameDesc* d_descs = nullptr;
};

int flush_pending(xm_handle* h);  // XM_FLAG_ADAPTIVE_BATCH: submit the frames held back (defined beside xm_process_batch)

int xm_handle::ensure_lds(const void* fn, size_t bytes) {
  std::lock_guard<std::mutex> lk(lds_mu);
  for (auto& e : lds_caps)
    if (e.first == fn) {
      if (bytes <= e.second) return XM_OK;
      HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
      e.second = bytes;
      return XM_OK;
    }
  HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  lds_caps.emplace_back(fn, bytes);
  return XM_OK;
}

namespace {

// Profile mode (xm_profile_frame): the three hot-path launches go through hipExtLaunchKernelGGL, which ties a start
// and a stop event to the dispatch packet itself -- the same timestamps rocprofv3 --kernel-trace reports -- instead
// of bracketing the launch with hipEventRecord (which adds ~3-5 us of event processing to every interval).
struct ProfCtx {
  hipEvent_t start = nullptr, stop = nullptr;
};
thread_local ProfCtx g_prof;

#define XM_LAUNCH(kernel, grid, block, lds, stream, ...)                                                   \
  do {                                                                                                      \
    if (g_prof.start)                                                                                       \
      hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)(lds), stream, g_prof.start, g_prof.stop, 0u, __VA_ARGS__); \
    else                                                                                                    \
      hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                    \
  } while (0)

inline unsigned grid_for(u64 items, unsigned per_block) {
  u64 g = (items + per_block - 1) / per_block;
  return (unsigned)(g ? g : 1);
}

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
inline long long rec_t_host(const uint4& r) { return (long long)(((u64)r.w << 32) | r.z); }

size_t t_size(int t_dtype) { return t_dtype == XM_T_FLOAT32 ? 4 : 8; }

int reset_slot(xm_handle* h, Slot& s, hipStream_t stream = nullptr) {
  if (!stream) stream = s.stream;
  hipLaunchKernelGGL(k_reset_slot, dim3(1024), dim3(BLOCK), 0, stream, s.st, s.key_frame, (u64)h->key_cells, s.dirty);
  HIP_TRY(hipGetLastError());
  if (s.key32) HIP_TRY(hipMemsetAsync(s.key32, 0, h->key_cells * sizeof(u32), stream));
  s.key32_valid_from = 0;
  s.host_tag = 0;
  s.api_tag = 0;
  if (s.h_flags) {  // tags start over: forget the verdicts of the old numbering (no frame of this slot is pending here)
    HIP_TRY(hipStreamSynchronize(stream));
    s.h_flags[0] = s.h_flags[1] = 0;
  }
  return XM_OK;
}

// ---- kernel dispatch ---------------------------------------------------------------------------------
template <typename T, bool AOS, bool HAS_P>
void launch_minmax_t(const EventsView& ev, SlotState* st, u32 tag_override, hipStream_t stream) {
  const u64 n = ev.n;
  const bool vec2 = !AOS && sizeof(T) == 8 && std::is_same<T, long long>::value && aligned(ev.t, 16) &&
                    (!HAS_P || aligned(ev.p, 4));
  // ~2048 events per thread-block iteration keeps every CU busy without drowning the 32 atomic slots
  const unsigned per_block = BLOCK * (vec2 ? 2 * K0_UN : 4);  // VEC2: K0_UN loads x 2 events per thread per sweep
  unsigned grid = grid_for(n, per_block);
  if (grid > 1024) grid = 1024;
  if constexpr (std::is_same<T, long long>::value && !AOS) {
    if (vec2) {
      XM_LAUNCH((k_minmax<T, false, HAS_P, 2>), dim3(grid), dim3(BLOCK), 0, stream, (const T*)ev.t, ev.p,
                (const uint4*)nullptr, n, st, tag_override);
      return;
    }
  }
  XM_LAUNCH((k_minmax<T, AOS, HAS_P, 1>), dim3(grid), dim3(BLOCK), 0, stream, (const T*)ev.t, ev.p,
            (const uint4*)ev.aos, n, st, tag_override);
}

void launch_minmax(const EventsView& ev, SlotState* st, u32 tag_override, hipStream_t stream) {
  if (ev.aos) {
    if (ev.use_p) launch_minmax_t<long long, true, true>(ev, st, tag_override, stream);
    else launch_minmax_t<long long, true, false>(ev, st, tag_override, stream);
    return;
  }
  const bool hp = ev.use_p;
  switch (ev.t_dtype) {
    case XM_T_INT64:
      hp ? launch_minmax_t<long long, false, true>(ev, st, tag_override, stream)
         : launch_minmax_t<long long, false, false>(ev, st, tag_override, stream);
      break;
    case XM_T_FLOAT32:
      hp ? launch_minmax_t<float, false, true>(ev, st, tag_override, stream)
         : launch_minmax_t<float, false, false>(ev, st, tag_override, stream);
      break;
    default:
      hp ? launch_minmax_t<double, false, true>(ev, st, tag_override, stream)
         : launch_minmax_t<double, false, false>(ev, st, tag_override, stream);
  }
}

struct ScatterArgs {
  xm_handle* h;
  const EventsView* ev;
  const DevTables* tb;
  int view;
  SlotState* st;
  u32 tag_override;
  u64 idx_offset, mm_lo, mm_hi;
  const void* mm_ext;  // sharded mode: {tmin, -tmax} in device memory (NULL: mm_lo / mm_hi)
  u64* frame;
  unsigned char* dirty;
  hipStream_t stream;
  int w_ts, w_x;
  size_t lds;
  bool direct;
  bool sorted;
  bool key32;
};

template <typename T, bool AOS, bool HAS_P, int VIEW>
int launch_scatter_tv(const ScatterArgs& a) {
  const EventsView& ev = *a.ev;
  const u64 n = ev.n;
  const bool vec16 = !AOS && aligned(ev.x, 16) && aligned(ev.y, 16) && aligned(ev.t, 16) && (!HAS_P || aligned(ev.p, 16));
  const bool vec = !AOS && aligned(ev.x, 8) && aligned(ev.y, 8) && aligned(ev.t, 16) && (!HAS_P || aligned(ev.p, 8));
  // The tiled kernel pays a fixed price per block (copy the bands, clear + scan w_ts * xmap_h slots), so it needs blocks
  // of >= 1024 events whose time slice still fits the LDS window of w_ts X-map columns.  A frame of n events spreads over
  // xmap_w columns: a block of E events spans about E * xmap_w / n of them.  Dense frames (C-1M: 1 M events / 640
  // columns) get 4096-event blocks; sparse ones (ESL-like: 150 K events / 1080 columns, < 1 event per slot, nothing to
  // de-duplicate) go to the one-thread-per-event kernel, whose cost is proportional to n.
  const double max_ev = a.tb->xmap_w > 0 ? (a.w_ts - 1.5) * (double)n / (double)a.tb->xmap_w : 0.0;
  if (!a.direct && a.w_ts > 0 && a.w_x > 0 && max_ev >= 1024.0) {
    // vector-load variant: 16-byte aligned SoA columns with int64 t (the EventCD time type); everything else takes the
    // lane-strided loads (any alignment)
    constexpr bool kHasVec = !AOS && std::is_same<T, long long>::value;
    auto kern = k_scatter_tiled<T, AOS, HAS_P, VIEW, false>;
    if constexpr (kHasVec) {
      if (vec16) kern = k_scatter_tiled<T, AOS, HAS_P, VIEW, true>;
    }
    if (a.key32) {  // compact key frame (a.frame points at it)
      kern = k_scatter_tiled<T, AOS, HAS_P, VIEW, false, true>;
      if constexpr (kHasVec) {
        if (vec16) kern = k_scatter_tiled<T, AOS, HAS_P, VIEW, true, true>;
      }
    }
    // raise the kernel's dynamic-LDS cap once per (handle = device, kernel instantiation); gfx950: 160 KB / CU
    {
      int rc_lds = a.h->ensure_lds(reinterpret_cast<const void*>(kern), a.lds);
      if (rc_lds) return rc_lds;
    }
    unsigned threads = TILE_THREADS;
    while (threads > 1024 / TILE_EPT && (double)(threads * TILE_EPT) > max_ev) threads >>= 1;  // smallest block: 1024 events
    static const int force_threads = getenv("XM_K1_THREADS") ? atoi(getenv("XM_K1_THREADS")) : 0;  // experiments
    if (force_threads >= 64 && force_threads <= TILE_THREADS && (force_threads & (force_threads - 1)) == 0) threads = force_threads;
    XM_LAUNCH(kern, dim3(grid_for(n, threads * TILE_EPT)), dim3(threads), a.lds, a.stream, ev.x, ev.y,
              (const T*)ev.t, ev.p, (const uint4*)ev.aos, n, a.idx_offset, *a.tb, a.st, a.tag_override,
              a.mm_lo, a.mm_hi, a.mm_ext, a.frame, a.dirty, a.w_ts, a.w_x, a.sorted ? 1 : 0);
    return XM_OK;
  }
  if constexpr (AOS) {
    XM_LAUNCH((k_scatter<T, true, HAS_P, 1, VIEW>), dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, a.stream,
              (const uint16_t*)nullptr, (const uint16_t*)nullptr, (const T*)nullptr, (const int16_t*)nullptr,
              (const uint4*)ev.aos, n, a.idx_offset, *a.tb, a.st, a.tag_override, a.mm_lo, a.mm_hi, a.mm_ext, a.frame, a.dirty,
              a.sorted ? 1 : 0);
  } else if (vec) {
    XM_LAUNCH((k_scatter<T, false, HAS_P, 4, VIEW>), dim3(grid_for(n, BLOCK * 4)), dim3(BLOCK), 0, a.stream,
              ev.x, ev.y, (const T*)ev.t, ev.p, (const uint4*)nullptr, n, a.idx_offset, *a.tb, a.st,
              a.tag_override, a.mm_lo, a.mm_hi, a.mm_ext, a.frame, a.dirty, a.sorted ? 1 : 0);
  } else {
    XM_LAUNCH((k_scatter<T, false, HAS_P, 1, VIEW>), dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, a.stream, ev.x,
              ev.y, (const T*)ev.t, ev.p, (const uint4*)nullptr, n, a.idx_offset, *a.tb, a.st, a.tag_override,
              a.mm_lo, a.mm_hi, a.mm_ext, a.frame, a.dirty, a.sorted ? 1 : 0);
  }
  return XM_OK;
}

template <typename T, bool AOS, bool HAS_P>
int launch_scatter_t(const ScatterArgs& a) {
  return a.view == XM_VIEW_PROJECTOR ? launch_scatter_tv<T, AOS, HAS_P, 0>(a) : launch_scatter_tv<T, AOS, HAS_P, 1>(a);
}

int launch_scatter(xm_handle* h, const EventsView& ev, SlotState* st, u32 tag_override, u64 idx_offset, u64 mm_lo,
                   u64 mm_hi, u64* frame, unsigned char* dirty, hipStream_t stream, bool sorted = false,
                   const void* mm_ext = nullptr, bool key32 = false) {
  ScatterArgs a{h, &ev, &h->tb, h->cfg.view, st, tag_override, idx_offset, mm_lo, mm_hi, mm_ext, frame, dirty, stream,
                h->w_ts, h->w_x, h->k1_lds, h->k1_direct, sorted, key32};
  if (ev.aos) return ev.use_p ? launch_scatter_t<long long, true, true>(a) : launch_scatter_t<long long, true, false>(a);
  switch (ev.t_dtype) {
    case XM_T_INT64: return ev.use_p ? launch_scatter_t<long long, false, true>(a) : launch_scatter_t<long long, false, false>(a);
    case XM_T_FLOAT32: return ev.use_p ? launch_scatter_t<float, false, true>(a) : launch_scatter_t<float, false, false>(a);
    default: return ev.use_p ? launch_scatter_t<double, false, true>(a) : launch_scatter_t<double, false, false>(a);
  }
}

// ---- column-tile K1 (xmaps_k1cols.hpp) ------------------------------------------------------------------------------------------
// kmode of a frame: 0 = 64-bit key frame (general), 1 = compact 32-bit key frame, 2 = column tiles + plain u16 frame
enum { KM_KEY64 = 0, KM_KEY32 = 1, KM_COLS = 2 };

// ---- K2 launches (tiled frame kernel, projector view) -----------------------------------------------------------------------
// Pixels per thread of a launch over n_frames frames: two (32 x 16-pixel tiles) when the launch fills the chip several times
// over -- a K2 wave is a chain of dependent round trips, what it costs there is resident waves x lifetime, so each wave carries
// two pixels through the chain --, one (16 x 16) for a lone small frame, where the chain's length IS the kernel's duration and
// twice the blocks start at once (C-1M, one frame: 7.3 us with one pixel per thread, 12.3 with two).
int k2_ppt(const xm_handle* h, int n_frames) {
  if (h->k2_force_ppt == 1 || h->k2_force_ppt == 2) return h->k2_force_ppt;
  const u64 blocks2 = (u64)grid_for(h->tb.proj_w, 2 * K2_TX) * grid_for(h->tb.proj_h, K2_TY) * (u64)std::max(n_frames, 1);
  // measured at C-1M (600 blocks of 32 x 16 pixels per frame, tools/ppt_threshold.sh), K2 us per launch with one / two pixels per
  // thread: 1 frame 5.3 / 7.2, 2 frames 8.7 / 8.6, 3: 11.1 / 10.6, 8: 23.4 / 21.3, 16: 47.5 / 41.1 -- the crossover is at about
  // four blocks per CU
  return blocks2 >= 1024 ? 2 : 1;
}

// K2's dynamic LDS: the tile's patch of u16 disparities (the row maxima replace it in place) + the overrun of its last read
size_t k2_lds_bytes(const xm_handle* h, int ppt) {
#ifdef XM_K2_TWO_BUFFERS
  return (size_t)(2 * h->k2_tile_cap[ppt - 1] + 16) * sizeof(uint16_t);
#else
  return (size_t)(h->k2_tile_cap[ppt - 1] + 32) * sizeof(uint16_t);
#endif
}

template <int FMT>
void launch_k2(xm_handle* h, hipStream_t stream, const u64* key_frame, SlotState* st, u32 tag_override, const unsigned char* dirty,
               float* depth, uint8_t* bgr, bool unsheared = false, int col_lo = 0, int col_hi = 0) {
  const int ppt = k2_ppt(h, 1);
  const dim3 grid(grid_for(h->tb.proj_w, K2_TX * ppt), grid_for(h->tb.proj_h, K2_TY));
  DevTables tb = h->tb;
  if (unsheared) tb.shear_m = tb.shear_bias = tb.shear_extra = 0;  // a plain [rect_w][rect_h] u16 frame (shards), not a slot's frame16
  if (ppt == 1)
    XM_LAUNCH((k_frame_proj_tiled<FMT, 1>), grid, dim3(K2_TX * K2_TY), k2_lds_bytes(h, 1), stream, key_frame, tb, st, tag_override,
              dirty, (const ulonglong2*)h->d_zero16, depth, bgr, h->k2_tile_cap[0], col_lo, col_hi);
  else
    XM_LAUNCH((k_frame_proj_tiled<FMT, 2>), grid, dim3(K2_TX * K2_TY), k2_lds_bytes(h, 2), stream, key_frame, tb, st, tag_override,
              dirty, (const ulonglong2*)h->d_zero16, depth, bgr, h->k2_tile_cap[1], col_lo, col_hi);
}

// the software-pipelined K2 (persistent blocks walking (frame, tile) items): groups on the plain u16 frame, two pixels per thread
size_t k2_pipe_lds_bytes(const xm_handle* h, int g) {
  return (size_t)((h->k2_tile_cap[g] + 32 + 7) & ~7) * sizeof(uint16_t) + (size_t)h->k2_pipe_nlds * sizeof(uint2);
}

bool launch_k2_pipe(xm_handle* h, hipStream_t stream, const FrameDesc* d_descs, int n_frames) {
  if (!h->k2_pipe || !h->k2_pipe_rig_ok || h->k2_pipe_nlds < 1 || k2_ppt(h, n_frames) != 2) return false;
  const int g = h->k2_pipe4 ? 2 : 1, ppt = 1 << g;
  const u32 gx = grid_for(h->tb.proj_w, K2_TX * ppt), gy = grid_for(h->tb.proj_h, K2_TY);
  const u64 total = (u64)gx * gy * (u64)n_frames;
  const size_t lds = k2_pipe_lds_bytes(h, g);
  static const int bpc_env = getenv("XM_K2_PIPE_BPC") ? atoi(getenv("XM_K2_PIPE_BPC")) : 0;  // experiments: blocks per CU
  const unsigned per_cu = bpc_env > 0 ? (unsigned)bpc_env : (unsigned)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / (lds + 2048)));
  unsigned blocks = (unsigned)h->n_cus * per_cu / 8 * 8;
  if (total < 3ull * blocks) return false;  // too few items per block for the pipeline to matter: one block per tile
  const void* fn = g == 2 ? reinterpret_cast<const void*>(k_frame_proj_pipe<4>) : reinterpret_cast<const void*>(k_frame_proj_pipe<2>);
  if (h->ensure_lds(fn, lds) != XM_OK) return false;
  K2PipeArgs pa;
  pa.proj_w = h->tb.proj_w; pa.proj_h = h->tb.proj_h; pa.rect_w = h->tb.rect_w; pa.rect_h = h->tb.rect_h;
  pa.shear_m = h->tb.shear_m; pa.shear_bias = h->tb.shear_bias;
  if (g == 2)
    XM_LAUNCH((k_frame_proj_pipe<4>), dim3(blocks), dim3(K2_TX * K2_TY), lds, stream, d_descs, (const int4*)h->d_k2_tiles[2],
              (const u32*)h->d_k2_pix[2], h->tb.dlut, pa, h->k2_tile_cap[2], (u32)n_frames, gx, gy, h->k2_pipe_nlds);
  else
    XM_LAUNCH((k_frame_proj_pipe<2>), dim3(blocks), dim3(K2_TX * K2_TY), lds, stream, d_descs, h->tb.k2_tiles, h->tb.k2_pix, h->tb.dlut, pa,
              h->k2_tile_cap[1], (u32)n_frames, gx, gy, h->k2_pipe_nlds);
  return true;
}

template <int FMT, int COND = 0>
void launch_k2_batch(xm_handle* h, hipStream_t stream, const FrameDesc* d_descs, int n_frames) {
  if constexpr (FMT == 2 && COND == 0) {
    if (launch_k2_pipe(h, stream, d_descs, n_frames)) return;
  }
  const int ppt = k2_ppt(h, n_frames);
  dim3 grid(grid_for(h->tb.proj_w, K2_TX * ppt), grid_for(h->tb.proj_h, K2_TY), n_frames);
  if (COND == 1) grid = dim3(std::min(grid.x * grid.y, 32u), 1, n_frames);  // redo node: a few blocks per frame walk its tiles
  if (ppt == 1)
    XM_LAUNCH((k_frame_proj_tiled_batch<FMT, COND, 1>), grid, dim3(K2_TX * K2_TY), k2_lds_bytes(h, 1), stream, d_descs, h->tb,
              (const ulonglong2*)h->d_zero16, h->k2_tile_cap[0]);
  else
    XM_LAUNCH((k_frame_proj_tiled_batch<FMT, COND, 2>), grid, dim3(K2_TX * K2_TY), k2_lds_bytes(h, 2), stream, d_descs, h->tb,
              (const ulonglong2*)h->d_zero16, h->k2_tile_cap[1]);
}

size_t cols_lds_bytes(const xm_handle* h, int W) {  // mirrors the carve-up at the top of scatter_cols_body
  const size_t lut_q = ((size_t)h->w_x * h->tb.cam_h + 3) / 4 + 1 + 64, xm_q = ((size_t)W * h->tb.xmap_h + 7) / 8 + 1 + 64,
               slot_q = ((size_t)W * h->tb.xmap_h + 3) / 4;
  return 16 * (lut_q + xm_q + slot_q);
}

size_t own_plan_lds_bytes(int nxs_max, int hrp, int extra_max) {  // mirrors the carve-up at the top of scatter_own_body
  return (size_t)4 * nxs_max * hrp + (size_t)4 * extra_max + (size_t)4 * hrp;
}
size_t own_lds_bytes(const xm_handle* h) { return own_plan_lds_bytes(h->tb.own_nxs_max, h->tb.own_hrp, h->tb.own_extra_max); }

// ---- owner tiles (xmaps_k1own.hpp): the rig's ownership tables, worked out once on the host -------------------------------------
struct OwnPlan {  // what own_plan() works out (host memory) and own_setup() uploads
  bool ok = false, all_in = true;
  int W = 0, halo = 0, r_lo = 0, hr = 0, hrp = 0, nxs_max = 0, extra_max = 0, m = 0, bias = 0, extra_cols = 0, delta_max = 0;
  std::vector<uint16_t> packed, xextra, masks;
  std::vector<int4> tiles;
  std::vector<int16_t> bases;
  std::vector<u32> extra_flat;
};

// Pure host code (no device needed: xm_own_plan_info runs it for the CPU tests).  pl.ok says whether the rig qualifies.
void own_plan(const xm_config* cfg, int xmap_h, int xr_min, OwnPlan& pl) {
  const int xmap_w = cfg->xmap_width, rect_w = cfg->rect_width, rect_h = cfg->rect_height, x_off = cfg->x_offset;
  const int rows = std::min(xmap_h - 1, rect_h);
  if (rows <= 0 || rect_h < xmap_h - 1 || xr_min <= -x_off) return;  // (an undefined X-map cell, 0, must read as dead)
  int yr_min = 32767, yr_max = -32768;
  const size_t cam_px = (size_t)cfg->cam_width * cfg->cam_height;
  for (size_t i = 0; i < cam_px; ++i) {
    yr_min = std::min<int>(yr_min, cfg->cam_mapy_i16[i]);
    yr_max = std::max<int>(yr_max, cfg->cam_mapy_i16[i]);
  }
  const int r_lo = std::max(0, yr_min) & ~7, r_hi = std::min(yr_max, rows - 1);
  if (r_hi < r_lo) return;
  const int hr = r_hi - r_lo + 1, hrp = (hr + 7) & ~7;
  int W = 8;
  if (const char* e = getenv("XM_OWN_W")) W = atoi(e);
  W = std::max(OWN_BW, std::min(W, 64)) / OWN_BW * OWN_BW;
  // 1. owner column of every cell, row by row: delta = column - first column of the row that maps to the same cell
  std::vector<uint16_t>& packed = pl.packed;  // [c][row], as tb.xmap
  packed.assign((size_t)xmap_w * xmap_h, 0);
  std::vector<int> first(rect_w, -1), fcs((size_t)xmap_w);
  int delta_max = 0;
  bool all_in = true;
  for (int r = r_lo; r <= r_hi; ++r) {
    const int16_t* X = cfg->proj_x_map + (size_t)r * xmap_w;
    for (int c = 0; c < xmap_w; ++c) {
      fcs[c] = -1;
      const int xp = X[c], fu = xp - x_off;
      if (fu < xr_min) continue;  // dead: no LUT entry gives disp >= 0
      if (xp < 0 || xp >= (1 << OWN_XP_BITS)) return;
      int fc = (int)(short)fu;
      if (fc < 0) fc += rect_w;  // NumPy's negative wrap
      const bool in = fc >= 0 && fc < rect_w;
      if (!in || fu < 0) all_in = false;  // the kernel's lean path takes fu as the column
      int delta = 0;
      if (in) {
        if (first[fc] < 0) first[fc] = c;
        delta = c - first[fc];
        fcs[c] = fc;
      }
      if (delta > OWN_MAX_DELTA) return;
      delta_max = std::max(delta_max, delta);
      packed[(size_t)c * xmap_h + r] = (uint16_t)(xp | (delta << OWN_XP_BITS));
    }
    for (int c = 0; c < xmap_w; ++c)
      if (fcs[c] >= 0) first[fcs[c]] = -1;
  }
  const int halo = (delta_max + OWN_BW - 1) / OWN_BW * OWN_BW;
  // 2. the shear: slope of the cell column against the row along the middle time columns (least squares over the live entries)
  double slope = 0.0;
  {
    double sn = 0, sx = 0, sy = 0, sxx = 0, sxy = 0;
    for (int c = xmap_w / 4; c < xmap_w; c += std::max(1, xmap_w / 4)) {
      sn = sx = sy = sxx = sxy = 0;
      for (int r = r_lo; r <= r_hi; ++r) {
        const int fu = cfg->proj_x_map[(size_t)r * xmap_w + c] - x_off;
        if (fu < xr_min || fu < 0 || fu >= rect_w) continue;
        sn += 1; sx += r; sy += fu; sxx += (double)r * r; sxy += (double)r * fu;
      }
      if (sn >= 16 && sn * sxx - sx * sx > 0) {
        slope = (sn * sxy - sx * sy) / (sn * sxx - sx * sx);
        if (c >= xmap_w / 2) break;  // prefer the middle column
      }
    }
  }
  int m = 0;
  if (std::fabs(slope) * hr >= 24.0) m = (int)std::lround(-slope * 8.0 * 4096.0);
  if (const char* e = getenv("XM_OWN_SHEAR")) m = atoi(e);  // experiments
  int sh_min = 0, sh_max = 0;
  for (int g = 0; g <= (rect_h - 1) >> 3; ++g) {
    const int sh = (g * m) >> 12;
    sh_min = std::min(sh_min, sh);
    sh_max = std::max(sh_max, sh);
  }
  const int bias = -sh_min, extra = sh_max - sh_min;
  if (rect_w + extra > 32767) return;
  // 3. per (tile, row): where its cells lie in the sheared frame.  The band of a row = the window of NX frame columns that
  //    holds most of the row's owner cells (a tile's cells of one row are a short run; the run moves with the row by what the
  //    frame's shear leaves of the X-map's slant); owner cells outside it are "extras" (where the rectified
  //    time map replicates its border the X-map jumps by hundreds of columns: first / last tile of the ESL rig).  NX = the
  //    narrowest band that leaves (almost) no more extras than the widest one.
  const int nt = (xmap_w + W - 1) / W, ng = hrp;  // (one band position per row)
  const auto owner_col = [&](int r, int c, int& xs) {  // owner pairs only: the cell's column in the sheared frame
    const uint16_t pk = packed[(size_t)c * xmap_h + r];
    if (!pk || (pk >> OWN_XP_BITS) != 0) return false;
    int fc = (int)(short)((int)(pk & ((1 << OWN_XP_BITS) - 1)) - x_off);
    if (fc < 0) fc += rect_w;
    if (fc < 0 || fc >= rect_w) return false;
    xs = fc + bias + (((r >> 3) * m) >> 12);
    return true;
  };
  std::vector<std::vector<int>> cells((size_t)nt * ng);  // sorted sheared columns of every (tile, row)'s owner cells
  for (int r = r_lo; r <= r_hi; ++r)
    for (int c = 0; c < xmap_w; ++c) {
      int xs;
      if (owner_col(r, c, xs)) cells[(size_t)(c / W) * ng + (r - r_lo)].push_back(xs);
    }
  for (auto& v : cells) std::sort(v.begin(), v.end());
  const auto best_window = [](const std::vector<int>& v, int nx, int& start) {  // most cells inside [start, start + nx)
    size_t best = 0, j = 0;
    start = v.empty() ? 0 : v[0];
    for (size_t i = 0; i < v.size(); ++i) {
      if (i && v[i] == v[i - 1]) continue;
      while (j < v.size() && v[j] < v[i] + nx) ++j;
      if (j - i > best) {
        best = j - i;
        start = v[i];
      }
    }
    return best;
  };
  size_t extras_at[OWN_MAX_NXS + 1] = {};
  for (int nx = 1; nx <= OWN_MAX_NXS; ++nx)
    for (const auto& v : cells) {
      int st;
      extras_at[nx] += v.size() - best_window(v, nx, st);
    }
  int NX = OWN_MAX_NXS;
  while (NX > 1 && extras_at[NX - 1] <= extras_at[OWN_MAX_NXS] + extras_at[OWN_MAX_NXS] / 8 + 64) NX -= 1;
  if (const char* e = getenv("XM_OWN_NX")) NX = std::max(1, std::min(atoi(e), (int)OWN_MAX_NXS));  // experiments
  std::vector<int4>& tiles = pl.tiles;
  std::vector<int16_t>& bases = pl.bases;
  tiles.assign(nt, make_int4(0, 0, 0, 0));
  bases.assign((size_t)nt * ng, 0);
  for (int t = 0; t < nt; ++t)
    for (int g = 0; g < ng; ++g) {
      int st;
      best_window(cells[(size_t)t * ng + g], NX, st);
      bases[(size_t)t * ng + g] = (int16_t)st;
    }
  std::vector<uint16_t>&masks = pl.masks, &xextra = pl.xextra;
  masks.assign((size_t)nt * hrp, 0);
  xextra.assign((size_t)xmap_w * xmap_h, 0);
  std::vector<std::vector<u32>> extra_cells(nt);
  for (int r = r_lo; r <= r_hi; ++r)
    for (int c = 0; c < xmap_w; ++c) {
      int xs;
      if (!owner_col(r, c, xs)) continue;
      const int t = c / W, k = xs - bases[(size_t)t * ng + (r - r_lo)];
      if (k >= 0 && k < NX) {
        masks[(size_t)t * hrp + (r - r_lo)] |= (uint16_t)(1u << k);
        tiles[t].x = std::max(tiles[t].x, k + 1);
      } else {
        extra_cells[t].push_back((u32)xs * (u32)rect_h + (u32)r);
        if (extra_cells[t].size() > 4096) return;  // (a wild X-map: the packed keys stay)
        xextra[(size_t)c * xmap_h + r] = (uint16_t)extra_cells[t].size();
      }
    }
  int nxs_max = 1, extra_max = 0;
  std::vector<u32>& extra_flat = pl.extra_flat;
  extra_flat.clear();
  for (int t = 0; t < nt; ++t) {
    tiles[t].y = (int)extra_flat.size();
    tiles[t].z = (int)extra_cells[t].size();
    extra_flat.insert(extra_flat.end(), extra_cells[t].begin(), extra_cells[t].end());
    nxs_max = std::max(nxs_max, tiles[t].x);
    extra_max = std::max(extra_max, tiles[t].z);
  }
  extra_max = (extra_max + 3) & ~3;
  if (own_plan_lds_bytes(nxs_max, hrp, extra_max) > 60 * 1024) return;  // LDS per block
  extra_flat.push_back(0);
  pl.W = W; pl.halo = halo; pl.r_lo = r_lo; pl.hr = hr; pl.hrp = hrp; pl.nxs_max = nxs_max; pl.extra_max = extra_max;
  pl.m = m; pl.bias = bias; pl.extra_cols = extra; pl.delta_max = delta_max; pl.all_in = all_in;
  pl.ok = true;
}

// Returns XM_OK whether or not the rig qualifies (h->own_mode says); an error only for HIP failures.
int own_setup(xm_handle* h, const xm_config* cfg, int xr_min) {
  OwnPlan pl;
  own_plan(cfg, h->tb.xmap_h, xr_min, pl);
  if (!pl.ok) return XM_OK;
  const auto up = [](auto** dst, const auto& v) -> hipError_t {
    typedef typename std::remove_reference<decltype(v)>::type::value_type E;
    hipError_t e = hipMalloc((void**)dst, v.size() * sizeof(E) + 64);
    return e != hipSuccess ? e : hipMemcpy(*dst, v.data(), v.size() * sizeof(E), hipMemcpyHostToDevice);
  };
  HIP_TRY(up(&h->d_xmap_own, pl.packed));
  HIP_TRY(up(&h->d_xmap_extra, pl.xextra));
  HIP_TRY(up(&h->d_own_tiles, pl.tiles));
  HIP_TRY(up(&h->d_own_base, pl.bases));
  HIP_TRY(up(&h->d_own_masks, pl.masks));
  HIP_TRY(up(&h->d_own_extra_cells, pl.extra_flat));
  h->own_extras = (int)pl.extra_flat.size() - 1;
  h->tb.xmap_own = h->d_xmap_own;
  h->tb.xmap_extra = h->d_xmap_extra;
  h->tb.own_tiles = h->d_own_tiles;
  h->tb.own_base = h->d_own_base;
  h->tb.own_masks = h->d_own_masks;
  h->tb.own_extra_cells = h->d_own_extra_cells;
  h->tb.own_r_lo = pl.r_lo;
  h->tb.own_hr = pl.hr;
  h->tb.own_hrp = pl.hrp;
  h->tb.own_nxs_max = pl.nxs_max;
  h->tb.own_extra_max = pl.extra_max;
  h->tb.shear_m = pl.m;
  h->tb.shear_bias = pl.bias;
  h->tb.shear_extra = pl.extra_cols;
  h->own_mode = true;
  h->own_w = pl.W;
  h->own_halo = pl.halo;
  if (pl.all_in) h->cols_flags |= COLS_F_ALL_IN_FRAME;
  return XM_OK;
}

// time columns per tile for frames of n events: about cols_target events per tile, within the LDS budget; 0 = not this path
int cols_width(const xm_handle* h, u64 n) {
  if (h->cols_ok && h->own_mode) {  // owner tiles: one fixed width (the ownership tables are built for it); not for nearly empty frames
    const u64 tiles = grid_for(h->tb.xmap_w, h->own_w);
    return n >= tiles * 128 && n < (1ull << 28) ? h->own_w : 0;
  }
  if (!h->cols_ok || h->cols_w_max < 1 || h->tb.xmap_w < 1 || n == 0 || n >= (1ull << 28)) return 0;
  const double per_col = (double)n / (double)h->tb.xmap_w;
  int W = (int)((double)h->cols_target / per_col);
  W = std::max(1, std::min(W, h->cols_w_max));
  if (per_col * W < 1024.0) return 0;  // sparse frames: the band copies and the slot scan would dominate (direct kernel instead)
  return W;
}

unsigned cols_threads(const xm_handle* h, u64 n, int W) {
  static const int force = getenv("XM_COLS_THREADS") ? atoi(getenv("XM_COLS_THREADS")) : 0;  // experiments
  if (force >= 64 && force <= COLS_MAX_THREADS && force % 64 == 0) return (unsigned)force;
  if (h->own_mode) {  // own + halo columns in one pass where 512 threads hold them
    const double per = (double)n / (double)h->tb.xmap_w * (W + h->own_halo);
    const unsigned t = ((unsigned)(per * 1.12 / COLS_EPT) + 63u) / 64u * 64u;
    return std::max(128u, std::min(t, (unsigned)COLS_MAX_THREADS));
  }
  const double per_tile = (double)n / (double)h->tb.xmap_w * W;
  // one pass for a tile 12 % above the mean (Poisson spread of an evenly filled scan); fuller tiles take a second pass.
  // Tiles of more than 2048 events get the full 512 threads even when 448 would hold them: three blocks per CU are then
  // 24 waves = every wave slot the kernel's 80 VGPRs allow (measured at C-1M, 3125 events per tile: 4.35 instead of 4.63 us
  // per frame at full occupancy; 384 threads = two passes: 5.9 us)
  unsigned t = ((unsigned)(per_tile * 1.12 / COLS_EPT) + 63u) / 64u * 64u;
  if (t > 256u) t = COLS_MAX_THREADS;
  return std::max(128u, std::min(t, (unsigned)COLS_MAX_THREADS));
}

// K0b: the tile boundaries + column thresholds of the frame (half a wave per boundary), left behind the slot's u16 frame
void launch_cols_bounds(xm_handle* h, const EventsView& ev, uint16_t* frame16, int W, hipStream_t stream) {
  if (h->own_mode) W = OWN_BW;  // owner tiles: boundaries every OWN_BW columns (tile = own_w columns + a halo behind them)
  const unsigned nb = grid_for(h->tb.xmap_w, W);
  if (ev.aos)
    XM_LAUNCH(k_cols_bounds<true>, dim3(grid_for(nb + 1, COLS_BOUNDS_PER_BLOCK)), dim3(256), 0, stream, ev.x,
              (const long long*)ev.t, (const uint4*)ev.aos, (u32)ev.n, h->tb, W, frame16);
  else
    XM_LAUNCH(k_cols_bounds<false>, dim3(grid_for(nb + 1, COLS_BOUNDS_PER_BLOCK)), dim3(256), 0, stream, ev.x,
              (const long long*)ev.t, (const uint4*)ev.aos, (u32)ev.n, h->tb, W, frame16);
}

int launch_scatter_cols(xm_handle* h, const EventsView& ev, SlotState* st, uint16_t* frame16, int W, hipStream_t stream) {
  const bool vec16 = !ev.aos && aligned(ev.x, 16) && aligned(ev.y, 16) && aligned(ev.t, 16);
  if (h->own_mode) {
    auto kern = k_scatter_own<false, false>;
    if (ev.aos) kern = k_scatter_own<true, false>;
    else if (vec16) kern = k_scatter_own<false, true>;
    const size_t lds = own_lds_bytes(h);
    int rc = h->ensure_lds(reinterpret_cast<const void*>(kern), lds);
    if (rc) return rc;
    XM_LAUNCH(kern, dim3(grid_for(h->tb.xmap_w, W)), dim3(cols_threads(h, ev.n, W)), lds, stream, ev.x, ev.y, (const long long*)ev.t,
              (const uint4*)ev.aos, (u32)ev.n, h->tb, st, frame16, W, h->own_halo, h->cols_flags);
    return XM_OK;
  }
  auto kern = k_scatter_cols<false, false>;
  if (ev.aos) kern = k_scatter_cols<true, false>;
  else if (vec16) kern = k_scatter_cols<false, true>;
  const size_t lds = cols_lds_bytes(h, W);
  int rc = h->ensure_lds(reinterpret_cast<const void*>(kern), lds);
  if (rc) return rc;
  XM_LAUNCH(kern, dim3(grid_for(h->tb.xmap_w, W)), dim3(cols_threads(h, ev.n, W)), lds, stream, ev.x, ev.y, (const long long*)ev.t,
            (const uint4*)ev.aos, (u32)ev.n, h->tb, st, frame16, W, h->w_x, h->cols_xr_min, h->cols_flags);
  return XM_OK;
}

void launch_frame_kernel(xm_handle* h, const u64* key_frame, SlotState* st, u32 tag_override, float* depth,
                         uint8_t* bgr, hipStream_t stream, const unsigned char* dirty = nullptr, int kmode = KM_KEY64) {
  KeyCells cells{key_frame, 0};
  const bool key32 = kmode == KM_KEY32;
  if (h->cfg.view == XM_VIEW_PROJECTOR && !h->k2_direct && kmode == KM_COLS) {
    launch_k2<2>(h, stream, key_frame, st, tag_override, nullptr, depth, bgr);
  } else if (h->cfg.view == XM_VIEW_PROJECTOR && !h->k2_direct && key32) {
    launch_k2<1>(h, stream, key_frame, st, tag_override, nullptr, depth, bgr);
  } else if (h->cfg.view == XM_VIEW_PROJECTOR && !h->k2_direct) {
    launch_k2<0>(h, stream, key_frame, st, tag_override, dirty, depth, bgr);
  } else if (h->cfg.view == XM_VIEW_PROJECTOR) {
    const u64 px = (u64)h->tb.proj_w * h->tb.proj_h;
    XM_LAUNCH((k_frame_proj<KeyCells, 0>), dim3(grid_for(px, BLOCK)), dim3(BLOCK), 0, stream, cells, h->tb, st,
              tag_override, depth, bgr);
  } else if (key32) {  // camera view, compact frame: (event index + 1) << 12 | disparity, zeroed as it is read
    const u64 px = (u64)h->tb.cam_w * h->tb.cam_h;
    XM_LAUNCH(k_frame_cam32, dim3(grid_for(px, BLOCK)), dim3(BLOCK), 0, stream, reinterpret_cast<u32*>(const_cast<u64*>(key_frame)), px, st,
              h->tb.dlut, depth, bgr);
  } else {
    const u64 px = (u64)h->tb.cam_w * h->tb.cam_h;
    XM_LAUNCH((k_frame_direct<KeyCells>), dim3(grid_for(px, BLOCK)), dim3(BLOCK), 0, stream, cells, px,
              h->tb.p03, h->tb.z_near, h->tb.z_far, st, tag_override, 1, h->tb.dlut, depth, bgr);
  }
}

int check_events(const EventsView& ev) {
  if (ev.n >= XM_KEY_MAX_EVENTS) return fail(XM_ERR_TOO_MANY, "frame of %zu events exceeds 2^%d", ev.n, XM_KEY_IDX_BITS);
  if (ev.n == 0) return XM_OK;
  if (ev.aos) {
    if (!aligned(ev.aos, 16)) return fail(XM_ERR_INVALID, "EventCD buffer must be 16-byte aligned");
    return XM_OK;
  }
  if (!ev.x || !ev.y || !ev.t) return fail(XM_ERR_INVALID, "x, y, t must be non-NULL when n > 0");
  if (ev.t_dtype != XM_T_INT64 && ev.t_dtype != XM_T_FLOAT32 && ev.t_dtype != XM_T_FLOAT64)
    return fail(XM_ERR_INVALID, "unknown t_dtype %d", ev.t_dtype);
  if (!aligned(ev.t, t_size(ev.t_dtype)) || !aligned(ev.x, 2) || !aligned(ev.y, 2) || (ev.p && !aligned(ev.p, 2)))
    return fail(XM_ERR_INVALID, "event columns must be naturally aligned");
  return XM_OK;
}

// enqueue K0 -> K1 -> K2 for one frame on a slot.  All pointers are device pointers.
// dense enough for the tiled K1?  (the same rule as launch_scatter_tv)
bool tiled_path(const xm_handle* h, u64 n) {
  const double max_ev = h->tb.xmap_w > 0 ? (h->w_ts - 1.5) * (double)n / (double)h->tb.xmap_w : 0.0;
  return !h->k1_direct && h->w_ts > 0 && h->w_x > 0 && max_ev >= 1024.0;
}

bool sorted_path(const xm_handle* h, const EventsView& ev) {
  // the verified (t[0], t[n-1]) shortcut: both K1 kernels take it (tiled, and one thread per event for sparse frames)
  return (h->time_sorted || (h->try_sorted && !h->capturing)) && !ev.use_p && ev.n > 0;
}

// may this (sorted-path) frame use the compact key frame?  Needs the automatic redo (try-sorted mode, not inside a capture)
bool key32_path(const xm_handle* h, const EventsView& ev, bool sorted) {
  if (!sorted || !h->key32_ok || !h->try_sorted || h->capturing || h->key32_pause.load(std::memory_order_relaxed) > 0 ||
      h->k2_direct || h->k2_flags || !tiled_path(h, ev.n))
    return false;
  if (h->cfg.view != XM_VIEW_PROJECTOR) return ev.n <= (u64)CAM32_MAX_EVENTS;  // the key's order field is the event index
  return ev.n / (u64)(1024 / TILE_EPT * TILE_EPT) < (1ull << KEY32_TILE_BITS);  // tiles of >= 1024 events
}

// may this (sorted-path) frame use the column tiles?  Same preconditions as the compact key frame (automatic redo at hand)
// + int64 time stamps; returns the tile width W (0: no)
int cols_path(const xm_handle* h, const EventsView& ev, bool sorted, bool group = true) {
  if (!group && !h->cols_single) return 0;
  if (!sorted || !h->cols_ok || !h->try_sorted || h->capturing || h->key32_pause.load(std::memory_order_relaxed) > 0 ||
      h->k2_direct || h->k2_flags || ev.use_p || (!ev.aos && ev.t_dtype != XM_T_INT64))
    return 0;
  return cols_width(h, ev.n);
}

// keep the slot's compact frame unambiguous for a frame with tag `tag` (4-bit tags repeat every 15 frames)
int key32_prepare(xm_handle* h, Slot& s, u32 tag, hipStream_t stream) {
  if (h->cfg.view != XM_VIEW_PROJECTOR) return XM_OK;  // camera view: no tag -- the frame kernel zeroes every pixel it reads
  if (tag - s.key32_valid_from >= 15u || tag < s.key32_valid_from) {
    HIP_TRY(hipMemsetAsync(s.key32, 0, h->key_cells * sizeof(u32), stream));
    s.key32_valid_from = tag;
  }
  return XM_OK;
}

void key32_note(xm_handle* h, bool failed) {
  if (failed) {
    if (h->key32_score.fetch_add(8, std::memory_order_relaxed) + 8 >= 24) {  // the stream keeps producing events outside the
      h->key32_pause.store(512, std::memory_order_relaxed);                  // LDS time window (sparse / bursty frames)
      h->key32_score.store(0, std::memory_order_relaxed);
    }
  } else {
    int v = h->key32_score.load(std::memory_order_relaxed);
    while (v > 0 && !h->key32_score.compare_exchange_weak(v, v - 1, std::memory_order_relaxed)) {
    }
  }
}

int enqueue_frame(xm_handle* h, Slot& s, const EventsView& ev, float* depth, uint8_t* bgr, hipEvent_t* prof,
                  bool allow_sorted = true, hipStream_t stream_override = nullptr) {
  const bool sorted = allow_sorted && sorted_path(h, ev);
  const int cols_w = cols_path(h, ev, sorted, false);
  const bool use32 = !cols_w && key32_path(h, ev, sorted);
  {
    int v = h->key32_pause.load(std::memory_order_relaxed);
    while (v > 0 && !h->key32_pause.compare_exchange_weak(v, v - 1, std::memory_order_relaxed)) {
    }
  }
  hipStream_t stream = stream_override ? stream_override : s.stream;
  if (s.pending_batch_ev) {  // the slot's previous frame ran inside a multi-frame launch, maybe on another stream
    if (s.pending_batch_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, s.pending_batch_ev, 0));
    s.pending_batch_ev = nullptr;
  }
  if (s.host_tag >= KEY_MAX_TAG) {  // tag field about to wrap: clear the frame once per 2^19 frames
    int rc = reset_slot(h, s, stream);
    if (rc) return rc;
  }
#ifdef XM_ABLATE
  static const int skip = getenv("XM_SKIP_MASK") ? atoi(getenv("XM_SKIP_MASK")) : 0;  // experiments: 1=K0 2=K1 4=K2
#else
  constexpr int skip = 0;
#endif
  // prof = 6 events {start0, stop0, start1, stop1, start2, stop2} attached to the three dispatch packets
  if (prof) g_prof = ProfCtx{prof[0], prof[1]};
  if (!(skip & 1) && !sorted) launch_minmax(ev, s.st, 0, stream);
  if (!(skip & 1) && cols_w) launch_cols_bounds(h, ev, s.frame16, cols_w, stream);  // K0b takes K0's place (and its profile events)
  if (use32) {
    int rc = key32_prepare(h, s, s.host_tag + 1, stream);
    if (rc) return rc;
  }
  if (prof) g_prof = ProfCtx{prof[2], prof[3]};
  if (!(skip & 2)) {
    int rc = cols_w ? launch_scatter_cols(h, ev, s.st, s.frame16, cols_w, stream)
                    : launch_scatter(h, ev, s.st, 0, 0, 0, 0, use32 ? reinterpret_cast<u64*>(s.key32) : s.key_frame, s.dirty, stream,
                                     sorted, nullptr, use32);
    if (rc) {
      g_prof = ProfCtx{};
      return rc;
    }
  }
  if (prof) g_prof = ProfCtx{prof[4], prof[5]};
  if (!(skip & 4))
    launch_frame_kernel(h, cols_w ? reinterpret_cast<const u64*>(s.frame16) : use32 ? reinterpret_cast<const u64*>(s.key32) : s.key_frame,
                        s.st, 0, depth, bgr, stream, h->k2_flags ? s.dirty : nullptr, cols_w ? KM_COLS : use32 ? KM_KEY32 : KM_KEY64);
  g_prof = ProfCtx{};
  HIP_TRY(hipGetLastError());
  s.last_key32 = use32 || cols_w;
  s.last_cols = cols_w != 0;
  h->path_counts[cols_w ? 3 : use32 ? 2 : sorted ? 1 : 0].fetch_add(1, std::memory_order_relaxed);
  if (use32 || cols_w) key32_note(h, false);
  s.host_tag += 1;
  s.any_frame = true;
  s.last_n = ev.n;
  s.last_sorted = sorted;
  s.last_t_dtype = ev.aos ? XM_T_INT64 : ev.t_dtype;
  if (!stream_override) s.eager_dirty = true;
  return XM_OK;
}

// ---- multi-frame launches -------------------------------------------------------------------------------------------
// One K0 / K1 / K2 launch each for a whole group of frames (grid = frames x tiles).  Frame f of the group runs on slot
// slots[f] (its own key frame + state), all on ONE stream.  A single frame's launches leave the chip half empty while
// they ramp up and drain (245 K1 blocks for 256 CUs, each a ~10 us dependent chain); a group's launch keeps every CU fed.
template <typename T, bool AOS, bool HAS_P>
int launch_batch_t(xm_handle* h, const FrameDesc* d_descs, int n_frames, u64 n_max, u64 n_mean, bool vec16, bool sorted,
                   hipStream_t stream, bool key32 = false, int cols_w = 0, hipEvent_t* prof = nullptr,
                   const FrameDesc* d_descs_redo = nullptr, bool direct_k1 = false) {
  // prof = 6 events {start0, stop0, start1, stop1, start2, stop2} attached to the dispatch packets of K0 / K0b, K1, K2
  struct ProfReset {
    ~ProfReset() { g_prof = ProfCtx{}; }
  } prof_reset;
  const auto prof_slot = [&](int i) { if (prof) g_prof = ProfCtx{prof[2 * i], prof[2 * i + 1]}; };
  if constexpr (std::is_same<T, long long>::value && !HAS_P) {
    if (cols_w) {  // column tiles: K1 grid = (tiles, frames), K2 on the plain u16 frames
      int rc;
      if (h->own_mode) {  // owner tiles (the rig's X-map is not injective): boundaries every OWN_BW columns, tiles of own_w + halo
        auto kern = k_scatter_own_batch<AOS, false>;
        if constexpr (!AOS) {
          if (vec16) kern = k_scatter_own_batch<false, true>;
        }
        const size_t lds = own_lds_bytes(h);
        rc = h->ensure_lds(reinterpret_cast<const void*>(kern), lds);
        if (rc) return rc;
        prof_slot(0);
        XM_LAUNCH(k_cols_bounds_batch<AOS>, dim3(grid_for(grid_for(h->tb.xmap_w, OWN_BW) + 1, COLS_BOUNDS_PER_BLOCK), n_frames),
                  dim3(256), 0, stream, d_descs, h->tb, OWN_BW);
        prof_slot(1);
        XM_LAUNCH(kern, dim3(grid_for(h->tb.xmap_w, cols_w), n_frames), dim3(cols_threads(h, n_mean, cols_w)), lds, stream, d_descs, h->tb,
                  cols_w, h->own_halo, h->cols_flags | (d_descs_redo ? COLS_F_DEVICE_REDO : 0));
      } else {
      auto kern = k_scatter_cols_batch<AOS, false>;
      if constexpr (!AOS) {
        if (vec16) kern = k_scatter_cols_batch<false, true>;
      }
      const size_t lds = cols_lds_bytes(h, cols_w);
      rc = h->ensure_lds(reinterpret_cast<const void*>(kern), lds);
      if (rc) return rc;
      prof_slot(0);
      XM_LAUNCH(k_cols_bounds_batch<AOS>, dim3(grid_for(grid_for(h->tb.xmap_w, cols_w) + 1, COLS_BOUNDS_PER_BLOCK), n_frames),
                dim3(256), 0, stream, d_descs, h->tb, cols_w);
      prof_slot(1);
      XM_LAUNCH(kern, dim3(grid_for(h->tb.xmap_w, cols_w), n_frames), dim3(cols_threads(h, n_mean, cols_w)), lds, stream, d_descs, h->tb,
                cols_w, h->w_x, h->cols_xr_min, h->cols_flags | (d_descs_redo ? COLS_F_DEVICE_REDO : 0));
      }
      prof_slot(2);
      if (!d_descs_redo) {
        launch_k2_batch<2>(h, stream, d_descs, n_frames);
        HIP_TRY(hipGetLastError());
        return XM_OK;
      }
      // Captured batch (hipGraph): no host at hand to redo a frame whose tiles objected, so the graph carries both paths and the
      // kernels decide per frame on the device (frame_attempt_failed): K2 on the u16 frame only where the attempt held, then --
      // for the frames where it did not, and for those only: every other block returns at once -- the counters cleared and
      // K0 -> K1 -> K2 on the 64-bit key frame (d_descs_redo = the same frames with key_frame = the slots' 64-bit frames).
      launch_k2_batch<2, 2>(h, stream, d_descs, n_frames);
      g_prof = ProfCtx{};
      XM_LAUNCH(k_redo_prepare_batch, dim3(n_frames), dim3(64), 0, stream, d_descs_redo);
      {
        const bool vec2 = !AOS && vec16;
        const unsigned per_block = BLOCK * (vec2 ? 2 * K0_UN : 4);
        unsigned gx = grid_for(n_max, per_block);
        if (gx > 64) gx = 64;  // (grid-stride kernel; a redo node: usually every block returns at once)
        if constexpr (!AOS) {
          if (vec2) XM_LAUNCH((k_minmax_batch<T, false, false, 2, 1>), dim3(gx, n_frames), dim3(BLOCK), 0, stream, d_descs_redo);
          else XM_LAUNCH((k_minmax_batch<T, false, false, 1, 1>), dim3(gx, n_frames), dim3(BLOCK), 0, stream, d_descs_redo);
        } else {
          XM_LAUNCH((k_minmax_batch<T, true, false, 1, 1>), dim3(gx, n_frames), dim3(BLOCK), 0, stream, d_descs_redo);
        }
      }
      {
        const double max_ev = (h->w_ts - 1.5) * (double)n_mean / (double)h->tb.xmap_w;
        unsigned threads = TILE_THREADS;
        while (threads > 1024 / TILE_EPT && (double)(threads * TILE_EPT) > max_ev) threads >>= 1;
        auto k1 = k_scatter_tiled_batch<T, AOS, false, 0, false, false, 1>;
        if constexpr (!AOS) {
          if (vec16) k1 = k_scatter_tiled_batch<T, false, false, 0, true, false, 1>;
        }
        rc = h->ensure_lds(reinterpret_cast<const void*>(k1), h->k1_lds);
        if (rc) return rc;
        XM_LAUNCH(k1, dim3(std::min(grid_for(n_max, threads * TILE_EPT), 32u), n_frames), dim3(threads), h->k1_lds, stream, d_descs_redo,
                  h->tb, h->w_ts, h->w_x, 0);
      }
      launch_k2_batch<0, 1>(h, stream, d_descs_redo, n_frames);
      HIP_TRY(hipGetLastError());
      return XM_OK;
    }
  }
  // K0: grid = (blocks of the largest frame, frames)
  prof_slot(0);
  if (!sorted) {
    const bool vec2 = !AOS && std::is_same<T, long long>::value && vec16;
    const unsigned per_block = BLOCK * (vec2 ? 2 * K0_UN : 4);
    unsigned gx = grid_for(n_max, per_block);
    if (gx > 1024) gx = 1024;
    if constexpr (std::is_same<T, long long>::value && !AOS) {
      if (vec2) XM_LAUNCH((k_minmax_batch<T, false, HAS_P, 2>), dim3(gx, n_frames), dim3(BLOCK), 0, stream, d_descs);
      else XM_LAUNCH((k_minmax_batch<T, false, HAS_P, 1>), dim3(gx, n_frames), dim3(BLOCK), 0, stream, d_descs);
    } else {
      XM_LAUNCH((k_minmax_batch<T, AOS, HAS_P, 1>), dim3(gx, n_frames), dim3(BLOCK), 0, stream, d_descs);
    }
  }
  // K1: block size from the mean frame (see launch_scatter_tv); a sparser frame of the group only sends more of its events
  // down the direct path inside the kernel
  const double max_ev = h->tb.xmap_w > 0 ? (h->w_ts - 1.5) * (double)n_mean / (double)h->tb.xmap_w : 0.0;
  unsigned threads = TILE_THREADS;
  while (threads > 1024 / TILE_EPT && (double)(threads * TILE_EPT) > max_ev) threads >>= 1;
  const unsigned gx1 = grid_for(n_max, threads * TILE_EPT);
  constexpr bool kHasVec = !AOS && std::is_same<T, long long>::value;
  auto launch_k1 = [&](auto view_tag) -> int {
    constexpr int VIEW = decltype(view_tag)::value;
    if (direct_k1) {  // frames too sparse for the tiles: one thread per event, grid = (blocks of the largest frame, frames)
      prof_slot(1);
      XM_LAUNCH((k_scatter_direct_batch<T, AOS, HAS_P, VIEW>), dim3(std::max(1u, grid_for(n_max, BLOCK)), n_frames), dim3(BLOCK), 0,
                stream, d_descs, h->tb, sorted ? 1 : 0);
      return XM_OK;
    }
    auto kern = k_scatter_tiled_batch<T, AOS, HAS_P, VIEW, false>;
    if constexpr (kHasVec) {
      if (vec16) kern = k_scatter_tiled_batch<T, AOS, HAS_P, VIEW, true>;
    }
    if (key32) {
      kern = k_scatter_tiled_batch<T, AOS, HAS_P, VIEW, false, true>;
      if constexpr (kHasVec) {
        if (vec16) kern = k_scatter_tiled_batch<T, AOS, HAS_P, VIEW, true, true>;
      }
    }
    int rc = h->ensure_lds(reinterpret_cast<const void*>(kern), h->k1_lds);
    if (rc) return rc;
    prof_slot(1);
    XM_LAUNCH(kern, dim3(gx1, n_frames), dim3(threads), h->k1_lds, stream, d_descs, h->tb, h->w_ts, h->w_x, sorted ? 1 : 0);
    return XM_OK;
  };
  int rc = h->cfg.view == XM_VIEW_PROJECTOR ? launch_k1(std::integral_constant<int, 0>{}) : launch_k1(std::integral_constant<int, 1>{});
  if (rc) return rc;
  // K2
  prof_slot(2);
  if (h->cfg.view == XM_VIEW_PROJECTOR) {
    if (key32)
      launch_k2_batch<1>(h, stream, d_descs, n_frames);
    else
      launch_k2_batch<0>(h, stream, d_descs, n_frames);
  } else {
    const u64 px = (u64)h->tb.cam_w * h->tb.cam_h;
    if (key32) XM_LAUNCH(k_frame_cam32_batch, dim3(grid_for(px, BLOCK), n_frames), dim3(BLOCK), 0, stream, d_descs, px, h->tb.dlut);
    else XM_LAUNCH(k_frame_direct_batch, dim3(grid_for(px, BLOCK), n_frames), dim3(BLOCK), 0, stream, d_descs, px, h->tb.dlut);
  }
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

// can this group of frames go through the multi-frame kernels?  (dense enough for the tiled K1, tiled K2 available)
bool batch_path(const xm_handle* h, u64 n_mean) {
  const double max_ev = h->tb.xmap_w > 0 ? (h->w_ts - 1.5) * (double)n_mean / (double)h->tb.xmap_w : 0.0;
  return !h->k1_direct && h->w_ts > 0 && h->w_x > 0 && max_ev >= 1024.0 && !(h->cfg.view == XM_VIEW_PROJECTOR && h->k2_direct) &&
         !h->k2_flags;
}

// Enqueue one group: frame f = evs[f] on slot slot_idx[f], outputs depth[f] / bgr[f] (device pointers), everything on `stream`.
// d_descs / h_descs: where the group's descriptors live (the caller owns their lifetime).  `upload`: copy them now
// (eager) -- false when the caller uploads once (graph capture).
int enqueue_batch(xm_handle* h, const int* slot_idx, const EventsView* evs, float* const* depth, uint8_t* const* bgr,
                  int n_frames, hipStream_t stream, FrameDesc* h_descs, FrameDesc* d_descs, bool upload, bool allow_sorted,
                  hipEvent_t* prof = nullptr, int* kinds = nullptr, FrameDesc* h_descs_redo = nullptr,
                  FrameDesc* d_descs_redo = nullptr) {
  u64 n_max = 0, n_sum = 0;
  bool vec16 = true;
  for (int f = 0; f < n_frames; ++f) {
    const EventsView& ev = evs[f];
    n_max = std::max<u64>(n_max, ev.n);
    n_sum += ev.n;
    if (!ev.aos) vec16 = vec16 && aligned(ev.x, 16) && aligned(ev.y, 16) && aligned(ev.t, 16) && (!ev.use_p || aligned(ev.p, 16));
  }
  const u64 n_mean = n_frames ? n_sum / (u64)n_frames : 0;
  const EventsView& e0 = evs[0];
  bool sorted = allow_sorted && n_frames > 0;
  for (int f = 0; f < n_frames && sorted; ++f) sorted = evs[f].n > 0 && sorted_path(h, evs[f]);
  for (int f = 0; f < n_frames; ++f) {  // order the group after whatever its slots did last on other streams
    Slot& s = h->slots[slot_idx[f]];
    if (s.pending_batch_ev) {
      if (s.pending_batch_stream != stream && !h->capturing) HIP_TRY(hipStreamWaitEvent(stream, s.pending_batch_ev, 0));
      s.pending_batch_ev = nullptr;
    }
    if (s.eager_dirty && !h->capturing) {
      if (s.stream != stream) {
        HIP_TRY(hipEventRecord(h->join_ev[slot_idx[f]], s.stream));
        HIP_TRY(hipStreamWaitEvent(stream, h->join_ev[slot_idx[f]], 0));
      }
      s.eager_dirty = false;
    }
  }
  // frames too sparse for the tiled K1 (the reference's own recordings: ~150 k events over 1080 time columns): the multi-frame
  // K0 and K2 with the one-thread-per-event K1 in between -- three launches per group instead of three per frame
  const bool direct_k1 = !batch_path(h, n_mean) && !(h->cfg.view == XM_VIEW_PROJECTOR && (h->k2_direct || !h->d_k2_tiles[1])) &&
                         !h->k2_flags && n_frames >= 2 && n_max < (1ull << 31);
  if (!batch_path(h, n_mean) && !direct_k1) {  // untiled K2: frame by frame, still on the group's stream
    for (int f = 0; f < n_frames; ++f) {
      int rc = enqueue_frame(h, h->slots[slot_idx[f]], evs[f], depth[f], bgr[f], nullptr, allow_sorted, stream);
      if (rc) return rc;
      h->slots[slot_idx[f]].api_tag = h->slots[slot_idx[f]].host_tag;
    }
    return XM_OK;
  }
  int cols_w = sorted ? cols_width(h, n_mean) : 0;  // one tile width for the group (from its mean frame)
  for (int f = 0; f < n_frames && cols_w; ++f)
    if (!cols_path(h, evs[f], sorted) || (evs[f].aos != nullptr) != (e0.aos != nullptr)) cols_w = 0;
  // a batch that is being captured into a hipGraph: the column tiles with the redo decided on the device (launch_batch_t)
  // (groups of >= 2 frames: a lone frame's seven launches -- four of them returning at once -- take longer than K0 -> K1 -> K2)
  const bool dev_redo = h->capturing && !cols_w && !direct_k1 && d_descs_redo && h->cols_ok && !h->k2_direct && !h->k2_flags && n_frames >= 2;
  if (dev_redo) {
    cols_w = cols_width(h, n_mean);
    for (int f = 0; f < n_frames && cols_w; ++f)
      if (evs[f].n == 0 || evs[f].use_p || (!evs[f].aos && evs[f].t_dtype != XM_T_INT64) || !cols_width(h, evs[f].n) ||
          (evs[f].aos != nullptr) != (e0.aos != nullptr))
        cols_w = 0;
  }
  const bool redo_descs = dev_redo && cols_w;
  bool use32 = sorted && !cols_w && !direct_k1;
  for (int f = 0; f < n_frames && use32; ++f) use32 = key32_path(h, evs[f], sorted);
  {
    int v = h->key32_pause.load(std::memory_order_relaxed);
    while (v > 0 && !h->key32_pause.compare_exchange_weak(v, std::max(0, v - n_frames), std::memory_order_relaxed)) {
    }
  }
  for (int f = 0; f < n_frames; ++f) {
    Slot& s = h->slots[slot_idx[f]];
    if (s.host_tag >= KEY_MAX_TAG && !h->capturing) {
      int rc = reset_slot(h, s, stream);
      if (rc) return rc;
    }
    if (use32) {
      int rc = key32_prepare(h, s, s.host_tag + 1, stream);
      if (rc) return rc;
    }
    FrameDesc& d = h_descs[f];
    const EventsView& ev = evs[f];
    d.x = ev.x; d.y = ev.y; d.t = ev.t; d.p = ev.use_p ? ev.p : nullptr; d.aos = (const uint4*)ev.aos;
    d.n = ev.n; d.key_frame = cols_w ? reinterpret_cast<u64*>(s.frame16) : use32 ? reinterpret_cast<u64*>(s.key32) : s.key_frame;
    d.st = s.st; d.depth = depth[f];
    d.bgr = bgr[f]; d.valid = 1; d.pad = 0;
    if (redo_descs) {  // the same frame on the slot's 64-bit key frame
      h_descs_redo[f] = d;
      h_descs_redo[f].key_frame = s.key_frame;
    }
  }
  if (upload) HIP_TRY(hipMemcpyAsync(d_descs, h_descs, sizeof(FrameDesc) * n_frames, hipMemcpyHostToDevice, stream));
  int rc;
  if (e0.aos) rc = e0.use_p ? launch_batch_t<long long, true, true>(h, d_descs, n_frames, n_max, n_mean, false, sorted, stream, use32, 0, nullptr, nullptr, direct_k1)
                            : launch_batch_t<long long, true, false>(h, d_descs, n_frames, n_max, n_mean, false, sorted, stream, use32, cols_w, prof, redo_descs ? d_descs_redo : nullptr, direct_k1);
  else switch (e0.t_dtype) {
    case XM_T_INT64: rc = e0.use_p ? launch_batch_t<long long, false, true>(h, d_descs, n_frames, n_max, n_mean, vec16, sorted, stream, use32, 0, nullptr, nullptr, direct_k1)
                                   : launch_batch_t<long long, false, false>(h, d_descs, n_frames, n_max, n_mean, vec16, sorted, stream, use32, cols_w, prof, redo_descs ? d_descs_redo : nullptr, direct_k1); break;
    case XM_T_FLOAT32: rc = e0.use_p ? launch_batch_t<float, false, true>(h, d_descs, n_frames, n_max, n_mean, vec16, sorted, stream, use32, 0, nullptr, nullptr, direct_k1)
                                     : launch_batch_t<float, false, false>(h, d_descs, n_frames, n_max, n_mean, vec16, sorted, stream, use32, 0, nullptr, nullptr, direct_k1); break;
    default: rc = e0.use_p ? launch_batch_t<double, false, true>(h, d_descs, n_frames, n_max, n_mean, vec16, sorted, stream, use32, 0, nullptr, nullptr, direct_k1)
                           : launch_batch_t<double, false, false>(h, d_descs, n_frames, n_max, n_mean, vec16, sorted, stream, use32, 0, nullptr, nullptr, direct_k1);
  }
  if (rc) return rc;
  if (kinds) {  // which launches the group consisted of: {K0 general / K0b bounds / none, K1 variant}
    kinds[0] = cols_w ? 2 : sorted ? 0 : 1;  // (K0 runs whenever the frames are not on a sorted path)
    kinds[1] = cols_w ? KM_COLS : use32 ? KM_KEY32 : KM_KEY64;
  }
  for (int f = 0; f < n_frames; ++f) {
    Slot& s = h->slots[slot_idx[f]];
    s.host_tag += 1;
    s.api_tag = s.host_tag;
    s.any_frame = true;
    s.last_n = evs[f].n;
    s.last_sorted = sorted || cols_w != 0;
    s.last_key32 = use32 || cols_w;
    s.last_cols = cols_w != 0;
    h->path_counts[cols_w ? 3 : use32 ? 2 : sorted ? 1 : 0].fetch_add(1, std::memory_order_relaxed);
    s.last_t_dtype = evs[f].aos ? XM_T_INT64 : evs[f].t_dtype;
    if (use32 || cols_w) key32_note(h, false);
  }
  return XM_OK;
}

template <typename T>
void decode_minmax(const SlotState& hs, u32 parity, double& lo, double& hi, bool& any) {
  u64 a = MM_INIT_MIN, b = MM_INIT_MAX;
  for (int i = 0; i < MM_SLOTS; ++i) {
    a = hs.mm[parity][i][0] < a ? hs.mm[parity][i][0] : a;
    b = hs.mm[parity][i][1] > b ? hs.mm[parity][i][1] : b;
  }
  any = !(a == MM_INIT_MIN && b == MM_INIT_MAX);
  lo = any ? (double)TimeCodec<T>::dec(a) : 0.0;
  hi = any ? (double)TimeCodec<T>::dec(b) : 0.0;
}

// read the slot's state back and fill stats for its most recent frame (stream must be idle)
int fetch_stats(xm_handle* h, Slot& s, int t_dtype, xm_frame_stats* out) {
  SlotState hs;
  HIP_TRY(hipMemcpy(&hs, s.st, sizeof hs, hipMemcpyDeviceToHost));
  const u32 parity = s.host_tag & 1;
  memset(out, 0, sizeof *out);
  out->n_events = s.last_n;
  for (int i = 0; i < CNT_SLOTS; ++i) {
    out->n_used += hs.cnt[parity][i][CNT_USED];
    out->n_inliers += hs.cnt[parity][i][CNT_INLIER];
    out->n_index_errors += hs.cnt[parity][i][CNT_OOB];
    out->n_unsorted += hs.cnt[parity][i][CNT_UNSORTED];
  }
  if (s.last_sorted) out->n_used = s.last_n;  // no polarity column on the time-sorted path; K0 (which counts) did not run
  bool any;
  if (t_dtype == XM_T_FLOAT32) decode_minmax<float>(hs, parity, out->t_min, out->t_max, any);
  else if (t_dtype == XM_T_FLOAT64) decode_minmax<double>(hs, parity, out->t_min, out->t_max, any);
  else decode_minmax<long long>(hs, parity, out->t_min, out->t_max, any);
  (void)h;
  return XM_OK;
}

int stage_in(DevBuf& b, const void* host, size_t bytes, hipStream_t st) {
  int rc = b.reserve(bytes ? bytes : 16);
  if (rc) return rc;
  if (bytes) HIP_TRY(hipMemcpyAsync(b.p, host, bytes, hipMemcpyHostToDevice, st));
  return XM_OK;
}

Slot& pick_slot(xm_handle* h) {
  h->last_slot = h->next_slot;
  h->next_slot = (h->next_slot + 1) % (int)h->slots.size();
  return h->slots[h->last_slot];
}

// XM_FLAG_TRY_SORTED: did the (t[0], t[n-1]) shortcut hold for the slot's last asynchronous frame?  The kernels answer in
// pinned host memory (no API call when the frame has finished, which it has when a slot comes round again); a frame that
// failed is redone here on the general path, into the same output buffers, before anything else happens on the slot.
// ---- worker threads --------------------------------------------------------------------------------------------------
void worker_main(xm_handle* h, Worker* w) {
  (void)hipSetDevice(h->cfg.device);
  for (;;) {
    unsigned long long t = w->tail.load(std::memory_order_relaxed);
    if (t == w->head.load(std::memory_order_acquire)) {  // empty: spin a little, then sleep
      bool got = false;
      for (int i = 0; i < 20000 && !got; ++i) {
        __builtin_ia32_pause();
        got = t != w->head.load(std::memory_order_acquire);
      }
      if (!got) {
        std::unique_lock<std::mutex> lk(w->mu);
        w->sleeping.store(true, std::memory_order_seq_cst);
        w->cv.wait(lk, [&] { return t != w->head.load(std::memory_order_acquire); });
        w->sleeping.store(false, std::memory_order_relaxed);
      }
    }
    const Job j = w->ring[t % Worker::CAP];
    w->tail.store(t + 1, std::memory_order_release);
    if (j.kind == Job::STOP) {
      w->done.store(t + 1, std::memory_order_release);
      return;
    }
    const int rc = enqueue_frame(h, h->slots[j.slot], j.ev, j.depth, j.bgr, nullptr, j.allow_sorted);
    if (rc != XM_OK && w->error.load(std::memory_order_relaxed) == 0) {
      w->error_text = g_err;  // thread-local text of this worker
      w->error.store(rc, std::memory_order_release);
    }
    w->done.store(t + 1, std::memory_order_release);
  }
}

void post_job(Worker* w, const Job& j) {
  const unsigned long long hd = w->head.load(std::memory_order_relaxed);
  while (hd - w->tail.load(std::memory_order_acquire) >= Worker::CAP) __builtin_ia32_pause();  // ring full: back-pressure
  w->ring[hd % Worker::CAP] = j;
  w->head.store(hd + 1, std::memory_order_seq_cst);
  if (w->sleeping.load(std::memory_order_seq_cst)) {
    std::lock_guard<std::mutex> lk(w->mu);
    w->cv.notify_one();
  }
}

// wait until the workers have issued everything posted so far (the GPU may still be running it); reports a failed job
int drain_workers(xm_handle* h, int only = -1) {
  int rc = XM_OK;
  for (size_t i = 0; i < h->workers.size(); ++i) {
    if (only >= 0 && (int)i != only) continue;
    Worker* w = h->workers[i].get();
    const unsigned long long hd = w->head.load(std::memory_order_acquire);
    while (w->done.load(std::memory_order_acquire) < hd) __builtin_ia32_pause();
    const int e = w->error.load(std::memory_order_acquire);
    if (e && rc == XM_OK) {
      rc = fail(e, "%s (reported by the launch worker of stream %zu)", w->error_text.c_str(), i);
      w->error.store(0, std::memory_order_release);
    }
  }
  return rc;
}

// device set + launch workers idle: the entry of every call that uses the slots' streams itself
#define XM_ENTER(h)                          \
  do {                                       \
    HIP_TRY(hipSetDevice((h)->cfg.device));  \
    int rc_enter_ = drain_workers(h);        \
    if (rc_enter_) return rc_enter_;         \
    if (!(h)->pending.empty() && (rc_enter_ = flush_pending(h))) return rc_enter_;  \
  } while (0)

#ifndef XM_POLL_FIRST_US
#define XM_POLL_FIRST_US 30
#define XM_POLL_NEXT_US 100
#endif
int resolve_prev(xm_handle* h, Slot& s, bool* redone = nullptr) {
  if (!s.prev.valid) return XM_OK;
  s.prev.valid = false;
  const u32 tag = s.prev.tag;
  // launch workers: the frame may be posted but not launched yet -- an empty stream also answers hipSuccess to the query
  // below, which would read as "shortcut held".  Wait until the worker has issued everything posted so far.
  if (s.worker >= 0) {
    const int rcw = drain_workers(h, s.worker);
    if (rcw) return rcw;
  }
  // still in flight?  Wait for K2's start marker by polling the pinned word: a blocking stream synchronisation costs a
  // ~200 us wake-up, per frame, whenever the host runs ahead of the GPU (few slots); the marker is a few us away.
  if (__atomic_load_n(&s.h_flags[1], __ATOMIC_ACQUIRE) != tag) {
    auto t_next = std::chrono::steady_clock::now() + std::chrono::microseconds(XM_POLL_FIRST_US);
    unsigned spins = 0;
    while (__atomic_load_n(&s.h_flags[1], __ATOMIC_ACQUIRE) != tag) {
      __builtin_ia32_pause();
      if ((++spins & 0x3f) == 0 && std::chrono::steady_clock::now() > t_next) {
        // not there after 30 us: make sure the runtime has really handed the slot's commands to the GPU (a query flushes
        // anything it still holds back -- seen: a frame that sat for 20 ms until something synchronised), and stop polling
        // once the stream itself reports completion
        hipError_t q = hipStreamQuery(s.prev.stream ? s.prev.stream : s.stream);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) HIP_TRY(q);
        t_next = std::chrono::steady_clock::now() + std::chrono::microseconds(XM_POLL_NEXT_US);
      }
    }
  }
  if (!s.prev.check || __atomic_load_n(&s.h_flags[0], __ATOMIC_ACQUIRE) != tag) return XM_OK;
  h->sorted_fallbacks += 1;
  if (s.last_key32) key32_note(h, true);
  if (s.worker >= 0 && !s.prev.host_depth && !s.prev.host_bgr) {  // the redo goes the way the frame went
    Job j;
    j.slot = (int)(&s - h->slots.data());
    j.ev = s.prev.ev;
    j.depth = s.prev.depth;
    j.bgr = s.prev.bgr;
    j.allow_sorted = false;
    s.api_tag = s.api_tag >= KEY_MAX_TAG ? 1 : s.api_tag + 1;
    post_job(h->workers[s.worker].get(), j);
    if (redone) *redone = true;
    return XM_OK;
  }
  int rc = s.worker >= 0 ? drain_workers(h, s.worker) : XM_OK;
  if (rc) return rc;
  rc = enqueue_frame(h, s, s.prev.ev, s.prev.depth, s.prev.bgr, nullptr, false);
  if (rc) return rc;
  s.api_tag = s.host_tag;
  const size_t px = (size_t)h->out_w * h->out_h;
  if (s.prev.host_depth) HIP_TRY(hipMemcpyAsync(s.prev.host_depth, s.prev.depth, px * 4, hipMemcpyDeviceToHost, s.stream));
  if (s.prev.host_bgr) HIP_TRY(hipMemcpyAsync(s.prev.host_bgr, s.prev.bgr, px * 3, hipMemcpyDeviceToHost, s.stream));
  if (redone) *redone = true;
  return XM_OK;
}

int process_common(xm_handle* h, EventsView ev, int mem, float* depth_out, uint8_t* bgr_out, xm_frame_stats* stats,
                   bool profile) {
  if (!h) return fail(XM_ERR_INVALID, "NULL handle");
  HIP_TRY(hipSetDevice(h->cfg.device));
  int rc = check_events(ev);
  if (rc) return rc;
  const size_t px = (size_t)h->out_w * h->out_h;
  if (h->ab_max >= 2 && mem == XM_MEM_DEVICE && !profile && !h->capturing && !stats) {  // adaptive batching (see xm_handle::pending)
    h->pending.push_back(xm_handle::Deferred{ev, depth_out, bgr_out});
    int in_flight = 0;
    for (hipEvent_t e : h->ab_inflight)
      if (e && hipEventQuery(e) == hipErrorNotReady) in_flight += 1;
    (void)hipGetLastError();
    if ((int)h->pending.size() >= h->ab_max || in_flight < 3) return flush_pending(h);
    return XM_OK;
  }
  if (!h->pending.empty() && (rc = flush_pending(h))) return rc;  // (a synchronous / host-memory call behind deferred frames)
  Slot& s = profile ? h->slots[0] : pick_slot(h);
  if (profile) h->last_slot = 0;
  if ((rc = resolve_prev(h, s))) return rc;
  if (s.worker >= 0 && mem == XM_MEM_DEVICE && !profile && !h->capturing) {
    // asynchronous device-pointer frame: the launches are the worker's job
    Job j;
    j.slot = (int)(&s - h->slots.data());
    j.ev = ev;
    j.depth = depth_out;
    j.bgr = bgr_out;
    s.api_tag = s.api_tag >= KEY_MAX_TAG ? 1 : s.api_tag + 1;
    post_job(h->workers[s.worker].get(), j);
    if (s.h_flags) {
      s.prev.valid = true;
      s.prev.check = h->try_sorted && sorted_path(h, ev);
      s.prev.ev = ev;
      s.prev.depth = depth_out;
      s.prev.bgr = bgr_out;
      s.prev.host_depth = nullptr;
      s.prev.host_bgr = nullptr;
      s.prev.tag = s.api_tag;
      s.prev.stream = s.stream;
    }
    return XM_OK;
  }
  if (s.worker >= 0 && (rc = drain_workers(h, s.worker))) return rc;  // this call uses the slot's stream itself
  float* d_depth = depth_out;
  uint8_t* d_bgr = bgr_out;
  const bool host_in = mem == XM_MEM_HOST || mem == XM_MEM_HOST_PINNED;
  if (host_in) {
    const size_t n = ev.n;
    if (ev.aos) {
      if ((rc = stage_in(s.ev_aos, ev.aos, n * 16, s.stream))) return rc;
      ev.aos = s.ev_aos.p;
    } else {
      if ((rc = stage_in(s.ev_x, ev.x, n * 2, s.stream))) return rc;
      if ((rc = stage_in(s.ev_y, ev.y, n * 2, s.stream))) return rc;
      if ((rc = stage_in(s.ev_t, ev.t, n * t_size(ev.t_dtype), s.stream))) return rc;
      ev.x = (const uint16_t*)s.ev_x.p;
      ev.y = (const uint16_t*)s.ev_y.p;
      ev.t = s.ev_t.p;
      if (ev.p) {
        if ((rc = stage_in(s.ev_p, ev.p, n * 2, s.stream))) return rc;
        ev.p = (const int16_t*)s.ev_p.p;
      }
    }
    if (depth_out) {
      if ((rc = s.out_depth.reserve(px * 4))) return rc;
      d_depth = (float*)s.out_depth.p;
    }
    if (bgr_out) {
      if ((rc = s.out_bgr.reserve(px * 3))) return rc;
      d_bgr = (uint8_t*)s.out_bgr.p;
    }
  } else if (mem != XM_MEM_DEVICE) {
    return fail(XM_ERR_INVALID, "mem must be XM_MEM_HOST, XM_MEM_HOST_PINNED or XM_MEM_DEVICE");
  }
  if ((rc = enqueue_frame(h, s, ev, d_depth, d_bgr, profile ? h->prof_ev : nullptr))) return rc;
  s.api_tag = s.host_tag;
  if (host_in) {
    if (depth_out) HIP_TRY(hipMemcpyAsync(depth_out, d_depth, px * 4, hipMemcpyDeviceToHost, s.stream));
    if (bgr_out) HIP_TRY(hipMemcpyAsync(bgr_out, d_bgr, px * 3, hipMemcpyDeviceToHost, s.stream));
  }
  if (s.h_flags && !h->capturing && mem != XM_MEM_HOST && !profile) {
    // asynchronous call: the slot is not reused before this frame has reached K2 (keeps the host from running queues
    // deep ahead of the GPU, which made the frame rate uneven), and a try-sorted verdict is read then
    s.prev.valid = true;
    s.prev.check = h->try_sorted && s.last_sorted;
    s.prev.ev = ev;  // device pointers (the slot's staging buffers for pinned host input)
    s.prev.depth = d_depth;
    s.prev.bgr = d_bgr;
    s.prev.host_depth = host_in ? depth_out : nullptr;
    s.prev.host_bgr = host_in ? bgr_out : nullptr;
    s.prev.tag = s.host_tag;
    s.prev.stream = s.stream;
  }
  if (mem == XM_MEM_HOST || profile) {
    HIP_TRY(hipStreamSynchronize(s.stream));
    xm_frame_stats st;
    if ((rc = fetch_stats(h, s, ev.aos ? XM_T_INT64 : ev.t_dtype, &st))) return rc;
    if (profile) {
      const int first = s.last_sorted && !s.last_cols ? 1 : 0;  // K0 is not launched on the time-sorted path (column tiles: K0b in its place)
#ifdef XM_ABLATE  // experiment builds may skip kernels (XM_SKIP_MASK): their events were never recorded
      for (int i = first; i < 3; ++i)
        if (hipEventElapsedTime(&st.gpu_ms[i], h->prof_ev[2 * i], h->prof_ev[2 * i + 1]) != hipSuccess) (void)hipGetLastError();
      if (hipEventElapsedTime(&st.gpu_ms[3], h->prof_ev[2 * first], h->prof_ev[5]) != hipSuccess) (void)hipGetLastError();
#else
      for (int i = first; i < 3; ++i) HIP_TRY(hipEventElapsedTime(&st.gpu_ms[i], h->prof_ev[2 * i], h->prof_ev[2 * i + 1]));
      HIP_TRY(hipEventElapsedTime(&st.gpu_ms[3], h->prof_ev[2 * first], h->prof_ev[5]));  // start of first .. end of K2
#endif
    }
    if (st.n_unsorted && s.last_sorted) {
      // the time-sorted declaration did not hold for this frame: redo it on the general path (K0 -> K1 -> K2)
      h->sorted_fallbacks += 1;
      if (s.last_key32) key32_note(h, true);
      if ((rc = enqueue_frame(h, s, ev, d_depth, d_bgr, nullptr, false))) return rc;
      s.api_tag = s.host_tag;
      if (mem == XM_MEM_HOST) {
        if (depth_out) HIP_TRY(hipMemcpyAsync(depth_out, d_depth, px * 4, hipMemcpyDeviceToHost, s.stream));
        if (bgr_out) HIP_TRY(hipMemcpyAsync(bgr_out, d_bgr, px * 3, hipMemcpyDeviceToHost, s.stream));
      }
      HIP_TRY(hipStreamSynchronize(s.stream));
      const uint64_t flagged = st.n_unsorted;
      if ((rc = fetch_stats(h, s, ev.aos ? XM_T_INT64 : ev.t_dtype, &st))) return rc;
      st.n_unsorted = flagged;
      HIP_TRY(hipMemsetAsync(&s.st->unsorted_sticky, 0, sizeof(u32), s.stream));  // handled here, not an xm_sync error
      HIP_TRY(hipStreamSynchronize(s.stream));
    }
    if (stats) *stats = st;
    if (st.n_index_errors)
      return fail(XM_ERR_INDEX, "%llu event(s) indexed outside a table/frame (IndexError in the reference)",
                  (unsigned long long)st.n_index_errors);
  }
  return XM_OK;
}

int ensure_stage_frame(xm_handle* h) {
  const size_t need = std::max((size_t)h->tb.rect_w * h->tb.rect_h, (size_t)h->tb.cam_w * h->tb.cam_h);
  if (h->stage_frame && h->stage_cells >= need) return XM_OK;
  if (h->stage_frame) (void)hipFree(h->stage_frame);
  h->stage_frame = nullptr;
  HIP_TRY(hipMalloc((void**)&h->stage_frame, need * sizeof(u64)));
  h->stage_cells = need;
  return XM_OK;
}

int rearm_aux(xm_handle* h, hipStream_t stream, u64* frame, u64 cells) {
  hipLaunchKernelGGL(k_reset_slot, dim3(cells ? 1024 : 1), dim3(BLOCK), 0, stream, h->aux_st, frame, cells,
                     (unsigned char*)nullptr);
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

template <typename T>
void host_minmax_out(const SlotState& hs, void* out) {
  u64 a = MM_INIT_MIN, b = MM_INIT_MAX;
  for (int i = 0; i < MM_SLOTS; ++i) {
    a = hs.mm[0][i][0] < a ? hs.mm[0][i][0] : a;
    b = hs.mm[0][i][1] > b ? hs.mm[0][i][1] : b;
  }
  T* o = (T*)out;
  if (a == MM_INIT_MIN && b == MM_INIT_MAX) {  // empty shard: neutral elements of min / max
    o[0] = std::numeric_limits<T>::has_infinity ? std::numeric_limits<T>::infinity() : std::numeric_limits<T>::max();
    o[1] = std::numeric_limits<T>::has_infinity ? -std::numeric_limits<T>::infinity() : std::numeric_limits<T>::lowest();
    return;
  }
  o[0] = TimeCodec<T>::dec(a);
  o[1] = TimeCodec<T>::dec(b);
}

}  // namespace

// =====================================================================================================
extern "C" {

int xm_api_version(void) { return XM_API_VERSION; }
const char* xm_last_error(void) { return g_err.c_str(); }

int xm_create(const xm_config* cfg, xm_handle** out) {
  if (!cfg || !out) return fail(XM_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(xm_config))
    return fail(XM_ERR_INVALID, "xm_config.struct_size %u != %zu", cfg->struct_size, sizeof(xm_config));
  if (cfg->cam_width <= 0 || cfg->cam_height <= 0 || cfg->rect_width <= 0 || cfg->rect_height <= 0 ||
      cfg->xmap_width <= 1)
    return fail(XM_ERR_INVALID, "bad dimensions");
  if (cfg->cam_width > 32767 || cfg->cam_height > 32767 || cfg->rect_width > 32767 || cfg->rect_height > 32767 ||
      cfg->xmap_width > 32767 || cfg->proj_width > 32767 || cfg->proj_height > 32767)
    return fail(XM_ERR_INVALID, "dimensions must fit int16 indices (x_maps_disparity.py:52-53)");
  if (cfg->x_offset < 0 || cfg->x_offset > 32767) return fail(XM_ERR_INVALID, "x_offset must fit int16");
  if (cfg->view != XM_VIEW_PROJECTOR && cfg->view != XM_VIEW_CAMERA) return fail(XM_ERR_INVALID, "bad view");
  if (!cfg->cam_mapx_i16 || !cfg->cam_mapy_i16 || !cfg->proj_x_map) return fail(XM_ERR_INVALID, "NULL table");
  if (cfg->view == XM_VIEW_PROJECTOR && (!cfg->disp_proj_mapxy_i16 || cfg->proj_width <= 0 || cfg->proj_height <= 0))
    return fail(XM_ERR_INVALID, "projector view needs disp_proj_mapxy_i16 and the projector size");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(XM_ERR_HIP, "no HIP device visible: the X-maps hot path needs an AMD GPU (no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(XM_ERR_INVALID, "device %d out of range (%d)", cfg->device, ndev);
  HIP_TRY(hipSetDevice(cfg->device));

  xm_handle* h = new (std::nothrow) xm_handle();
  if (!h) return fail(XM_ERR_NOMEM, "out of host memory");
  h->cfg = *cfg;
  const int n_slots = cfg->n_slots > 0 ? cfg->n_slots : 1;
  const int xmap_h = cfg->xmap_height > 0 ? cfg->xmap_height : cfg->rect_height;
  h->cfg.n_slots = n_slots;
  h->time_sorted = (cfg->flags & XM_FLAG_TIME_SORTED) != 0;
  // default: the verified (t[0], t[n-1]) shortcut with automatic redo (exact for any event order); XM_FLAG_GENERAL forces the
  // extrema pass on every frame; XM_GENERAL=1 in the environment does the same (experiments)
  const char* eg = getenv("XM_GENERAL");
  h->try_sorted = !h->time_sorted && !(cfg->flags & XM_FLAG_GENERAL) && !(eg && eg[0] == '1');
  if (const char* e = getenv("XM_GATE_SLOTS")) h->gate_slots = e[0] != '0';
  h->cfg.xmap_height = xmap_h;
  if ((cfg->flags & XM_FLAG_ADAPTIVE_BATCH) && n_slots >= 8) h->ab_max = std::min(n_slots / 4, 32);  // (four groups' worth of slots)

#define XM_TRY_CREATE(expr)                   \
  do {                                        \
    hipError_t e_ = (expr);                   \
    if (e_ != hipSuccess) {                   \
      int rc_ = fail(XM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
      xm_destroy(h);                          \
      return rc_;                             \
    }                                         \
  } while (0)

  // re-pack the int16 tables: one 4-byte gather per event instead of two 2-byte ones, and the scan axis made the
  // slow axis (column-major) so that a time slice of events touches a few contiguous runs (see DevTables)
  const size_t cam_px = (size_t)cfg->cam_width * cfg->cam_height;
  {
    std::vector<u32> lut(cam_px);
    for (int y = 0; y < cfg->cam_height; ++y)
      for (int x = 0; x < cfg->cam_width; ++x) {
        const size_t i = (size_t)y * cfg->cam_width + x;
        lut[(size_t)x * cfg->cam_height + y] =
            ((u32)(uint16_t)cfg->cam_mapy_i16[i] << 16) | (u32)(uint16_t)cfg->cam_mapx_i16[i];
      }
    XM_TRY_CREATE(hipMalloc((void**)&h->d_lut, cam_px * 4 + 64));  // +slack: bands are read in aligned 16-B vectors
    XM_TRY_CREATE(hipMemcpy(h->d_lut, lut.data(), cam_px * 4, hipMemcpyHostToDevice));
  }
  const size_t xm_cells = (size_t)xmap_h * cfg->xmap_width;
  {
    std::vector<int16_t> xt(xm_cells);
    for (int r = 0; r < xmap_h; ++r)
      for (int c = 0; c < cfg->xmap_width; ++c) xt[(size_t)c * xmap_h + r] = cfg->proj_x_map[(size_t)r * cfg->xmap_width + c];
    XM_TRY_CREATE(hipMalloc((void**)&h->d_xmap, xm_cells * 2 + 64));
    XM_TRY_CREATE(hipMemcpy(h->d_xmap, xt.data(), xm_cells * 2, hipMemcpyHostToDevice));
  }
  if (cfg->disp_proj_mapxy_i16 && cfg->proj_width > 0 && cfg->proj_height > 0) {
    const size_t ppx = (size_t)cfg->proj_width * cfg->proj_height;
    std::vector<u32> pm(ppx);
    for (size_t i = 0; i < ppx; ++i)
      pm[i] = ((u32)(uint16_t)cfg->disp_proj_mapxy_i16[2 * i + 1] << 16) | (u32)(uint16_t)cfg->disp_proj_mapxy_i16[2 * i];
    XM_TRY_CREATE(hipMalloc((void**)&h->d_pmap, ppx * 4));
    XM_TRY_CREATE(hipMemcpy(h->d_pmap, pm.data(), ppx * 4, hipMemcpyHostToDevice));
  }
  XM_TRY_CREATE(hipMalloc((void**)&h->d_zero16, 256));
  XM_TRY_CREATE(hipMemset(h->d_zero16, 0, 256));
  XM_TRY_CREATE(hipMalloc((void**)&h->d_dlut, 65536 * sizeof(uint2)));
  hipLaunchKernelGGL(k_build_dlut, dim3(65536 / BLOCK), dim3(BLOCK), 0, 0, h->d_dlut, cfg->p03, cfg->z_near, cfg->z_far);
  XM_TRY_CREATE(hipGetLastError());
  XM_TRY_CREATE(hipDeviceSynchronize());
  h->tb.dlut = h->d_dlut;
  h->tb.lut = h->d_lut;
  h->tb.xmap = h->d_xmap;
  h->tb.pmap = h->d_pmap;
  h->tb.cam_w = cfg->cam_width;
  h->tb.cam_h = cfg->cam_height;
  h->tb.proj_w = cfg->proj_width;
  h->tb.proj_h = cfg->proj_height;
  h->tb.rect_w = cfg->rect_width;
  h->tb.rect_h = cfg->rect_height;
  h->tb.xmap_w = cfg->xmap_width;
  h->tb.xmap_h = xmap_h;
  h->tb.x_offset = cfg->x_offset;
  h->tb.t_px_scale = cfg->xmap_width - 1;
  h->tb.p03 = cfg->p03;
  h->tb.z_near = cfg->z_near;
  h->tb.z_far = cfg->z_far;
  if (h->d_pmap) {  // K2's static per-tile patch rectangles and per-pixel offsets, for both of its geometries
    if (const char* e = getenv("XM_K2_PPT")) h->k2_force_ppt = atoi(e);
    double mean_cells2 = 0.0;
    for (int g = 0; g < 3; ++g) {
      const int ppt = 1 << g;
      const unsigned tiles_x = grid_for(cfg->proj_width, K2_TX * ppt), tiles_y = grid_for(cfg->proj_height, K2_TY);
      XM_TRY_CREATE(hipMalloc((void**)&h->d_k2_tiles[g], (size_t)tiles_x * tiles_y * sizeof(int4)));
      XM_TRY_CREATE(hipMalloc((void**)&h->d_k2_pix[g], (size_t)cfg->proj_width * cfg->proj_height * sizeof(u32)));
      if (g == 0) hipLaunchKernelGGL(k_build_k2_tables<1>, dim3(tiles_x, tiles_y), dim3(K2_TX * K2_TY), 0, 0, h->tb, h->d_k2_tiles[g], h->d_k2_pix[g]);
      else if (g == 1) hipLaunchKernelGGL(k_build_k2_tables<2>, dim3(tiles_x, tiles_y), dim3(K2_TX * K2_TY), 0, 0, h->tb, h->d_k2_tiles[g], h->d_k2_pix[g]);
      else hipLaunchKernelGGL(k_build_k2_tables<4>, dim3(tiles_x, tiles_y), dim3(K2_TX * K2_TY), 0, 0, h->tb, h->d_k2_tiles[g], h->d_k2_pix[g]);
      XM_TRY_CREATE(hipGetLastError());
      XM_TRY_CREATE(hipDeviceSynchronize());
      // largest LDS patch any tile of this rig needs -> K2's dynamic LDS
      std::vector<int4> tiles((size_t)tiles_x * tiles_y);
      XM_TRY_CREATE(hipMemcpy(tiles.data(), h->d_k2_tiles[g], tiles.size() * sizeof(int4), hipMemcpyDeviceToHost));
      int cap = 8;
      bool pipe_ok = (cfg->rect_height & 7) == 0;
      double cells = 0.0;
      for (const int4& r : tiles) {
        if (r.z > 0) cap = std::max(cap, r.z * r.w);
        if (r.z > 0) cells += (double)r.z * r.w;
        pipe_ok = pipe_ok && k2_pipe_tile_ok(r);
      }
      if (g < 2)
        for (const int4& r : tiles) h->k2_patch_cols_max = r.z < 0 || h->k2_patch_cols_max < 0 ? -1 : std::max(h->k2_patch_cols_max, r.z);
      if (g == 1) {
        h->k2_pipe_rig_ok = pipe_ok;
        mean_cells2 = cells / (double)std::max<size_t>(tiles.size(), 1);
      }
      // Four pixels per thread when the 32 x 16-pixel tiles' patches are small against the tile (a projector image finer than
      // the rectified frame: < 2 patch cells per pixel): an item's fixed costs -- five barriers, the descriptor reads, the tile
      // arithmetic -- then weigh more than its patch, and half as many items carry the same pixels
      if (g == 2) {
        const char* e4 = getenv("XM_K2_PIPE_PPT");  // experiments: 2 / 4
        h->k2_pipe4 = pipe_ok && h->k2_pipe_rig_ok && (e4 ? atoi(e4) == 4 : mean_cells2 < 2.0 * (2 * K2_TX * K2_TY));
      }
      h->k2_tile_cap[g] = std::min((cap + 7) & ~7, (int)K2_TILE_MAX);
    }
    h->tb.k2_tiles1 = h->d_k2_tiles[0];
    h->tb.k2_pix1 = h->d_k2_pix[0];
    h->tb.k2_tiles = h->d_k2_tiles[1];
    h->tb.k2_pix = h->d_k2_pix[1];
  }
  {  // does the rig qualify for the compact (32-bit) key frame?  (see key32_tag in xmaps_kernels.hpp)
    int xr_min = 32767, xp_max = 0;
    for (size_t i = 0; i < cam_px; ++i) xr_min = std::min<int>(xr_min, cfg->cam_mapx_i16[i]);
    for (size_t i = 0; i < xm_cells; ++i) xp_max = std::max<int>(xp_max, cfg->proj_x_map[i]);
    const long max_disp = std::max<long>((long)xp_max - xr_min - cfg->x_offset, (long)0 - xr_min - cfg->x_offset);
    const char* e32 = getenv("XM_KEY32");
    // (camera view: (event index + 1) << 12 | disparity on the camera frame -- only the disparity range matters)
    h->key32_ok = (cfg->view != XM_VIEW_PROJECTOR || (cfg->rect_height & 3) == 0) && max_disp < (1l << KEY32_DISP_BITS) &&
                  !(e32 && e32[0] == '0');
    // the pipelined K2 keeps the per-disparity table in LDS: every disparity an event of this rig can have (<= 4096 entries, 32 KB)
    h->k2_pipe_nlds = max_disp + 1 <= 4096 ? (int)std::max<long>(1, max_disp + 1) : 0;
    if (const char* e = getenv("XM_K2_PIPE")) h->k2_pipe = e[0] != '0';
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0) h->n_cus = prop.multiProcessorCount;
  }
  {  // does the rig qualify for the column-tile K1?  (xmaps_k1cols.hpp)
    int xr_min = 32767, xr_max = -32768, xp_min = 32767, xp_max = -32768;
    for (size_t i = 0; i < cam_px; ++i) {
      xr_min = std::min<int>(xr_min, cfg->cam_mapx_i16[i]);
      xr_max = std::max<int>(xr_max, cfg->cam_mapx_i16[i]);
    }
    for (size_t i = 0; i < xm_cells; ++i) {
      xp_min = std::min<int>(xp_min, cfg->proj_x_map[i]);
      xp_max = std::max<int>(xp_max, cfg->proj_x_map[i]);
    }
    const char* ec = getenv("XM_COLS");
    // the reference's int16 wrap-around in disp = xp - xr - x_offset (xmd:27) must never trigger on this rig: then
    // disp >= 0 <=> xp - x_offset >= xr, which is what makes "dead" X-map cells recognisable
    const bool no_wrap = (long)xp_max - xr_min - cfg->x_offset <= 32767 && (long)xp_min - xr_max - cfg->x_offset >= -32768;
    h->cols_xr_min = xr_min;
    bool injective = false;
    if (cfg->view == XM_VIEW_PROJECTOR && no_wrap && cfg->rect_width <= 65536) {
      u32* d_dup = nullptr;
      XM_TRY_CREATE(hipMalloc((void**)&d_dup, 2 * sizeof(u32)));
      XM_TRY_CREATE(hipMemset(d_dup, 0, 2 * sizeof(u32)));
      const int rows = std::min(xmap_h - 1, cfg->rect_height);
      if (rows > 0) hipLaunchKernelGGL(k_cols_check, dim3(rows), dim3(BLOCK), 0, 0, h->tb, xr_min, d_dup);
      u32 dup[2] = {1, 1};
      const hipError_t e1 = hipGetLastError(), e2 = hipMemcpy(dup, d_dup, sizeof dup, hipMemcpyDeviceToHost);
      (void)hipFree(d_dup);
      XM_TRY_CREATE(e1);
      XM_TRY_CREATE(e2);
      injective = dup[0] == 0;  // every frame cell has at most one (row, time column) that can write it
      // every live pair has its cell inside the frame and every row an event can land in was looked at: no per-event cell test
      if (dup[1] == 0 && cfg->rect_height >= xmap_h - 1) h->cols_flags |= COLS_F_ALL_IN_FRAME;
    }
    h->cols_ok = injective && h->d_pmap && !(ec && ec[0] == '0');
    h->cols_single = ec && ec[0] == '2';
    if (!injective && cfg->view == XM_VIEW_PROJECTOR && no_wrap && h->d_pmap && !(ec && ec[0] == '0')) {
      // the reference's own calibration: several time columns per frame cell -> owner tiles (xmaps_k1own.hpp)
      h->cols_flags = 0;
      const int rc_own = own_setup(h, cfg, xr_min);
      if (rc_own) {
        xm_destroy(h);
        return rc_own;
      }
      h->cols_ok = h->own_mode;
      // single-frame calls take the owner tiles too unless XM_COLS=1 says groups only (measured on ESL-like frames, four frames in
      // flight: 12.05 us per frame against 13.6 with the one-thread-per-event kernel and its 150 k divergent atomics)
      if (h->own_mode && !(ec && ec[0] == '1')) h->cols_single = true;
    }
    // the compact key frame orders the writers of a cell by TILE only: two time columns of one tile that share a cell would be
    // ordered by their disparity bits -- it needs the same property (the 64-bit keys carry the full event index and do not)
    if (cfg->view == XM_VIEW_PROJECTOR) h->key32_ok = h->key32_ok && injective;
    if (const char* e = getenv("XM_COLS_TARGET")) h->cols_target = std::max(256, atoi(e));
  }
  if (cfg->view == XM_VIEW_PROJECTOR) {
    h->key_cells = (size_t)cfg->rect_width * cfg->rect_height;
    h->out_w = cfg->proj_width;
    h->out_h = cfg->proj_height;
  } else {
    h->key_cells = cam_px;
    h->out_w = cfg->cam_width;
    h->out_h = cfg->cam_height;
  }

  {  // K1 LDS windows (w_ts X-map columns, w_x camera columns) within the LDS budget
    const char* e1 = getenv("XM_K1_DIRECT");
    const char* e2 = getenv("XM_K2_DIRECT");
    h->k1_direct = e1 && e1[0] == '1';
    h->k2_direct = e2 && e2[0] == '1';
    if (const char* e3 = getenv("XM_K2_FLAGS")) h->k2_flags = e3[0] == '1';
    // C-1M needs 44 KB (w_ts = 5, w_x = 16): three blocks per CU beside K2's 12 KB blocks
    size_t budget = 76 * 1024;
    if (const char* e = getenv("XM_LDS_KB")) budget = (size_t)atoi(e) * 1024;
    int w_ts = 5, w_x = 16;  // 5 time columns, 16 camera columns: 70 KB at C-1M
    if (const char* e = getenv("XM_W_TS")) w_ts = atoi(e);
    if (w_ts > 64) w_ts = 64;
    if (const char* e = getenv("XM_W_X")) w_x = atoi(e);
    auto need = [&](int wt, int wx) {
      // must mirror the carve-up at the top of k_scatter_tiled (uint4 units, +1 uint4 of alignment slack per band)
      const size_t win_words = cfg->view == XM_VIEW_PROJECTOR ? (size_t)wt * xmap_h : (size_t)wx * cfg->cam_height;
#ifndef XM_NO_LDS_DMA
      constexpr size_t slack = 64;  // LDS-direct band loads write whole waves: one wave of slack behind each band
#else
      constexpr size_t slack = 0;
#endif
      const size_t win_q = (win_words + 3) / 4, lut_q = ((size_t)wx * cfg->cam_height + 3) / 4 + 1 + slack,
                   xm_q = ((size_t)wt * xmap_h + 7) / 8 + 1 + slack;
      return 16 * (std::max(win_q, lut_q) + xm_q + 1 + slack);  // slots and LUT band share a region; +1 (+ a wave): dump area of the band loads
    };
    while (need(w_ts, w_x) > budget && (w_ts > 1 || w_x > 1)) {
      if (w_ts * xmap_h * 6 >= w_x * cfg->cam_height * 4 && w_ts > 1) w_ts -= 1;
      else if (w_x > 1) w_x /= 2;
      else w_ts -= 1;
    }
    if (need(w_ts, w_x) <= budget && w_ts >= 1 && w_x >= 1) {
      h->w_ts = w_ts;
      h->w_x = w_x;
      h->k1_lds = need(w_ts, w_x);
      if (const char* e = getenv("XM_K1_LDS_PAD_KB")) h->k1_lds += (size_t)atoi(e) * 1024;  // experiments: fewer blocks per CU
    } else {
      h->k1_direct = true;  // tables too tall for LDS: every event takes the direct path
    }
    // column tiles: the widest tile whose bands + slots fit the same budget (the LUT band is the tiled kernel's)
    if (h->cols_ok && !h->k1_direct && h->w_x > 0) {
      int wm = 0;
      while (wm < 16 && cols_lds_bytes(h, wm + 1) <= budget) wm += 1;
      h->cols_w_max = wm;
    }
    if (h->cols_w_max < 1 && !h->own_mode) h->cols_ok = false;
  }
#ifdef XM_ABLATE
  if (const char* e = getenv("XM_ABLATE")) {
    int v = atoi(e);
    XM_TRY_CREATE(hipMemcpyToSymbol(HIP_SYMBOL(xm::g_ablate), &v, sizeof v));
  }
#endif
  XM_TRY_CREATE(hipMalloc((void**)&h->d_states, sizeof(SlotState) * (n_slots + 1)));
  XM_TRY_CREATE(hipMemset(h->d_states, 0, sizeof(SlotState) * (n_slots + 1)));  // host_flags = NULL
  h->aux_st = h->d_states + n_slots;
  h->slots.resize(n_slots);
  for (int i = 0; i < n_slots; ++i) {
    Slot& s = h->slots[i];
    {
      // The slots' streams get their own hardware queues: HIP multiplexes all streams of one priority onto
      // GPU_MAX_HW_QUEUES (4) hardware queues, the application's default stream included, and how the eight slot streams
      // happened to interleave with it cost up to 17 % of the pipelined frame rate (first engine of a process: 64 Gev/s,
      // second: 75; tools/engine_order_probe.py).  Streams of another priority live in another queue pool.
      static const char* pe = getenv("XM_STREAM_PRIORITY");  // experiments: high (default) / low / normal
      int lo = 0, hi = 0;
      XM_TRY_CREATE(hipDeviceGetStreamPriorityRange(&lo, &hi));  // lo = least, hi = greatest priority (numerically lowest)
      // one stream per hardware queue; slots beyond that share them (more streams than queues is where the runtime's
      // stream -> queue assignment starts to matter, and it only added buffering, no overlap)
      static const int hw_q = getenv("GPU_MAX_HW_QUEUES") && atoi(getenv("GPU_MAX_HW_QUEUES")) > 0 ? atoi(getenv("GPU_MAX_HW_QUEUES")) : 4;
      static const int n_streams = getenv("XM_N_STREAMS") ? atoi(getenv("XM_N_STREAMS")) : hw_q;  // XM_N_STREAMS: experiments
      if (n_streams > 0 && i >= n_streams) {
        s.stream = h->slots[i % n_streams].stream;
        s.owns_stream = false;
      } else if ((pe && pe[0] == 'n') || (cfg->flags & XM_FLAG_DEFAULT_STREAMS))
        XM_TRY_CREATE(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
      else XM_TRY_CREATE(hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, pe && pe[0] == 'l' ? lo : hi));
    }
    XM_TRY_CREATE(hipMalloc((void**)&s.key_frame, h->key_cells * sizeof(u64)));
    if (h->key32_ok) {
      XM_TRY_CREATE(hipMalloc((void**)&s.key32, h->key_cells * sizeof(u32)));
      XM_TRY_CREATE(hipMemset(s.key32, 0, h->key_cells * sizeof(u32)));
    }
    if (h->cols_ok) {  // cells no (row, column) pair maps to are never written: they stay 0 from here on
      const size_t bytes = cols_frame_bytes(frame16_cells(h->tb), cfg->xmap_width);  // frame + K0b's bounds and thresholds
      XM_TRY_CREATE(hipMalloc((void**)&s.frame16, bytes));
      XM_TRY_CREATE(hipMemset(s.frame16, 0, bytes));
    }
    if (cfg->view == XM_VIEW_PROJECTOR && h->k2_flags)
      XM_TRY_CREATE(hipMalloc((void**)&s.dirty, ((h->key_cells + 15) >> 4) + 64));
    s.st = h->d_states + i;
    if (h->try_sorted || h->gate_slots) {
      XM_TRY_CREATE(hipHostMalloc((void**)&s.h_flags, 64, hipHostMallocMapped));
      s.h_flags[0] = s.h_flags[1] = 0;
      u32* d_flags = nullptr;
      XM_TRY_CREATE(hipHostGetDevicePointer((void**)&d_flags, s.h_flags, 0));
      XM_TRY_CREATE(hipMemcpy(&s.st->host_flags, &d_flags, sizeof d_flags, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_reset_slot, dim3(1024), dim3(BLOCK), 0, s.stream, s.st, s.key_frame, (u64)h->key_cells, s.dirty);
    XM_TRY_CREATE(hipGetLastError());
#ifdef XM_BLOG
    {
      const u32 id = (u32)i;  // experiments only: the block log is indexed by slot
      XM_TRY_CREATE(hipMemcpyAsync(&s.st->pad[0], &id, sizeof id, hipMemcpyHostToDevice, s.stream));
      XM_TRY_CREATE(hipStreamSynchronize(s.stream));
    }
#endif
  }
  hipLaunchKernelGGL(k_reset_slot, dim3(1), dim3(BLOCK), 0, h->slots[0].stream, h->aux_st, (u64*)nullptr, (u64)0,
                     (unsigned char*)nullptr);
  XM_TRY_CREATE(hipGetLastError());
  for (int i = 0; i < 6; ++i) XM_TRY_CREATE(hipEventCreate(&h->prof_ev[i]));
  XM_TRY_CREATE(hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming));
  h->join_ev.resize(n_slots, nullptr);
  for (int i = 0; i < n_slots; ++i) XM_TRY_CREATE(hipEventCreateWithFlags(&h->join_ev[i], hipEventDisableTiming));
  for (int i = 0; i < n_slots; ++i) XM_TRY_CREATE(hipStreamSynchronize(h->slots[i].stream));
  {  // multi-frame launches: distinct slot streams, their end-of-batch events, the descriptor ring
    for (int i = 0; i < n_slots; ++i) {
      bool seen = false;
      for (hipStream_t st : h->streams) seen = seen || st == h->slots[i].stream;
      if (!seen) h->streams.push_back(h->slots[i].stream);
    }
    h->batch_ev.resize(h->streams.size());
    h->batch_ev_next.assign(h->streams.size(), 0);
    for (auto& ring : h->batch_ev) {
      ring.assign(8, nullptr);
      for (auto& e : ring) XM_TRY_CREATE(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    XM_TRY_CREATE(hipHostMalloc((void**)&h->h_descs, sizeof(FrameDesc) * xm_handle::DESC_RING * n_slots, hipHostMallocDefault));
    XM_TRY_CREATE(hipMalloc((void**)&h->d_descs, sizeof(FrameDesc) * xm_handle::DESC_RING * n_slots));
    for (auto& e : h->desc_ev) XM_TRY_CREATE(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : h->graph_ev) XM_TRY_CREATE(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  {  // launch workers: one per distinct slot stream (XM_FLAG_LAUNCH_WORKERS; off: launches stay in the calling thread)
    const char* we = getenv("XM_WORKERS");  // overrides the flag either way
    const bool want = we ? we[0] != '0' : (cfg->flags & XM_FLAG_LAUNCH_WORKERS) != 0;
    if (want) {
      std::vector<hipStream_t> seen;
      for (int i = 0; i < n_slots; ++i) {
        Slot& s = h->slots[i];
        int w = -1;
        for (size_t k = 0; k < seen.size(); ++k)
          if (seen[k] == s.stream) w = (int)k;
        if (w < 0) {
          w = (int)seen.size();
          seen.push_back(s.stream);
          h->workers.emplace_back(new Worker());
        }
        s.worker = w;
      }
      for (auto& w : h->workers) w->th = std::thread(worker_main, h, w.get());
    }
  }
#undef XM_TRY_CREATE
  *out = h;
  return XM_OK;
}

void xm_destroy(xm_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->cfg.device);
  for (auto& w : h->workers) {
    Job stop;
    stop.kind = Job::STOP;
    post_job(w.get(), stop);
  }
  for (auto& w : h->workers)
    if (w->th.joinable()) w->th.join();
  h->workers.clear();
  for (auto gs : h->gstreams) if (gs) (void)hipStreamSynchronize(gs);
  for (Slot& s : h->slots) {
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    s.ev_x.release(); s.ev_y.release(); s.ev_t.release(); s.ev_p.release(); s.ev_aos.release();
    s.out_depth.release(); s.out_bgr.release();
    for (auto& d : s.dbg) d.release();
    if (s.key_frame) (void)hipFree(s.key_frame);
    if (s.key32) (void)hipFree(s.key32);
    if (s.frame16) (void)hipFree(s.frame16);
    if (s.dirty) (void)hipFree(s.dirty);
    if (s.stream && s.owns_stream) (void)hipStreamDestroy(s.stream);
    if (s.h_flags) (void)hipHostFree(s.h_flags);
  }
  for (auto& ring : h->batch_ev)
    for (auto& e : ring) if (e) (void)hipEventDestroy(e);
  for (auto& e : h->desc_ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : h->graph_ev) if (e) (void)hipEventDestroy(e);
  if (h->h_descs) (void)hipHostFree(h->h_descs);
  if (h->d_descs) (void)hipFree(h->d_descs);
  for (auto gs : h->gstreams) if (gs) (void)hipStreamDestroy(gs);
  for (auto& e : h->prof_ev) if (e) (void)hipEventDestroy(e);
  if (h->fork_ev) (void)hipEventDestroy(h->fork_ev);
  for (auto& e : h->join_ev) if (e) (void)hipEventDestroy(e);
  if (h->stage_frame) (void)hipFree(h->stage_frame);
  if (h->d_states) (void)hipFree(h->d_states);
  if (h->d_lut) (void)hipFree(h->d_lut);
  if (h->d_xmap) (void)hipFree(h->d_xmap);
  if (h->d_xmap_own) (void)hipFree(h->d_xmap_own);
  if (h->d_own_tiles) (void)hipFree(h->d_own_tiles);
  if (h->d_xmap_extra) (void)hipFree(h->d_xmap_extra);
  if (h->d_own_base) (void)hipFree(h->d_own_base);
  if (h->d_own_extra_cells) (void)hipFree(h->d_own_extra_cells);
  if (h->d_own_masks) (void)hipFree(h->d_own_masks);
  if (h->d_pmap) (void)hipFree(h->d_pmap);
  if (h->d_dlut) (void)hipFree(h->d_dlut);
  for (int g = 0; g < 3; ++g) {
    if (h->d_k2_tiles[g]) (void)hipFree(h->d_k2_tiles[g]);
    if (h->d_k2_pix[g]) (void)hipFree(h->d_k2_pix[g]);
  }
  if (h->d_zero16) (void)hipFree(h->d_zero16);
  delete h;
}

int xm_path_counts(xm_handle* h, uint64_t counts[4]) {
  if (!h || !counts) return fail(XM_ERR_INVALID, "NULL argument");
  for (int i = 0; i < 4; ++i) counts[i] = h->path_counts[i].load(std::memory_order_relaxed);
  return XM_OK;
}

int xm_cols_info(xm_handle* h, int32_t info[12]) {
  if (!h || !info) return fail(XM_ERR_INVALID, "NULL argument");
  for (int i = 0; i < 12; ++i) info[i] = 0;
  info[0] = !h->cols_ok ? 0 : h->own_mode ? 2 : 1;
  if (h->cols_ok && h->own_mode) {
    info[1] = h->own_w;
    info[2] = h->own_halo;
    info[3] = h->tb.own_nxs_max;
    info[4] = h->tb.shear_m;
    info[5] = h->tb.shear_extra;
    info[6] = h->tb.own_r_lo;
    info[7] = h->tb.own_hr;
    info[8] = h->own_extras;
    info[9] = h->tb.own_extra_max;
  }
  return XM_OK;
}

int xm_own_plan_info(const xm_config* cfg, int32_t info[12]) {
  if (!cfg || !info) return fail(XM_ERR_INVALID, "NULL argument");
  if (cfg->struct_size != sizeof(xm_config)) return fail(XM_ERR_INVALID, "xm_config.struct_size");
  if (!cfg->cam_mapx_i16 || !cfg->cam_mapy_i16 || !cfg->proj_x_map || cfg->cam_width <= 0 || cfg->cam_height <= 0 ||
      cfg->xmap_width <= 1 || cfg->rect_width <= 0 || cfg->rect_height <= 0)
    return fail(XM_ERR_INVALID, "bad tables");
  for (int i = 0; i < 12; ++i) info[i] = 0;
  const int xmap_h = cfg->xmap_height > 0 ? cfg->xmap_height : cfg->rect_height;
  int xr_min = 32767;
  for (size_t i = 0; i < (size_t)cfg->cam_width * cfg->cam_height; ++i) xr_min = std::min<int>(xr_min, cfg->cam_mapx_i16[i]);
  OwnPlan pl;
  own_plan(cfg, xmap_h, xr_min, pl);
  if (!pl.ok) return XM_OK;
  info[0] = 2; info[1] = pl.W; info[2] = pl.halo; info[3] = pl.nxs_max; info[4] = pl.m; info[5] = pl.extra_cols; info[6] = pl.r_lo;
  info[7] = pl.hr; info[8] = (int)pl.extra_flat.size() - 1; info[9] = pl.extra_max; info[10] = pl.delta_max;
  info[11] = (int)own_plan_lds_bytes(pl.nxs_max, pl.hrp, pl.extra_max);
  return XM_OK;
}

int xm_sorted_fallbacks(xm_handle* h, uint64_t* count) {
  if (!h || !count) return fail(XM_ERR_INVALID, "NULL argument");
  *count = h->sorted_fallbacks;
  return XM_OK;
}

// Wait for a stream: poll it for a while before blocking.  A blocking hipStreamSynchronize wakes up tens of microseconds
// after the stream has drained (interrupt path); the hot loop's frames are ~10 us, so a caller that brackets short bursts with
// xm_sync() (bench.py --steps 20: 0.2 ms of work) would spend a quarter of its time asleep.
static int wait_stream(hipStream_t st) {
  const auto give_up = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
  unsigned spins = 0;
  for (;;) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) return XM_OK;
    if (q != hipErrorNotReady) HIP_TRY(q);
    __builtin_ia32_pause();
    if ((++spins & 0xff) == 0 && std::chrono::steady_clock::now() > give_up) break;
  }
  HIP_TRY(hipStreamSynchronize(st));
  return XM_OK;
}

int xm_sync(xm_handle* h) {
  if (!h) return fail(XM_ERR_INVALID, "NULL handle");
  XM_ENTER(h);
  for (hipStream_t st : h->streams) {
    int rcw = wait_stream(st);
    if (rcw) return rcw;
  }
  for (hipStream_t gs : h->gstreams) {  // graph replays run on streams of their own
    int rcw = wait_stream(gs);
    if (rcw) return rcw;
  }
  for (Slot& s : h->slots) {  // every stream is idle: nothing left to order against
    s.pending_batch_ev = nullptr;
    s.eager_dirty = false;
  }
  if (h->try_sorted || h->gate_slots) {  // frames whose shortcut failed are redone now, then waited for
    for (Slot& s : h->slots) {
      bool redone = false;
      int rc = resolve_prev(h, s, &redone);
      if (rc) return rc;
      if (redone) {
        if ((rc = drain_workers(h))) return rc;
        HIP_TRY(hipStreamSynchronize(s.stream));
      }
    }
  }
  if (h->time_sorted) {  // any asynchronously processed frame that was not sorted after all?
    u32 bad = 0;
    // one copy of all slot states (3 KB each) instead of one synchronous 4-byte copy per slot (60 slots: 0.9 ms)
    std::vector<SlotState> hs(h->slots.size());
    HIP_TRY(hipMemcpy(hs.data(), h->d_states, sizeof(SlotState) * hs.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < hs.size(); ++i) {
      if (hs[i].unsorted_sticky) {
        bad += hs[i].unsorted_sticky;
        HIP_TRY(hipMemset(&h->slots[i].st->unsorted_sticky, 0, sizeof(u32)));
      }
    }
    if (bad) return fail(XM_ERR_UNSORTED, "XM_FLAG_TIME_SORTED: %u wavefront(s) saw events outside [t[0], t[n-1]] -- a frame "
                         "processed since the last xm_sync was not time-sorted, its output is invalid", bad);
  }
  return XM_OK;
}

#ifdef XM_BLOG
// experiments only: copy out (and clear) the per-block log of the hot kernels
int xm_debug_blog(unsigned long long* out /*[BLOG_FRAMES * BLOG_SLOTS * BLOG_PER][4]*/, unsigned int cap, unsigned int* n_out) {
  HIP_TRY(hipDeviceSynchronize());
  const unsigned int n = xm::BLOG_FRAMES * xm::BLOG_SLOTS * xm::BLOG_PER;
  if (out && cap >= n) HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(xm::g_blog), sizeof(unsigned long long) * 4 * n));
  void* p = nullptr;
  HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(xm::g_blog)));
  HIP_TRY(hipMemset(p, 0, sizeof(unsigned long long) * 4 * n));
  if (n_out) *n_out = n;
  return XM_OK;
}
#endif
#ifdef XM_ABLATE
// experiments only: copy out the s_memtime timeline written by k_scatter_tiled
int xm_debug_timeline(unsigned long long* out /*[64][16]*/) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(xm::g_timeline), sizeof(unsigned long long) * 64 * 16));
  return XM_OK;
}
#endif

// tests: the column-tile path's integer time thresholds of a frame with the given first / last stamp (thr[0 .. xmap_w])
int xm_debug_cols_thresholds(xm_handle* h, long long t_first, long long t_last, uint32_t* out_host) {
  if (!h || !out_host) return fail(XM_ERR_INVALID, "NULL argument");
  if ((unsigned long long)(t_last - t_first) >= 0xffffffffull && t_last >= t_first)
    return fail(XM_ERR_INVALID, "frames of 2^32 us or more do not take the column tiles");
  XM_ENTER(h);
  const int n = h->tb.xmap_w + 1;
  u32* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, sizeof(u32) * n));
  hipLaunchKernelGGL(k_debug_cols_thresholds, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, h->slots[0].stream, t_first, t_last,
                     h->tb.t_px_scale, h->tb.xmap_w, d);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(h->slots[0].stream);
  if (e == hipSuccess) e = hipMemcpy(out_host, d, sizeof(u32) * n, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HIP_TRY(e);
  return XM_OK;
}

void* xm_stream(xm_handle* h, int slot) {
  if (!h || slot < 0 || slot >= (int)h->slots.size()) return nullptr;
  return (void*)h->slots[slot].stream;
}

int xm_process_frame(xm_handle* h, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p, size_t n,
                     int t_dtype, int mem, float* depth_out, uint8_t* bgr_out, xm_frame_stats* stats) {
  EventsView ev;
  ev.x = x; ev.y = y; ev.t = t; ev.p = p; ev.n = n; ev.t_dtype = t_dtype; ev.use_p = p != nullptr;
  return process_common(h, ev, mem, depth_out, bgr_out, stats, false);
}

int xm_process_frame_aos(xm_handle* h, const void* eventcd16, size_t n, int use_polarity, int mem, float* depth_out,
                         uint8_t* bgr_out, xm_frame_stats* stats) {
  if (n && !eventcd16) return fail(XM_ERR_INVALID, "NULL event buffer");
  EventsView ev;
  static const uint4 dummy = {0, 0, 0, 0};
  ev.aos = eventcd16 ? eventcd16 : (const void*)&dummy;
  ev.n = n; ev.t_dtype = XM_T_INT64; ev.use_p = use_polarity != 0;
  if (mem == XM_MEM_DEVICE && n == 0) ev.aos = h ? (const void*)h->d_lut : ev.aos;  // any valid device address
  return process_common(h, ev, mem, depth_out, bgr_out, stats, false);
}

int xm_profile_frame(xm_handle* h, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p, size_t n,
                     int t_dtype, float* depth_out, uint8_t* bgr_out, xm_frame_stats* stats) {
  EventsView ev;
  ev.x = x; ev.y = y; ev.t = t; ev.p = p; ev.n = n; ev.t_dtype = t_dtype; ev.use_p = p != nullptr;
  return process_common(h, ev, XM_MEM_DEVICE, depth_out, bgr_out, stats, true);
}

int xm_last_frame_stats(xm_handle* h, xm_frame_stats* stats) {
  if (!h || !stats) return fail(XM_ERR_INVALID, "NULL argument");
  XM_ENTER(h);
  Slot& s = h->slots[h->last_slot];
  if (s.pending_batch_ev) {  // the slot's last frame ran inside a multi-frame launch / graph replay on another stream
    HIP_TRY(hipEventSynchronize(s.pending_batch_ev));
    s.pending_batch_ev = nullptr;
  }
  HIP_TRY(hipStreamSynchronize(s.stream));
  return fetch_stats(h, s, s.last_t_dtype, stats);
}

int xm_profile_event_overhead(xm_handle* h, int reps, float* ms_out) {
  if (!h || !ms_out || reps <= 0) return fail(XM_ERR_INVALID, "bad argument");
  XM_ENTER(h);
  Slot& s = h->slots[0];
  std::vector<float> v;
  for (int i = 0; i < reps; ++i) {
    HIP_TRY(hipEventRecord(h->prof_ev[0], s.stream));
    HIP_TRY(hipEventRecord(h->prof_ev[1], s.stream));
    HIP_TRY(hipStreamSynchronize(s.stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->prof_ev[0], h->prof_ev[1]));
    v.push_back(ms);
  }
  std::sort(v.begin(), v.end());
  *ms_out = v[v.size() / 2];
  return XM_OK;
}

// ---- a group of frames in one set of multi-frame launches ---------------------------------------------------
static int submit_group(xm_handle* h, const std::vector<EventsView>& evs, const std::vector<float*>& dep, const std::vector<uint8_t*>& bg,
                        float* gpu_ms, hipEvent_t* done_out = nullptr);

static int process_batch_impl(xm_handle* h, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p, int t_dtype,
                              const uint64_t* offsets_host, int n_frames, float* depth_out, uint8_t* bgr_out, float* gpu_ms,
                              const void* aos = nullptr) {
  if (!h || !offsets_host || n_frames <= 0) return fail(XM_ERR_INVALID, "bad argument");
  const int ns = (int)h->slots.size();
  if (n_frames > ns) return fail(XM_ERR_INVALID, "a batch of %d frames needs n_slots >= %d (handle has %d)", n_frames, n_frames, ns);
  XM_ENTER(h);
  const size_t px = (size_t)h->out_w * h->out_h;
  const size_t tsz = t_size(t_dtype);
  std::vector<EventsView> evs(n_frames);
  std::vector<float*> dep(n_frames);
  std::vector<uint8_t*> bg(n_frames);
  for (int f = 0; f < n_frames; ++f) {
    const u64 a = offsets_host[f], b = offsets_host[f + 1];
    if (b < a) return fail(XM_ERR_INVALID, "offsets must be non-decreasing");
    EventsView& ev = evs[f];
    if (aos) {  // Metavision EventCD records (16 bytes each), every event used
      ev.aos = (const char*)aos + a * 16;
      ev.t_dtype = XM_T_INT64;
    } else {
      ev.x = x + a; ev.y = y + a; ev.t = (const char*)t + a * tsz; ev.p = p ? p + a : nullptr;
      ev.t_dtype = t_dtype; ev.use_p = p != nullptr;
    }
    ev.n = (size_t)(b - a);
    int rc = check_events(ev);
    if (rc) return rc;
    dep[f] = depth_out ? depth_out + f * px : nullptr;
    bg[f] = bgr_out ? bgr_out + f * px * 3 : nullptr;
  }
  return submit_group(h, evs, dep, bg, gpu_ms);
}

// frames evs[f] -> outputs dep[f] / bg[f] (device pointers) as ONE group on the next n slots: one set of multi-frame launches
static int submit_group(xm_handle* h, const std::vector<EventsView>& evs, const std::vector<float*>& dep, const std::vector<uint8_t*>& bg,
                        float* gpu_ms, hipEvent_t* done_out) {
  const int n_frames = (int)evs.size(), ns = (int)h->slots.size();
  std::vector<int> idx(n_frames);
  for (int f = 0; f < n_frames; ++f) {
    idx[f] = (h->next_slot + f) % ns;
    int rc = resolve_prev(h, h->slots[idx[f]]);  // try-sorted verdict of the slot's previous frame (may redo it)
    if (rc) return rc;
  }
  h->next_slot = (h->next_slot + n_frames) % ns;
  h->last_slot = idx[n_frames - 1];
  // the group's stream: groups rotate over the distinct slot streams, so that the tail of one group's launches overlaps
  // the head of the next group's (whose slots are different ones)
  const int si = (int)(h->batch_counter++ % h->streams.size());
  hipStream_t stream = h->streams[si];
  const int k = h->desc_next;
  h->desc_next = (k + 1) % xm_handle::DESC_RING;
  if (h->desc_used[k]) HIP_TRY(hipEventSynchronize(h->desc_ev[k]));  // the ring entry's previous batch has long finished
  FrameDesc* hd = h->h_descs + (size_t)k * ns;
  FrameDesc* dd = h->d_descs + (size_t)k * ns;
  int kinds[2] = {-1, -1};  // (stay -1 when the group fell back to frame-by-frame launches: nothing was attached then)
  int rc = enqueue_batch(h, idx.data(), evs.data(), dep.data(), bg.data(), n_frames, stream, hd, dd, true, true,
                         gpu_ms ? h->prof_ev : nullptr, kinds);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(h->desc_ev[k], stream));
  h->desc_used[k] = true;
  hipEvent_t done = h->batch_ev[si][h->batch_ev_next[si]++ % 8];
  HIP_TRY(hipEventRecord(done, stream));
  if (done_out) *done_out = done;
  for (int f = 0; f < n_frames; ++f) {
    Slot& s = h->slots[idx[f]];
    s.pending_batch_ev = done;
    s.pending_batch_stream = stream;
    if (s.h_flags) {  // slot gate + try-sorted verdict, read when the slot comes round again or in xm_sync
      s.prev.valid = true;
      s.prev.check = h->try_sorted && s.last_sorted;
      s.prev.ev = evs[f];
      s.prev.depth = dep[f];
      s.prev.bgr = bg[f];
      s.prev.host_depth = nullptr;
      s.prev.host_bgr = nullptr;
      s.prev.tag = s.host_tag;
      s.prev.stream = stream;
    }
  }
  if (gpu_ms) {  // profile mode: durations of the group's dispatches (the events were attached to the dispatch packets)
    HIP_TRY(hipStreamSynchronize(stream));
    gpu_ms[0] = gpu_ms[1] = gpu_ms[2] = gpu_ms[3] = 0.0f;
    const int first = kinds[0] > 0 ? 0 : 1;  // no K0 / K0b launch on the verified-sorted keyed paths
    if (kinds[1] >= 0) {
      for (int i = first; i < 3; ++i) HIP_TRY(hipEventElapsedTime(&gpu_ms[i], h->prof_ev[2 * i], h->prof_ev[2 * i + 1]));
      HIP_TRY(hipEventElapsedTime(&gpu_ms[3], h->prof_ev[2 * first], h->prof_ev[5]));
    }
  }
  return XM_OK;
}

}  // extern "C"

// XM_FLAG_ADAPTIVE_BATCH: everything on the pending list goes out as one group (at most ab_max = n_slots / 4 frames)
int flush_pending(xm_handle* h) {
  if (h->pending.empty()) return XM_OK;
  const size_t n = h->pending.size();
  std::vector<EventsView> evs(n);
  std::vector<float*> dep(n);
  std::vector<uint8_t*> bg(n);
  for (size_t i = 0; i < n; ++i) {
    evs[i] = h->pending[i].ev;
    dep[i] = h->pending[i].depth;
    bg[i] = h->pending[i].bgr;
  }
  h->pending.clear();  // (first: submit_group's callees pass through XM_ENTER-free paths only, but keep re-entry harmless)
  hipEvent_t done = nullptr;
  int rc = submit_group(h, evs, dep, bg, nullptr, &done);
  if (rc) return rc;
  h->ab_inflight[h->ab_groups & 3] = done;
  h->ab_groups += 1;
  h->ab_frames += n;
  return XM_OK;
}

extern "C" {

int xm_process_batch(xm_handle* h, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p, int t_dtype,
                     const uint64_t* offsets_host, int n_frames, float* depth_out, uint8_t* bgr_out) {
  return process_batch_impl(h, x, y, t, p, t_dtype, offsets_host, n_frames, depth_out, bgr_out, nullptr);
}

int xm_process_batch_aos(xm_handle* h, const void* eventcd16, const uint64_t* offsets_host, int n_frames, float* depth_out,
                         uint8_t* bgr_out) {
  if (!eventcd16) return fail(XM_ERR_INVALID, "NULL event buffer");
  return process_batch_impl(h, nullptr, nullptr, nullptr, nullptr, XM_T_INT64, offsets_host, n_frames, depth_out, bgr_out, nullptr,
                            eventcd16);
}

int xm_profile_batch(xm_handle* h, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p, int t_dtype,
                     const uint64_t* offsets_host, int n_frames, float* depth_out, uint8_t* bgr_out, float gpu_ms[4]) {
  if (!gpu_ms) return fail(XM_ERR_INVALID, "NULL argument");
  return process_batch_impl(h, x, y, t, p, t_dtype, offsets_host, n_frames, depth_out, bgr_out, gpu_ms);
}

// ---- hipGraph batch ------------------------------------------------------------------------------------
// Default: the frames are captured as GROUPS of multi-frame launches (3 kernel nodes per group instead of 3 per frame).
// With n_slots >= n_frames the whole batch is one group; otherwise groups of n_slots / 2 frames alternate between two
// capture streams (each half of the slots always on the same branch), so that one group's tail overlaps the next one's head.
// XM_GRAPH_PER_FRAME=1 (experiments) keeps the round-1 form: three nodes per frame, frames forked over the slots' streams.
static int graph_capture_batched(xm_handle* h, xm_graph* g, const uint16_t* x, const uint16_t* y, const void* t,
                                 const int16_t* p, int t_dtype, const uint64_t* offsets_host, int n_frames,
                                 float* depth_out, uint8_t* bgr_out) {
  const int ns = (int)h->slots.size();
  const size_t px = (size_t)h->out_w * h->out_h;
  const size_t tsz = t_size(t_dtype);
  const bool two = ns >= 2 && n_frames > ns;
  const int G = two ? ns / 2 : std::min(ns, n_frames);
  g->h_descs.resize(2 * (size_t)n_frames);  // [n_frames] the frames, [n_frames] the same frames on their slots' 64-bit key frames
  for (auto& d : g->h_descs) d = FrameDesc{};  // (valid = 0: unused entries are skipped by every kernel)
  HIP_TRY(hipMalloc((void**)&g->d_descs, sizeof(FrameDesc) * 2 * n_frames));
  hipStream_t origin = h->gstreams[0], second = two ? h->gstreams[1] : nullptr;
  int rc = XM_OK;
  hipError_t e = hipSuccess;
  h->capturing = true;
  e = hipStreamBeginCapture(origin, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) {
    h->capturing = false;
    return fail(XM_ERR_HIP, "hipStreamBeginCapture: %s", hipGetErrorString(e));
  }
  do {
    if (two) {
      if ((e = hipEventRecord(h->fork_ev, origin)) != hipSuccess) break;
      if ((e = hipStreamWaitEvent(second, h->fork_ev, 0)) != hipSuccess) break;
    }
    int gi = 0;
    for (int f0 = 0; f0 < n_frames && rc == XM_OK; f0 += G, ++gi) {
      const int nf = std::min(G, n_frames - f0);
      const int half = two ? gi & 1 : 0;
      std::vector<int> idx(nf);
      std::vector<EventsView> evs(nf);
      std::vector<float*> dep(nf);
      std::vector<uint8_t*> bg(nf);
      for (int j = 0; j < nf; ++j) {
        const int f = f0 + j;
        const u64 a = offsets_host[f], b = offsets_host[f + 1];
        EventsView& ev = evs[j];
        ev.x = x + a; ev.y = y + a; ev.t = (const char*)t + a * tsz; ev.p = p ? p + a : nullptr;
        ev.n = (size_t)(b - a); ev.t_dtype = t_dtype; ev.use_p = p != nullptr;
        if ((rc = check_events(ev))) break;
        idx[j] = half * G + j;
        dep[j] = depth_out ? depth_out + f * px : nullptr;
        bg[j] = bgr_out ? bgr_out + f * px * 3 : nullptr;
        h->slots[idx[j]].host_tag = 0;  // tags advance on the device inside a graph; no reset mid-capture
        g->frames_on_slot[idx[j]] += 1;
      }
      if (rc) break;
      rc = enqueue_batch(h, idx.data(), evs.data(), dep.data(), bg.data(), nf, half ? second : origin,
                         g->h_descs.data() + f0, g->d_descs + f0, false, true, nullptr, nullptr,
                         g->h_descs.data() + n_frames + f0, g->d_descs + n_frames + f0);
    }
    if (two && rc == XM_OK) {
      if ((e = hipEventRecord(h->join_ev[1], second)) != hipSuccess) break;
      if ((e = hipStreamWaitEvent(origin, h->join_ev[1], 0)) != hipSuccess) break;
    }
  } while (0);
  hipError_t e2 = hipStreamEndCapture(origin, &g->graph);
  h->capturing = false;
  if (rc == XM_OK && (e != hipSuccess || e2 != hipSuccess))
    rc = fail(XM_ERR_HIP, "graph capture failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
  if (rc == XM_OK)  // the descriptors are static: one upload for the graph's lifetime
    HIP_TRY(hipMemcpy(g->d_descs, g->h_descs.data(), sizeof(FrameDesc) * 2 * n_frames, hipMemcpyHostToDevice));
  return rc;
}

int xm_graph_create(xm_handle* h, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p, int t_dtype,
                    const uint64_t* offsets_host, int n_frames, float* depth_out, uint8_t* bgr_out, xm_graph** out) {
  if (!h || !out || !offsets_host || n_frames <= 0) return fail(XM_ERR_INVALID, "bad argument");
  *out = nullptr;
  XM_ENTER(h);
  const int ns = (int)h->slots.size();
  if ((u64)n_frames / ns + 1 >= KEY_MAX_TAG) return fail(XM_ERR_INVALID, "too many frames per graph");
  for (Slot& s : h->slots) HIP_TRY(hipStreamSynchronize(s.stream));
  for (Slot& s : h->slots) {  // idle: nothing to order the capture against
    s.pending_batch_ev = nullptr;
    s.eager_dirty = false;
  }
  xm_graph* g = new (std::nothrow) xm_graph();
  if (!g) return fail(XM_ERR_NOMEM, "out of host memory");
  g->h = h;
  g->n_frames = n_frames;
  g->frames_on_slot.assign(ns, 0);
  const size_t px = (size_t)h->out_w * h->out_h;
  const size_t tsz = t_size(t_dtype);
  struct Saved {
    u32 host_tag, api_tag;
    bool any_frame, last_sorted;
    uint64_t last_n;
    int last_t_dtype;
  };
  std::vector<Saved> saved(ns);
  for (int i = 0; i < ns; ++i) {
    const Slot& s = h->slots[i];
    saved[i] = Saved{s.host_tag, s.api_tag, s.any_frame, s.last_sorted, s.last_n, s.last_t_dtype};
  }
  // Graphs are captured on (and launched from) default-priority streams of their own: launched from the slots'
  // high-priority streams the replay ran its branches one after the other (28 instead of 61 Gevents/s).
  if (h->gstreams.empty()) {
    h->gstreams.assign(std::max(ns, 2), nullptr);
    for (auto& gs : h->gstreams) {
      hipError_t ce = hipStreamCreateWithFlags(&gs, hipStreamNonBlocking);
      if (ce != hipSuccess) {
        delete g;
        return fail(XM_ERR_HIP, "hipStreamCreateWithFlags: %s", hipGetErrorString(ce));
      }
    }
  }
  static const bool per_frame = getenv("XM_GRAPH_PER_FRAME") && getenv("XM_GRAPH_PER_FRAME")[0] == '1';
  int rc = XM_OK;
  if (!per_frame) {
    rc = graph_capture_batched(h, g, x, y, t, p, t_dtype, offsets_host, n_frames, depth_out, bgr_out);
  } else {
    for (int i = 0; i < ns; ++i) std::swap(h->slots[i].stream, h->gstreams[i]);  // enqueue_frame launches on slot.stream
    hipStream_t origin = h->slots[0].stream;
    h->capturing = true;
    hipError_t e = hipStreamBeginCapture(origin, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
      h->capturing = false;
      for (int i = 0; i < ns; ++i) std::swap(h->slots[i].stream, h->gstreams[i]);
      delete g;
      return fail(XM_ERR_HIP, "hipStreamBeginCapture: %s", hipGetErrorString(e));
    }
    do {
      if (ns > 1) {
        if ((e = hipEventRecord(h->fork_ev, origin)) != hipSuccess) break;
        for (int i = 1; i < ns; ++i)
          if ((e = hipStreamWaitEvent(h->slots[i].stream, h->fork_ev, 0)) != hipSuccess) break;
        if (e != hipSuccess) break;
      }
      for (int f = 0; f < n_frames && rc == XM_OK; ++f) {
        Slot& s = h->slots[f % ns];
        EventsView ev;
        const u64 a = offsets_host[f], b = offsets_host[f + 1];
        ev.x = x + a; ev.y = y + a; ev.t = (const char*)t + a * tsz; ev.p = p ? p + a : nullptr;
        ev.n = (size_t)(b - a); ev.t_dtype = t_dtype; ev.use_p = p != nullptr;
        if ((rc = check_events(ev))) break;
        // tags inside a graph advance on the device; keep the host mirror from triggering a reset mid-capture
        s.host_tag = 0;
        rc = enqueue_frame(h, s, ev, depth_out ? depth_out + f * px : nullptr, bgr_out ? bgr_out + f * px * 3 : nullptr,
                           nullptr);
        g->frames_on_slot[f % ns] += 1;
      }
      if (ns > 1) {
        for (int i = 1; i < ns; ++i) {
          if ((e = hipEventRecord(h->join_ev[i], h->slots[i].stream)) != hipSuccess) break;
          if ((e = hipStreamWaitEvent(origin, h->join_ev[i], 0)) != hipSuccess) break;
        }
      }
    } while (0);
    hipError_t e2 = hipStreamEndCapture(origin, &g->graph);
    h->capturing = false;
    for (int i = 0; i < ns; ++i) std::swap(h->slots[i].stream, h->gstreams[i]);
    if (rc == XM_OK && (e != hipSuccess || e2 != hipSuccess))
      rc = fail(XM_ERR_HIP, "graph capture failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
  }
  for (int i = 0; i < ns; ++i) {  // capture only recorded launches: the slots are where they were
    Slot& s = h->slots[i];
    s.host_tag = saved[i].host_tag; s.api_tag = saved[i].api_tag; s.any_frame = saved[i].any_frame;
    s.last_sorted = saved[i].last_sorted; s.last_n = saved[i].last_n; s.last_t_dtype = saved[i].last_t_dtype;
    s.pending_batch_ev = nullptr;
    s.eager_dirty = false;
  }
  if (rc == XM_OK) {
    hipError_t e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) rc = fail(XM_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
  }
  if (rc != XM_OK) {
    xm_graph_destroy(g);
    return rc;
  }
  *out = g;
  return XM_OK;
}

int xm_graph_launch(xm_graph* g) {
  if (!g || !g->exec) return fail(XM_ERR_INVALID, "NULL graph");
  xm_handle* h = g->h;
  XM_ENTER(h);
  const int ns = (int)h->slots.size();
  hipStream_t origin = h->gstreams[0];
  if (h->try_sorted || h->gate_slots)
    for (Slot& s : h->slots) {  // settle pending try-sorted verdicts before the replay advances the slots' tags
      int rc = resolve_prev(h, s);
      if (rc) return rc;
    }
  // order the replay after whatever the slots did last -- per distinct stream, and only where something is pending (a
  // handle that only replays graphs pays one hipGraphLaunch + one hipEventRecord per replay, not 3 API calls per slot)
  for (size_t si = 0; si < h->streams.size(); ++si) {
    bool dirty = false;
    for (Slot& s : h->slots)
      if (s.stream == h->streams[si] && s.eager_dirty) dirty = true;
    if (dirty) {
      HIP_TRY(hipEventRecord(h->join_ev[si % h->join_ev.size()], h->streams[si]));
      HIP_TRY(hipStreamWaitEvent(origin, h->join_ev[si % h->join_ev.size()], 0));
    }
  }
  for (Slot& s : h->slots) {
    s.eager_dirty = false;
    if (s.pending_batch_ev) {
      if (s.pending_batch_stream != origin) HIP_TRY(hipStreamWaitEvent(origin, s.pending_batch_ev, 0));
      s.pending_batch_ev = nullptr;
    }
  }
  for (int i = 0; i < ns; ++i) {  // tag wrap per slot
    Slot& s = h->slots[i];
    if ((u64)s.host_tag + g->frames_on_slot[i] >= KEY_MAX_TAG) {
      hipLaunchKernelGGL(k_reset_slot, dim3(1024), dim3(BLOCK), 0, origin, s.st, s.key_frame, (u64)h->key_cells, s.dirty);
      HIP_TRY(hipGetLastError());
      s.host_tag = 0;
      s.api_tag = 0;
    }
  }
  HIP_TRY(hipGraphLaunch(g->exec, origin));
  // whatever a slot does next on its own stream waits for the replay (lazily, see enqueue_frame / enqueue_batch)
  hipEvent_t done = h->graph_ev[h->graph_ev_next++ % 8];
  HIP_TRY(hipEventRecord(done, origin));
  for (int i = 0; i < ns; ++i) {
    Slot& s = h->slots[i];
    s.host_tag += g->frames_on_slot[i];
    s.api_tag = s.host_tag;  // the worker path derives the next frame's tag from api_tag
    if (g->frames_on_slot[i]) {
      s.any_frame = true;
      s.pending_batch_ev = done;
      s.pending_batch_stream = origin;
    }
  }
  return XM_OK;
}

void xm_graph_destroy(xm_graph* g) {
  if (!g) return;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  if (g->d_descs) (void)hipFree(g->d_descs);
  delete g;
}

// ---- debug: all per-event intermediates -----------------------------------------------------------------
int xm_debug_event_outputs(xm_handle* h, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p, size_t n,
                           int t_dtype, int mem, int16_t* xr, int16_t* yr, int16_t* ts, int16_t* disp, uint8_t* mask) {
  if (!h) return fail(XM_ERR_INVALID, "NULL handle");
  XM_ENTER(h);
  EventsView ev;
  ev.x = x; ev.y = y; ev.t = t; ev.p = p; ev.n = n; ev.t_dtype = t_dtype; ev.use_p = p != nullptr;
  int rc = check_events(ev);
  if (rc) return rc;
  if (n == 0) return XM_OK;
  Slot& s = h->slots[0];
  void* outs_host[5] = {xr, yr, ts, disp, mask};
  void* outs_dev[5] = {xr, yr, ts, disp, mask};
  const size_t osz[5] = {2, 2, 2, 2, 1};
  if (mem == XM_MEM_HOST) {
    if ((rc = stage_in(s.ev_x, x, n * 2, s.stream))) return rc;
    if ((rc = stage_in(s.ev_y, y, n * 2, s.stream))) return rc;
    if ((rc = stage_in(s.ev_t, t, n * t_size(t_dtype), s.stream))) return rc;
    ev.x = (const uint16_t*)s.ev_x.p; ev.y = (const uint16_t*)s.ev_y.p; ev.t = s.ev_t.p;
    if (p) {
      if ((rc = stage_in(s.ev_p, p, n * 2, s.stream))) return rc;
      ev.p = (const int16_t*)s.ev_p.p;
    }
    for (int i = 0; i < 5; ++i)
      if (outs_host[i]) {
        if ((rc = s.dbg[i].reserve(n * osz[i]))) return rc;
        outs_dev[i] = s.dbg[i].p;
      }
  }
  if ((rc = rearm_aux(h, s.stream, nullptr, 0))) return rc;
  launch_minmax(ev, h->aux_st, 2, s.stream);
  const unsigned grid = grid_for(n, BLOCK);
#define XM_DBG(T, HP)                                                                                           \
  hipLaunchKernelGGL((k_debug_events<T, HP>), dim3(grid), dim3(BLOCK), 0, s.stream, ev.x, ev.y, (const T*)ev.t, \
                     ev.p, (u64)n, h->tb, h->aux_st, 2u, (int16_t*)outs_dev[0], (int16_t*)outs_dev[1],          \
                     (int16_t*)outs_dev[2], (int16_t*)outs_dev[3], (uint8_t*)outs_dev[4])
  switch (t_dtype) {
    case XM_T_INT64: if (p) XM_DBG(long long, true); else XM_DBG(long long, false); break;
    case XM_T_FLOAT32: if (p) XM_DBG(float, true); else XM_DBG(float, false); break;
    default: if (p) XM_DBG(double, true); else XM_DBG(double, false);
  }
#undef XM_DBG
  HIP_TRY(hipGetLastError());
  if (mem == XM_MEM_HOST)
    for (int i = 0; i < 5; ++i)
      if (outs_host[i]) HIP_TRY(hipMemcpyAsync(outs_host[i], outs_dev[i], n * osz[i], hipMemcpyDeviceToHost, s.stream));
  HIP_TRY(hipStreamSynchronize(s.stream));
  return XM_OK;
}

// ---- stage API (host pointers, synchronous) ---------------------------------------------------------------
static int read_oob(xm_handle* h, hipStream_t stream, const char* what) {
  u32 c[CNT_STRIDE];
  HIP_TRY(hipMemcpyAsync(c, &h->aux_st->cnt[0][0][0], sizeof c, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  if (c[CNT_OOB]) return fail(XM_ERR_INDEX, "%s: %u index(es) out of range (IndexError in the reference)", what, c[CNT_OOB]);
  return XM_OK;
}

int xm_stage_rectify(xm_handle* h, const uint16_t* x, const uint16_t* y, size_t n, int16_t* xr, int16_t* yr) {
  if (!h || (n && (!x || !y || !xr || !yr))) return fail(XM_ERR_INVALID, "NULL argument");
  XM_ENTER(h);
  if (n == 0) return XM_OK;
  Slot& s = h->slots[0];
  int rc;
  if ((rc = stage_in(s.ev_x, x, n * 2, s.stream))) return rc;
  if ((rc = stage_in(s.ev_y, y, n * 2, s.stream))) return rc;
  if ((rc = s.dbg[0].reserve(n * 2)) || (rc = s.dbg[1].reserve(n * 2))) return rc;
  if ((rc = rearm_aux(h, s.stream, nullptr, 0))) return rc;
  hipLaunchKernelGGL(k_stage_rectify, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, s.stream, (const uint16_t*)s.ev_x.p,
                     (const uint16_t*)s.ev_y.p, (u64)n, h->tb, (int16_t*)s.dbg[0].p, (int16_t*)s.dbg[1].p,
                     &h->aux_st->cnt[0][0][CNT_OOB]);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(xr, s.dbg[0].p, n * 2, hipMemcpyDeviceToHost, s.stream));
  HIP_TRY(hipMemcpyAsync(yr, s.dbg[1].p, n * 2, hipMemcpyDeviceToHost, s.stream));
  return read_oob(h, s.stream, "rectify_cam_coords_i16");
}

int xm_stage_rectify_f32(xm_handle* h, const float* mapx_f32, const float* mapy_f32, const uint16_t* x, const uint16_t* y,
                         size_t n, float* xr, float* yr) {
  if (!h || !mapx_f32 || !mapy_f32 || (n && (!x || !y || !xr || !yr))) return fail(XM_ERR_INVALID, "NULL argument");
  XM_ENTER(h);
  if (n == 0) return XM_OK;
  Slot& s = h->slots[0];
  const size_t map_bytes = (size_t)h->cfg.cam_width * h->cfg.cam_height * 4;
  int rc;
  if ((rc = stage_in(s.ev_x, x, n * 2, s.stream))) return rc;
  if ((rc = stage_in(s.ev_y, y, n * 2, s.stream))) return rc;
  if ((rc = stage_in(s.dbg[2], mapx_f32, map_bytes, s.stream))) return rc;
  if ((rc = stage_in(s.dbg[3], mapy_f32, map_bytes, s.stream))) return rc;
  if ((rc = s.dbg[0].reserve(n * 4)) || (rc = s.dbg[1].reserve(n * 4))) return rc;
  if ((rc = rearm_aux(h, s.stream, nullptr, 0))) return rc;
  hipLaunchKernelGGL(k_stage_rectify_f32, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, s.stream, (const uint16_t*)s.ev_x.p,
                     (const uint16_t*)s.ev_y.p, (u64)n, h->cfg.cam_width, h->cfg.cam_height, (const float*)s.dbg[2].p,
                     (const float*)s.dbg[3].p, (float*)s.dbg[0].p, (float*)s.dbg[1].p, &h->aux_st->cnt[0][0][CNT_OOB]);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(xr, s.dbg[0].p, n * 4, hipMemcpyDeviceToHost, s.stream));
  HIP_TRY(hipMemcpyAsync(yr, s.dbg[1].p, n * 4, hipMemcpyDeviceToHost, s.stream));
  return read_oob(h, s.stream, "rectify_cam_coords_f32");
}

int xm_stage_point_cloud(xm_handle* h, const double* Q, const float* xpr, const float* ypr, const float* disp, size_t n,
                         float* cloud) {
  if (!h || !Q || (n && (!xpr || !ypr || !disp || !cloud))) return fail(XM_ERR_INVALID, "NULL argument");
  XM_ENTER(h);
  if (n == 0) return XM_OK;
  Slot& s = h->slots[0];
  int rc;
  if ((rc = stage_in(s.dbg[0], xpr, n * 4, s.stream))) return rc;
  if ((rc = stage_in(s.dbg[1], ypr, n * 4, s.stream))) return rc;
  if ((rc = stage_in(s.dbg[2], disp, n * 4, s.stream))) return rc;
  if ((rc = s.dbg[3].reserve(n * 12))) return rc;
  Mat4f q;
  for (int i = 0; i < 16; ++i) q.m[i] = (float)Q[i];  // self.Q.astype(np.float32)
  hipLaunchKernelGGL(k_point_cloud, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, s.stream, (const float*)s.dbg[0].p,
                     (const float*)s.dbg[1].p, (const float*)s.dbg[2].p, (u64)n, q, (float*)s.dbg[3].p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(cloud, s.dbg[3].p, n * 12, hipMemcpyDeviceToHost, s.stream));
  HIP_TRY(hipStreamSynchronize(s.stream));
  return XM_OK;
}

int xm_stage_event_disparity(xm_handle* h, const int16_t* xr, const int16_t* yr, const void* t, size_t n, int t_dtype,
                             int16_t* disp, uint8_t* mask) {
  if (!h || (n && (!xr || !yr || !t || !disp || !mask))) return fail(XM_ERR_INVALID, "NULL argument");
  XM_ENTER(h);
  if (n == 0) return XM_OK;
  Slot& s = h->slots[0];
  int rc;
  if ((rc = stage_in(s.ev_x, xr, n * 2, s.stream))) return rc;
  if ((rc = stage_in(s.ev_y, yr, n * 2, s.stream))) return rc;
  if ((rc = stage_in(s.ev_t, t, n * t_size(t_dtype), s.stream))) return rc;
  if ((rc = s.dbg[3].reserve(n * 2)) || (rc = s.dbg[4].reserve(n))) return rc;
  if ((rc = rearm_aux(h, s.stream, nullptr, 0))) return rc;
  EventsView ev;
  ev.t = s.ev_t.p; ev.n = n; ev.t_dtype = t_dtype;
  ev.x = (const uint16_t*)s.ev_x.p; ev.y = (const uint16_t*)s.ev_y.p;
  launch_minmax(ev, h->aux_st, 2, s.stream);
  const unsigned grid = grid_for(n, BLOCK);
#define XM_ED(T)                                                                                                   \
  hipLaunchKernelGGL((k_stage_event_disparity<T>), dim3(grid), dim3(BLOCK), 0, s.stream, (const int16_t*)s.ev_x.p, \
                     (const int16_t*)s.ev_y.p, (const T*)s.ev_t.p, (u64)n, h->tb, h->aux_st, 2u,                   \
                     (int16_t*)s.dbg[3].p, (uint8_t*)s.dbg[4].p)
  switch (t_dtype) {
    case XM_T_INT64: XM_ED(long long); break;
    case XM_T_FLOAT32: XM_ED(float); break;
    case XM_T_FLOAT64: XM_ED(double); break;
    default: return fail(XM_ERR_INVALID, "unknown t_dtype");
  }
#undef XM_ED
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(disp, s.dbg[3].p, n * 2, hipMemcpyDeviceToHost, s.stream));
  HIP_TRY(hipMemcpyAsync(mask, s.dbg[4].p, n, hipMemcpyDeviceToHost, s.stream));
  HIP_TRY(hipStreamSynchronize(s.stream));
  return XM_OK;
}

static int stage_scatter_common(xm_handle* h, int view, const void* a, const void* b, const int16_t* disp,
                                const uint8_t* mask, size_t n, float* disp_map) {
  if (!h || !disp_map || (n && (!a || !b || !disp || !mask))) return fail(XM_ERR_INVALID, "NULL argument");
  XM_ENTER(h);
  if (n >= XM_KEY_MAX_EVENTS) return fail(XM_ERR_TOO_MANY, "too many events");
  Slot& s = h->slots[0];
  int rc;
  if ((rc = ensure_stage_frame(h))) return rc;
  const u64 cells = view == 0 ? (u64)h->tb.rect_w * h->tb.rect_h : (u64)h->tb.cam_w * h->tb.cam_h;
  if ((rc = stage_in(s.ev_x, a, n * 2, s.stream))) return rc;
  if ((rc = stage_in(s.ev_y, b, n * 2, s.stream))) return rc;
  if ((rc = stage_in(s.dbg[3], disp, n * 2, s.stream))) return rc;
  if ((rc = stage_in(s.dbg[4], mask, n, s.stream))) return rc;
  if ((rc = s.out_depth.reserve(cells * 4))) return rc;
  if ((rc = rearm_aux(h, s.stream, h->stage_frame, cells))) return rc;
  if (n) {
    const unsigned grid = grid_for(n, BLOCK);
    if (view == 0)
      hipLaunchKernelGGL((k_stage_scatter<0>), dim3(grid), dim3(BLOCK), 0, s.stream, (const int16_t*)s.ev_x.p,
                         (const int16_t*)s.ev_y.p, (const uint16_t*)nullptr, (const uint16_t*)nullptr,
                         (const int16_t*)s.dbg[3].p, (const uint8_t*)s.dbg[4].p, (u64)n, h->tb, 1u, h->stage_frame,
                         &h->aux_st->cnt[0][0][CNT_OOB]);
    else
      hipLaunchKernelGGL((k_stage_scatter<1>), dim3(grid), dim3(BLOCK), 0, s.stream, (const int16_t*)nullptr,
                         (const int16_t*)nullptr, (const uint16_t*)s.ev_x.p, (const uint16_t*)s.ev_y.p,
                         (const int16_t*)s.dbg[3].p, (const uint8_t*)s.dbg[4].p, (u64)n, h->tb, 1u, h->stage_frame,
                         &h->aux_st->cnt[0][0][CNT_OOB]);
    HIP_TRY(hipGetLastError());
  }
  hipLaunchKernelGGL(k_decode_keys_signed, dim3(grid_for(cells, BLOCK)), dim3(BLOCK), 0, s.stream, h->stage_frame, cells,
                     1u, (float*)s.out_depth.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(disp_map, s.out_depth.p, cells * 4, hipMemcpyDeviceToHost, s.stream));
  return read_oob(h, s.stream, view == 0 ? "compute_disp_map_projector_view" : "compute_disp_map_camera_view");
}

int xm_stage_disp_map_projector_view(xm_handle* h, const int16_t* xr, const int16_t* yr, const int16_t* disp,
                                     const uint8_t* mask, size_t n, float* disp_map) {
  return stage_scatter_common(h, 0, xr, yr, disp, mask, n, disp_map);
}

int xm_stage_disp_map_camera_view(xm_handle* h, const uint16_t* x, const uint16_t* y, const int16_t* disp,
                                  const uint8_t* mask, size_t n, float* disp_map) {
  return stage_scatter_common(h, 1, x, y, disp, mask, n, disp_map);
}

int xm_stage_remap_rectified_disp_map_to_proj(xm_handle* h, const float* rect_disp, float* proj_disp) {
  if (!h || !rect_disp || !proj_disp) return fail(XM_ERR_INVALID, "NULL argument");
  if (!h->d_pmap) return fail(XM_ERR_INVALID, "handle was created without disp_proj_mapxy_i16");
  XM_ENTER(h);
  Slot& s = h->slots[0];
  const size_t cells = (size_t)h->tb.rect_w * h->tb.rect_h, px = (size_t)h->tb.proj_w * h->tb.proj_h;
  int rc;
  if ((rc = stage_in(s.dbg[0], rect_disp, cells * 4, s.stream))) return rc;
  if ((rc = s.out_depth.reserve(px * 4))) return rc;
  F32Cells cellsv{(const float*)s.dbg[0].p};
  hipLaunchKernelGGL((k_frame_proj<F32Cells, 1>), dim3(grid_for(px, BLOCK)), dim3(BLOCK), 0, s.stream, cellsv, h->tb,
                     (SlotState*)nullptr, 0u, (float*)s.out_depth.p, (uint8_t*)nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(proj_disp, s.out_depth.p, px * 4, hipMemcpyDeviceToHost, s.stream));
  HIP_TRY(hipStreamSynchronize(s.stream));
  return XM_OK;
}

static int stage_pixels(xm_handle* h, const float* disp, int height, int width, float* depth, uint8_t* bgr) {
  if (!h || !disp || height <= 0 || width <= 0) return fail(XM_ERR_INVALID, "bad argument");
  XM_ENTER(h);
  Slot& s = h->slots[0];
  const size_t px = (size_t)height * width;
  int rc;
  if ((rc = stage_in(s.dbg[0], disp, px * 4, s.stream))) return rc;
  if (depth && (rc = s.out_depth.reserve(px * 4))) return rc;
  if (bgr && (rc = s.out_bgr.reserve(px * 3))) return rc;
  F32Cells cellsv{(const float*)s.dbg[0].p};
  hipLaunchKernelGGL((k_frame_direct<F32Cells>), dim3(grid_for(px, BLOCK)), dim3(BLOCK), 0, s.stream, cellsv, (u64)px,
                     h->tb.p03, h->tb.z_near, h->tb.z_far, (SlotState*)nullptr, 0u, 0, (const uint2*)nullptr,
                     depth ? (float*)s.out_depth.p : nullptr, bgr ? (uint8_t*)s.out_bgr.p : nullptr);
  HIP_TRY(hipGetLastError());
  if (depth) HIP_TRY(hipMemcpyAsync(depth, s.out_depth.p, px * 4, hipMemcpyDeviceToHost, s.stream));
  if (bgr) HIP_TRY(hipMemcpyAsync(bgr, s.out_bgr.p, px * 3, hipMemcpyDeviceToHost, s.stream));
  HIP_TRY(hipStreamSynchronize(s.stream));
  return XM_OK;
}

int xm_stage_disparity_to_depth(xm_handle* h, const float* disp, int height, int width, float* depth) {
  if (!depth) return fail(XM_ERR_INVALID, "NULL output");
  return stage_pixels(h, disp, height, width, depth, nullptr);
}

int xm_stage_colorize_depth_from_disp(xm_handle* h, const float* disp, int height, int width, uint8_t* bgr) {
  if (!bgr) return fail(XM_ERR_INVALID, "NULL output");
  return stage_pixels(h, disp, height, width, nullptr, bgr);
}

// ---- shards -----------------------------------------------------------------------------------------------
int xm_shard_minmax(xm_handle* h, const void* t, const int16_t* p, size_t n, int t_dtype, void* minmax_out_host) {
  if (!h || !minmax_out_host || (n && !t)) return fail(XM_ERR_INVALID, "NULL argument");
  XM_ENTER(h);
  Slot& s = h->slots[0];
  int rc;
  if ((rc = rearm_aux(h, s.stream, nullptr, 0))) return rc;
  EventsView ev;
  ev.t = n ? t : (const void*)h->d_lut; ev.p = p; ev.n = n; ev.t_dtype = t_dtype; ev.use_p = p != nullptr;
  ev.x = (const uint16_t*)h->d_lut; ev.y = ev.x;
  launch_minmax(ev, h->aux_st, 2, s.stream);
  HIP_TRY(hipGetLastError());
  SlotState hs;
  HIP_TRY(hipMemcpyAsync(&hs, h->aux_st, sizeof hs, hipMemcpyDeviceToHost, s.stream));
  HIP_TRY(hipStreamSynchronize(s.stream));
  switch (t_dtype) {
    case XM_T_INT64: host_minmax_out<long long>(hs, minmax_out_host); break;
    case XM_T_FLOAT32: host_minmax_out<float>(hs, minmax_out_host); break;
    case XM_T_FLOAT64: host_minmax_out<double>(hs, minmax_out_host); break;
    default: return fail(XM_ERR_INVALID, "unknown t_dtype");
  }
  return XM_OK;
}

int xm_shard_minmax_device(xm_handle* h, const void* t, const int16_t* p, size_t n, int t_dtype, void* mm_dev) {
  if (!h || !mm_dev || (n && !t)) return fail(XM_ERR_INVALID, "NULL argument");
  if (t_dtype != XM_T_INT64 && t_dtype != XM_T_FLOAT32 && t_dtype != XM_T_FLOAT64) return fail(XM_ERR_INVALID, "unknown t_dtype");
  XM_ENTER(h);
  Slot& s = h->slots[0];
  int rc;
  if ((rc = rearm_aux(h, s.stream, nullptr, 0))) return rc;
  EventsView ev;
  ev.t = n ? t : (const void*)h->d_lut; ev.p = p; ev.n = n; ev.t_dtype = t_dtype; ev.use_p = p != nullptr;
  ev.x = (const uint16_t*)h->d_lut; ev.y = ev.x;
  launch_minmax(ev, h->aux_st, 2, s.stream);
  switch (t_dtype) {
    case XM_T_INT64: hipLaunchKernelGGL(k_minmax_export<long long>, dim3(1), dim3(64), 0, s.stream, h->aux_st, 2u, mm_dev); break;
    case XM_T_FLOAT32: hipLaunchKernelGGL(k_minmax_export<float>, dim3(1), dim3(64), 0, s.stream, h->aux_st, 2u, mm_dev); break;
    default: hipLaunchKernelGGL(k_minmax_export<double>, dim3(1), dim3(64), 0, s.stream, h->aux_st, 2u, mm_dev);
  }
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

int xm_shard_scatter_device(xm_handle* h, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p, size_t n,
                            int t_dtype, uint64_t idx_offset, const void* frame_mm_dev, uint32_t tag, uint64_t* key_frame) {
  if (!h || !key_frame || !frame_mm_dev) return fail(XM_ERR_INVALID, "NULL argument");
  if (tag == 0 || tag > KEY_MAX_TAG) return fail(XM_ERR_INVALID, "tag must be in [1, 2^19)");
  XM_ENTER(h);
  if (n == 0) return XM_OK;
  if (idx_offset + n >= XM_KEY_MAX_EVENTS) return fail(XM_ERR_TOO_MANY, "global event index exceeds 2^%d", XM_KEY_IDX_BITS);
  EventsView ev;
  ev.x = x; ev.y = y; ev.t = t; ev.p = p; ev.n = n; ev.t_dtype = t_dtype; ev.use_p = p != nullptr;
  int rc = check_events(ev);
  if (rc) return rc;
  if ((rc = launch_scatter(h, ev, h->aux_st, tag, idx_offset, 0, 0, (u64*)key_frame, nullptr, h->slots[0].stream, false,
                           frame_mm_dev)))
    return rc;
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

int xm_shard_clear(xm_handle* h, uint64_t* key_frame) {
  if (!h || !key_frame) return fail(XM_ERR_INVALID, "NULL argument");
  XM_ENTER(h);
  HIP_TRY(hipMemsetAsync(key_frame, 0, h->key_cells * sizeof(u64), h->slots[0].stream));
  return XM_OK;
}

int xm_shard_scatter(xm_handle* h, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p, size_t n,
                     int t_dtype, uint64_t idx_offset, const void* frame_minmax_host, uint32_t tag, uint64_t* key_frame) {
  if (!h || !key_frame || !frame_minmax_host) return fail(XM_ERR_INVALID, "NULL argument");
  if (tag == 0 || tag > KEY_MAX_TAG) return fail(XM_ERR_INVALID, "tag must be in [1, 2^19)");
  XM_ENTER(h);
  if (n == 0) return XM_OK;
  if (idx_offset + n >= XM_KEY_MAX_EVENTS) return fail(XM_ERR_TOO_MANY, "global event index exceeds 2^%d", XM_KEY_IDX_BITS);
  EventsView ev;
  ev.x = x; ev.y = y; ev.t = t; ev.p = p; ev.n = n; ev.t_dtype = t_dtype; ev.use_p = p != nullptr;
  int rc = check_events(ev);
  if (rc) return rc;
  u64 lo, hi;
  switch (t_dtype) {
    case XM_T_INT64: lo = TimeCodec<long long>::enc(((const long long*)frame_minmax_host)[0]);
                     hi = TimeCodec<long long>::enc(((const long long*)frame_minmax_host)[1]); break;
    case XM_T_FLOAT32: lo = TimeCodec<float>::enc(((const float*)frame_minmax_host)[0]);
                       hi = TimeCodec<float>::enc(((const float*)frame_minmax_host)[1]); break;
    case XM_T_FLOAT64: lo = TimeCodec<double>::enc(((const double*)frame_minmax_host)[0]);
                       hi = TimeCodec<double>::enc(((const double*)frame_minmax_host)[1]); break;
    default: return fail(XM_ERR_INVALID, "unknown t_dtype");
  }
  if ((rc = launch_scatter(h, ev, h->aux_st, tag, idx_offset, lo, hi, (u64*)key_frame, nullptr, h->slots[0].stream))) return rc;
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

int xm_shard_finish(xm_handle* h, const uint64_t* key_frame, uint32_t tag, float* depth_out, uint8_t* bgr_out) {
  if (!h || !key_frame) return fail(XM_ERR_INVALID, "NULL argument");
  if (tag == 0 || tag > KEY_MAX_TAG) return fail(XM_ERR_INVALID, "tag must be in [1, 2^19)");
  XM_ENTER(h);
  launch_frame_kernel(h, (const u64*)key_frame, h->aux_st, tag, depth_out, bgr_out, h->slots[0].stream);
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

int xm_shard_decode_u16(xm_handle* h, const uint64_t* key_cells, size_t n_cells, uint32_t tag, uint16_t* disp_out) {
  if (!h || (n_cells && (!key_cells || !disp_out))) return fail(XM_ERR_INVALID, "NULL argument");
  if (tag == 0 || tag > KEY_MAX_TAG) return fail(XM_ERR_INVALID, "tag must be in [1, 2^19)");
  XM_ENTER(h);
  if (n_cells == 0) return XM_OK;
  hipLaunchKernelGGL(k_decode_keys_u16, dim3(grid_for(n_cells, BLOCK)), dim3(BLOCK), 0, h->slots[0].stream, (const u64*)key_cells,
                     (u64)n_cells, tag, disp_out);
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

// Band-sharded finish: the frame kernel for the projector tiles whose patch is centred on a frame column of [col_lo, col_hi)
// only; the caller's depth / BGR buffers keep what they held everywhere else (zero them first, MAX-reduce them over the ranks).
int xm_shard_finish_u16_band(xm_handle* h, const uint16_t* disp_frame, int col_lo, int col_hi, float* depth_out, uint8_t* bgr_out) {
  if (!h || !disp_frame) return fail(XM_ERR_INVALID, "NULL argument");
  if (h->cfg.view != XM_VIEW_PROJECTOR || h->k2_direct) return fail(XM_ERR_INVALID, "the band-sharded finish is the tiled projector-view frame kernel");
  if (col_lo < 0 || col_hi <= col_lo) return fail(XM_ERR_INVALID, "empty column band");
  XM_ENTER(h);
  launch_k2<2>(h, h->slots[0].stream, reinterpret_cast<const u64*>(disp_frame), h->aux_st, 1u, nullptr, depth_out, bgr_out, true, col_lo, col_hi);
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

// Widest patch (frame columns) any projector tile reads, over both tile geometries; -1 when a tile's patch does not fit LDS (such
// a tile reads the frame wherever its map points: no band can be cut for it).  The halo a band-sharded rank needs on either side.
int xm_k2_patch_cols_max(xm_handle* h, int* cols_out) {
  if (!h || !cols_out) return fail(XM_ERR_INVALID, "NULL argument");
  *cols_out = h->k2_patch_cols_max;
  return XM_OK;
}

int xm_shard_finish_u16(xm_handle* h, const uint16_t* disp_frame, float* depth_out, uint8_t* bgr_out) {
  if (!h || !disp_frame) return fail(XM_ERR_INVALID, "NULL argument");
  XM_ENTER(h);
  hipStream_t stream = h->slots[0].stream;
  if (h->cfg.view == XM_VIEW_PROJECTOR) {
    if (h->k2_direct) return fail(XM_ERR_INVALID, "xm_shard_finish_u16 needs the tiled frame kernel (XM_K2_DIRECT is set)");
    launch_k2<2>(h, stream, reinterpret_cast<const u64*>(disp_frame), h->aux_st, 1u, nullptr, depth_out, bgr_out, true);
  } else {
    const u64 px = (u64)h->tb.cam_w * h->tb.cam_h;
    hipLaunchKernelGGL(k_frame_direct_u16, dim3(grid_for(px, BLOCK)), dim3(BLOCK), 0, stream, disp_frame, px, h->tb.dlut, depth_out, bgr_out);
  }
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

// ---- N3: frame event filters ------------------------------------------------------------------------------------
int xm_frame_event_filter(xm_handle* h, int filter, int intended_semantics, const void* eventcd16_in, size_t n,
                          const int16_t* xp_i16, int map_height, int map_width, void* eventcd16_out, size_t* n_out) {
  if (!h || !n_out || (n && !eventcd16_in)) return fail(XM_ERR_INVALID, "NULL argument");
  if (filter < FILTER_FIRST_PER_YT || filter > FILTER_MEAN_PER_XY) return fail(XM_ERR_INVALID, "unknown filter %d", filter);
  if (filter == FILTER_FIRST_PER_YT && n && !xp_i16) return fail(XM_ERR_INVALID, "FirstEventPerYT needs xp_i16");
  if (n >= 0xffffffffull) return fail(XM_ERR_TOO_MANY, "too many events");
  *n_out = 0;
  if (n == 0 || map_height <= 0 || map_width <= 0) return XM_OK;
  XM_ENTER(h);
  Slot& s = h->slots[0];
  const size_t cells = (size_t)map_height * map_width;
  if (cells >= 0x7fffffffull) return fail(XM_ERR_INVALID, "map too large");
  if (!eventcd16_out) return fail(XM_ERR_INVALID, "NULL output");
  const u32 n_blocks = (u32)((cells + SCAN_BLOCK - 1) / SCAN_BLOCK);
  int rc;
  if ((rc = stage_in(s.ev_aos, eventcd16_in, n * 16, s.stream))) return rc;
  if (filter == FILTER_FIRST_PER_YT && (rc = stage_in(s.ev_p, xp_i16, n * 2, s.stream))) return rc;
  // scratch: first[cells] last[cells] pos[cells] sums[n_blocks] total[1]
  if ((rc = s.dbg[0].reserve((3 * cells + n_blocks + 4) * sizeof(u32)))) return rc;
  if ((rc = s.dbg[1].reserve(cells * 16))) return rc;
  u32* first = (u32*)s.dbg[0].p;
  u32* last = first + cells;
  u32* pos = last + cells;
  u32* sums = pos + cells;
  u32* total = sums + n_blocks;
  HIP_TRY(hipMemsetAsync(first, 0xff, cells * sizeof(u32), s.stream));
  HIP_TRY(hipMemsetAsync(last, 0, cells * sizeof(u32), s.stream));
  if ((rc = rearm_aux(h, s.stream, nullptr, 0))) return rc;
  hipLaunchKernelGGL(k_filter_scatter, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, s.stream, (const uint4*)s.ev_aos.p,
                     (const int16_t*)s.ev_p.p, (u64)n, filter == FILTER_FIRST_PER_YT ? 1 : 0, map_height, map_width, first,
                     last, &h->aux_st->cnt[0][0][CNT_OOB]);
  hipLaunchKernelGGL(k_filter_scan_blocks, dim3(n_blocks), dim3(SCAN_BLOCK), 0, s.stream, last, (u32)cells, pos, sums);
  hipLaunchKernelGGL(k_filter_scan_sums, dim3(1), dim3(SCAN_BLOCK), 0, s.stream, sums, n_blocks, total);
  hipLaunchKernelGGL(k_filter_emit, dim3(n_blocks), dim3(SCAN_BLOCK), 0, s.stream, (const uint4*)s.ev_aos.p, first, last, pos,
                     sums, (u32)cells, map_width, filter, intended_semantics, (uint4*)s.dbg[1].p);
  HIP_TRY(hipGetLastError());
  u32 cnt = 0;
  HIP_TRY(hipMemcpyAsync(&cnt, total, sizeof cnt, hipMemcpyDeviceToHost, s.stream));
  if ((rc = read_oob(h, s.stream, "frame event filter"))) return rc;  // synchronises the stream
  if (cnt) HIP_TRY(hipMemcpy(eventcd16_out, s.dbg[1].p, (size_t)cnt * 16, hipMemcpyDeviceToHost));
  *n_out = cnt;
  return XM_OK;
}

// ---- N2: pause detection -----------------------------------------------------------------------------------------
int xm_find_pauses(xm_handle* h, const int64_t* t, const void* eventcd16, size_t n, int mem, int64_t thresh_us,
                   uint32_t* idx_out, size_t idx_capacity, size_t* n_out) {
  if (!h || !n_out || (!t == !eventcd16 && n)) return fail(XM_ERR_INVALID, "give exactly one of t / eventcd16");
  if (n >= 0x7fffffffull) return fail(XM_ERR_TOO_MANY, "too many events");
  *n_out = 0;
  if (n < 2) return XM_OK;
  XM_ENTER(h);
  Slot& s = h->slots[0];
  int rc;
  const long long* d_t = (const long long*)t;
  const uint4* d_aos = (const uint4*)eventcd16;
  if (mem == XM_MEM_HOST) {
    if (t) {
      if ((rc = stage_in(s.ev_t, t, n * 8, s.stream))) return rc;
      d_t = (const long long*)s.ev_t.p;
    } else {
      if ((rc = stage_in(s.ev_aos, eventcd16, n * 16, s.stream))) return rc;
      d_aos = (const uint4*)s.ev_aos.p;
    }
  } else if (mem != XM_MEM_DEVICE) {
    return fail(XM_ERR_INVALID, "mem must be XM_MEM_HOST or XM_MEM_DEVICE");
  }
  const u32 n_blocks = (u32)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
  // scratch: flags[n] pos[n] out[n] sums[n_blocks] total[1]
  if ((rc = s.dbg[0].reserve((3 * n + n_blocks + 4) * sizeof(u32)))) return rc;
  u32* flags = (u32*)s.dbg[0].p;
  u32* pos = flags + n;
  u32* out = pos + n;
  u32* sums = out + n;
  u32* total = sums + n_blocks;
  hipLaunchKernelGGL(k_pause_flags, dim3(n_blocks), dim3(SCAN_BLOCK), 0, s.stream, d_t, d_aos, (u32)n, (long long)thresh_us, flags);
  hipLaunchKernelGGL(k_filter_scan_blocks, dim3(n_blocks), dim3(SCAN_BLOCK), 0, s.stream, flags, (u32)n, pos, sums);
  hipLaunchKernelGGL(k_filter_scan_sums, dim3(1), dim3(SCAN_BLOCK), 0, s.stream, sums, n_blocks, total);
  hipLaunchKernelGGL(k_pause_emit, dim3(n_blocks), dim3(SCAN_BLOCK), 0, s.stream, flags, pos, sums, (u32)n, out);
  HIP_TRY(hipGetLastError());
  u32 cnt = 0;
  HIP_TRY(hipMemcpyAsync(&cnt, total, sizeof cnt, hipMemcpyDeviceToHost, s.stream));
  HIP_TRY(hipStreamSynchronize(s.stream));
  *n_out = cnt;
  const size_t ncopy = cnt < idx_capacity ? cnt : idx_capacity;
  if (ncopy && idx_out) HIP_TRY(hipMemcpy(idx_out, out, ncopy * sizeof(u32), hipMemcpyDeviceToHost));
  return XM_OK;
}

// ---- N2: device-side ingest ----------------------------------------------------------------------------------------
}  // extern "C"

struct xm_ingest {
  xm_handle* h = nullptr;
  xm_ingest_config cfg{};
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;  // H2D of packet k+1 runs beside the kernels of packet k
  hipEvent_t copied_ev[4] = {};       // per staging entry: its H2D has finished (the compute stream waits for it)
  u64 capacity = 0, max_packet = 0;
  double period = 0.0;
  long long act_thresh = 0;
  // device
  uint4* buf[2] = {nullptr, nullptr};
  u32* first_idx = nullptr;
  long long* last_ts = nullptr;
  u32 *keep = nullptr, *pos = nullptr, *sums = nullptr, *total = nullptr;        // packet-sized scan scratch
  u32 *flags = nullptr, *pos2 = nullptr, *pauses = nullptr, *sums2 = nullptr, *n_pauses = nullptr;  // buffer-sized
  IngestState* st = nullptr;
  FrameDesc* desc = nullptr;
  u64* key_frame = nullptr;
  SlotState* slot = nullptr;
  float** d_depth_ring = nullptr;
  uint8_t** d_bgr_ring = nullptr;
  // staging (pinned host -> device), a small ring so that the copy of packet k+1 does not wait for packet k's kernels
  static constexpr int STAGE = 4;
  uint4* h_pkt[STAGE] = {};
  uint4* d_pkt[STAGE] = {};
  hipEvent_t pkt_ev[STAGE] = {};
  bool pkt_used[STAGE] = {};
  int pkt_next = 0;
  // results (pinned host, written by the kernels)
  int ring = 0;
  IngestStatus* h_status = nullptr;
  std::vector<float*> h_depth;
  std::vector<uint8_t*> h_bgr;
  uint64_t next_seq = 0;     // frames delivered through xm_ingest_poll so far
  uint64_t pushed = 0;       // events handed in
  uint64_t pushes = 0;
  // The slot's frame tag advances on the device by one per cut frame (<= one per push) and the host never reads it: the slot is
  // cleared (k_reset_slot: tags back to 0, key frame emptied) before the pushes since the last clear can have brought the tag to
  // KEY_MAX_TAG -- the tag field of the packed keys is 19 bits wide, and at 2^20 the shifted tag would leave the 64-bit key
  uint64_t pushes_since_clear = 0, clear_every = KEY_MAX_TAG - 16;
  // Upper bound of the live part of the device buffer (its real size is known to the device only): grows with every push,
  // shrinks when a delivered frame reports how much was left after its cut.  Sizes the grids of the segmentation / frame kernels.
  uint64_t ub_live = 0;
  std::vector<std::pair<uint64_t, uint64_t>> recent;  // (push number, events) of the pushes a frame may still report on
  uint64_t est_frame_events = 0;
};

namespace {

template <bool DIRECT>
int ingest_launch_frame(xm_ingest* g, u64 n_bound, u64 est_n) {
  xm_handle* h = g->h;
  hipStream_t s = g->stream;
  // K0 over the frame (general path: the cut frame is sorted whenever the camera stream is, but nothing here relies on it)
  {
    unsigned gx = grid_for(n_bound, BLOCK * 4);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL((k_minmax_batch<long long, true, false, 1>), dim3(gx, 1), dim3(BLOCK), 0, s, (const FrameDesc*)g->desc);
  }
  if constexpr (DIRECT) {
    const unsigned gx = grid_for(n_bound, BLOCK);
    if (h->cfg.view == XM_VIEW_PROJECTOR)
      hipLaunchKernelGGL((k_scatter_direct_batch<long long, true, false, 0>), dim3(gx, 1), dim3(BLOCK), 0, s, (const FrameDesc*)g->desc, h->tb, 0);
    else
      hipLaunchKernelGGL((k_scatter_direct_batch<long long, true, false, 1>), dim3(gx, 1), dim3(BLOCK), 0, s, (const FrameDesc*)g->desc, h->tb, 0);
  } else {
    const double max_ev = h->tb.xmap_w > 0 ? (h->w_ts - 1.5) * (double)est_n / (double)h->tb.xmap_w : 0.0;
    unsigned threads = TILE_THREADS;
    while (threads > 1024 / TILE_EPT && (double)(threads * TILE_EPT) > max_ev) threads >>= 1;
    const unsigned gx = grid_for(n_bound, threads * TILE_EPT);
    auto launch = [&](auto view_tag) -> int {
      constexpr int VIEW = decltype(view_tag)::value;
      auto kern = k_scatter_tiled_batch<long long, true, false, VIEW, false>;
      int rc = h->ensure_lds(reinterpret_cast<const void*>(kern), h->k1_lds);
      if (rc) return rc;
      hipLaunchKernelGGL(kern, dim3(gx, 1), dim3(threads), h->k1_lds, s, (const FrameDesc*)g->desc, h->tb, h->w_ts, h->w_x, 0);
      return XM_OK;
    };
    int rc = h->cfg.view == XM_VIEW_PROJECTOR ? launch(std::integral_constant<int, 0>{}) : launch(std::integral_constant<int, 1>{});
    if (rc) return rc;
  }
  if (h->cfg.view == XM_VIEW_PROJECTOR) {
    if (!h->k2_direct) {
      launch_k2_batch<0>(h, s, (const FrameDesc*)g->desc, 1);
    } else {
      return fail(XM_ERR_INVALID, "ingest needs the tiled frame kernel (XM_K2_DIRECT is set)");
    }
  } else {
    const u64 px = (u64)h->tb.cam_w * h->tb.cam_h;
    hipLaunchKernelGGL(k_frame_direct_batch, dim3(grid_for(px, BLOCK), 1), dim3(BLOCK), 0, s, (const FrameDesc*)g->desc, px, h->tb.dlut);
  }
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

}  // namespace

extern "C" {

int xm_ingest_create(xm_handle* h, const xm_ingest_config* cfg, xm_ingest** out) {
  if (!h || !cfg || !out) return fail(XM_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(xm_ingest_config)) return fail(XM_ERR_INVALID, "xm_ingest_config.struct_size mismatch");
  if (cfg->projector_fps <= 0) return fail(XM_ERR_INVALID, "projector_fps must be positive");
  XM_ENTER(h);
  xm_ingest* g = new (std::nothrow) xm_ingest();
  if (!g) return fail(XM_ERR_NOMEM, "out of host memory");
  g->h = h;
  g->cfg = *cfg;
  g->capacity = cfg->capacity_events ? cfg->capacity_events : (1u << 21);
  g->max_packet = cfg->max_packet_events ? cfg->max_packet_events : (1u << 19);
  if (g->capacity >= 0x7fffffffull || g->max_packet * 2 > g->capacity) {
    delete g;
    return fail(XM_ERR_INVALID, "capacity must be < 2^31 events and at least twice max_packet_events");
  }
  g->period = 1e6 / (double)cfg->projector_fps;                       // trigger_finder.py: 1e6 / self.projector_fps (float)
  g->act_thresh = cfg->activity_thresh_us > 0 ? cfg->activity_thresh_us : (long long)(1e6 / cfg->projector_fps);  // pipe:65-68
  if (g->cfg.pause_thresh_us <= 0) g->cfg.pause_thresh_us = 40;       // trigger_finder.py:98
  if (g->cfg.min_events_per_frame <= 0) g->cfg.min_events_per_frame = 1000;  // trigger_finder.py:8
  if (g->cfg.min_events_per_frame < 4) {  // the cut is evs[prev + 2 : next - 2] (trigger_finder.py:172): fewer than 4 events between
    delete g;                             // two pauses would be an empty frame, on which the reference's t.min() raises
    return fail(XM_ERR_INVALID, "min_events_per_frame must be >= 4 (the frame is evs[prev + 2 : next - 2])");
  }
  if (const char* e = getenv("XM_INGEST_CLEAR_EVERY")) g->clear_every = (uint64_t)std::max(1, atoi(e));  // tests: exercise the tag clear
  g->ring = cfg->result_ring > 0 ? cfg->result_ring : 8;
  const size_t cam_px = (size_t)h->tb.cam_w * h->tb.cam_h;
  const size_t px = (size_t)h->out_w * h->out_h;
  const u32 nb_pkt = (u32)((g->max_packet + SCAN_BLOCK - 1) / SCAN_BLOCK), nb_buf = (u32)((g->capacity + SCAN_BLOCK - 1) / SCAN_BLOCK);
#define ING_TRY(expr)                                                                 \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) {                                                           \
      int rc_ = fail(XM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));      \
      xm_ingest_destroy(g);                                                           \
      return rc_;                                                                     \
    }                                                                                 \
  } while (0)
  int lo = 0, hi = 0;
  ING_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
  ING_TRY(hipStreamCreateWithPriority(&g->stream, hipStreamNonBlocking, hi));
  if (!(getenv("XM_INGEST_ONE_STREAM") && getenv("XM_INGEST_ONE_STREAM")[0] == '1')) {
    ING_TRY(hipStreamCreateWithFlags(&g->copy_stream, hipStreamNonBlocking));
    for (auto& e : g->copied_ev) ING_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  for (int i = 0; i < 2; ++i) ING_TRY(hipMalloc((void**)&g->buf[i], g->capacity * 16));
  ING_TRY(hipMalloc((void**)&g->first_idx, cam_px * 4));
  ING_TRY(hipMalloc((void**)&g->last_ts, cam_px * 8));
  {
    std::vector<long long> init(cam_px, ING_NO_TS);
    ING_TRY(hipMemcpy(g->last_ts, init.data(), cam_px * 8, hipMemcpyHostToDevice));
  }
  ING_TRY(hipMalloc((void**)&g->keep, (g->max_packet * 2 + nb_pkt + 8) * 4));
  g->pos = g->keep + g->max_packet;
  g->sums = g->pos + g->max_packet;
  g->total = g->sums + nb_pkt;
  ING_TRY(hipMalloc((void**)&g->flags, (g->capacity * 3 + nb_buf + 8) * 4));
  g->pos2 = g->flags + g->capacity;
  g->pauses = g->pos2 + g->capacity;
  g->sums2 = g->pauses + g->capacity;
  g->n_pauses = g->sums2 + nb_buf;
  ING_TRY(hipMalloc((void**)&g->st, sizeof(IngestState)));
  ING_TRY(hipMemset(g->st, 0, sizeof(IngestState)));
  ING_TRY(hipMalloc((void**)&g->desc, sizeof(FrameDesc)));
  ING_TRY(hipMemset(g->desc, 0, sizeof(FrameDesc)));
  ING_TRY(hipMalloc((void**)&g->key_frame, h->key_cells * sizeof(u64)));
  ING_TRY(hipMalloc((void**)&g->slot, sizeof(SlotState)));
  ING_TRY(hipMemset(g->slot, 0, sizeof(SlotState)));
  hipLaunchKernelGGL(k_reset_slot, dim3(1024), dim3(BLOCK), 0, g->stream, g->slot, g->key_frame, (u64)h->key_cells, (unsigned char*)nullptr);
  ING_TRY(hipGetLastError());
  for (int i = 0; i < xm_ingest::STAGE; ++i) {
    ING_TRY(hipHostMalloc((void**)&g->h_pkt[i], g->max_packet * 16, hipHostMallocDefault));
    ING_TRY(hipMalloc((void**)&g->d_pkt[i], g->max_packet * 16));
    ING_TRY(hipEventCreateWithFlags(&g->pkt_ev[i], hipEventDisableTiming));
  }
  ING_TRY(hipHostMalloc((void**)&g->h_status, sizeof(IngestStatus) * g->ring, hipHostMallocMapped));
  memset(g->h_status, 0, sizeof(IngestStatus) * g->ring);
  g->h_depth.assign(g->ring, nullptr);
  g->h_bgr.assign(g->ring, nullptr);
  std::vector<float*> dd(g->ring, nullptr);
  std::vector<uint8_t*> db(g->ring, nullptr);
  for (int i = 0; i < g->ring; ++i) {
    if (cfg->want_depth) {
      ING_TRY(hipHostMalloc((void**)&g->h_depth[i], px * 4, hipHostMallocMapped));
      ING_TRY(hipHostGetDevicePointer((void**)&dd[i], g->h_depth[i], 0));
    }
    if (cfg->want_bgr) {
      ING_TRY(hipHostMalloc((void**)&g->h_bgr[i], px * 3, hipHostMallocMapped));
      ING_TRY(hipHostGetDevicePointer((void**)&db[i], g->h_bgr[i], 0));
    }
  }
  ING_TRY(hipMalloc((void**)&g->d_depth_ring, sizeof(float*) * g->ring));
  ING_TRY(hipMalloc((void**)&g->d_bgr_ring, sizeof(uint8_t*) * g->ring));
  ING_TRY(hipMemcpy(g->d_depth_ring, dd.data(), sizeof(float*) * g->ring, hipMemcpyHostToDevice));
  ING_TRY(hipMemcpy(g->d_bgr_ring, db.data(), sizeof(uint8_t*) * g->ring, hipMemcpyHostToDevice));
  ING_TRY(hipStreamSynchronize(g->stream));
  g->est_frame_events = cfg->expected_events_per_frame;
#undef ING_TRY
  *out = g;
  return XM_OK;
}

void xm_ingest_destroy(xm_ingest* g) {
  if (!g) return;
  (void)hipSetDevice(g->h->cfg.device);
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  for (int i = 0; i < 2; ++i) if (g->buf[i]) (void)hipFree(g->buf[i]);
  if (g->first_idx) (void)hipFree(g->first_idx);
  if (g->last_ts) (void)hipFree(g->last_ts);
  if (g->keep) (void)hipFree(g->keep);
  if (g->flags) (void)hipFree(g->flags);
  if (g->st) (void)hipFree(g->st);
  if (g->desc) (void)hipFree(g->desc);
  if (g->key_frame) (void)hipFree(g->key_frame);
  if (g->slot) (void)hipFree(g->slot);
  if (g->d_depth_ring) (void)hipFree(g->d_depth_ring);
  if (g->d_bgr_ring) (void)hipFree(g->d_bgr_ring);
  for (int i = 0; i < xm_ingest::STAGE; ++i) {
    if (g->h_pkt[i]) (void)hipHostFree(g->h_pkt[i]);
    if (g->d_pkt[i]) (void)hipFree(g->d_pkt[i]);
    if (g->pkt_ev[i]) (void)hipEventDestroy(g->pkt_ev[i]);
  }
  if (g->h_status) (void)hipHostFree(g->h_status);
  for (auto p : g->h_depth) if (p) (void)hipHostFree(p);
  for (auto p : g->h_bgr) if (p) (void)hipHostFree(p);
  if (g->copy_stream) {
    (void)hipStreamSynchronize(g->copy_stream);
    (void)hipStreamDestroy(g->copy_stream);
  }
  for (auto& e : g->copied_ev) if (e) (void)hipEventDestroy(e);
  if (g->stream) (void)hipStreamDestroy(g->stream);
  delete g;
}

static int ingest_push(xm_ingest* g, const void* eventcd16, size_t n, bool pinned);
int xm_ingest_push(xm_ingest* g, const void* eventcd16, size_t n) { return ingest_push(g, eventcd16, n, false); }
int xm_ingest_push_pinned(xm_ingest* g, const void* eventcd16_pinned, size_t n) { return ingest_push(g, eventcd16_pinned, n, true); }

static int ingest_push(xm_ingest* g, const void* eventcd16, size_t n, bool pinned) {
  if (!g || (n && !eventcd16)) return fail(XM_ERR_INVALID, "NULL argument");
  xm_handle* h = g->h;
  HIP_TRY(hipSetDevice(h->cfg.device));
  if (n > g->max_packet) return fail(XM_ERR_TOO_MANY, "packet of %zu events exceeds max_packet_events %llu", n, (unsigned long long)g->max_packet);
  hipStream_t s = g->stream;
  const int k = g->pkt_next;
  g->pkt_next = (k + 1) % xm_ingest::STAGE;
  const uint4* hp = pinned ? (const uint4*)eventcd16 : g->h_pkt[k];
  if (n) {
    if (g->pkt_used[k]) HIP_TRY(hipEventSynchronize(g->pkt_ev[k]));  // the staging entry's previous packet has been consumed
    if (!pinned) memcpy(g->h_pkt[k], eventcd16, n * 16);  // pageable memory: through the pinned staging ring
    if (g->copy_stream) {  // the copy overlaps the previous packets' kernels; the kernels of this packet wait for it
      HIP_TRY(hipMemcpyAsync(g->d_pkt[k], hp, n * 16, hipMemcpyHostToDevice, g->copy_stream));
      HIP_TRY(hipEventRecord(g->copied_ev[k], g->copy_stream));
      HIP_TRY(hipStreamWaitEvent(s, g->copied_ev[k], 0));
    } else {
      HIP_TRY(hipMemcpyAsync(g->d_pkt[k], hp, n * 16, hipMemcpyHostToDevice, s));
    }
  }
  if (g->pushes_since_clear >= g->clear_every) {  // (stream-ordered behind every frame cut so far)
    hipLaunchKernelGGL(k_reset_slot, dim3(1024), dim3(BLOCK), 0, s, g->slot, g->key_frame, (u64)h->key_cells, (unsigned char*)nullptr);
    HIP_TRY(hipGetLastError());
    g->pushes_since_clear = 0;
  }
  g->pushes_since_clear += 1;
  // room for this packet behind the write cursor (device-side decision; the live part moves to the other buffer)
  hipLaunchKernelGGL(k_ing_compact, dim3(256), dim3(BLOCK), 0, s, g->st, g->buf[0], g->buf[1], g->capacity, (u64)g->max_packet);
  hipLaunchKernelGGL(k_ing_compact_commit, dim3(1), dim3(1), 0, s, g->st, g->capacity, (u64)g->max_packet);
  const int use_pol = g->cfg.use_polarity ? 1 : 0, act = g->cfg.activity_filter ? 1 : 0;
  const int cw = h->tb.cam_w, ch = h->tb.cam_h;
  // sub-packets whose time span (max - min) stays within the activity threshold (see xmaps_ingest.hpp)
  size_t a = 0;
  while (a < n) {
    size_t b = n;
    if (act) {
      long long lo = rec_t_host(hp[a]), hi = lo;
      b = a + 1;
      while (b < n) {
        const long long t = rec_t_host(hp[b]);
        const long long nlo = t < lo ? t : lo, nhi = t > hi ? t : hi;
        if (nhi - nlo > g->act_thresh) break;
        lo = nlo; hi = nhi;
        ++b;
      }
    }
    const u32 m = (u32)(b - a);
    const uint4* dp = g->d_pkt[k] + a;
    const u32 nb = (m + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (act) {
      HIP_TRY(hipMemsetAsync(g->first_idx, 0xff, (size_t)cw * ch * 4, s));
      hipLaunchKernelGGL(k_ing_first, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, s, dp, m, use_pol, cw, ch, g->first_idx);
    }
    hipLaunchKernelGGL(k_ing_mark, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, s, dp, m, use_pol, act, g->act_thresh, cw, ch,
                       (const u32*)g->first_idx, (const long long*)g->last_ts, g->keep);
    hipLaunchKernelGGL(k_filter_scan_blocks, dim3(nb), dim3(SCAN_BLOCK), 0, s, (const u32*)g->keep, m, g->pos, g->sums);
    hipLaunchKernelGGL(k_filter_scan_sums, dim3(1), dim3(SCAN_BLOCK), 0, s, g->sums, nb, g->total);
    hipLaunchKernelGGL(k_ing_append, dim3(nb), dim3(SCAN_BLOCK), 0, s, dp, m, use_pol, cw, ch, (const u32*)g->keep, (const u32*)g->pos,
                       (const u32*)g->sums, (const u32*)g->total, g->st, g->buf[0], g->buf[1], g->capacity, act ? g->last_ts : nullptr);
    hipLaunchKernelGGL(k_ing_commit, dim3(1), dim3(1), 0, s, g->st, (const u32*)g->total, g->capacity);
    a = b;
  }
  if (n) {
    HIP_TRY(hipEventRecord(g->pkt_ev[k], s));
    g->pkt_used[k] = true;
  }
  g->pushed += n;
  g->pushes += 1;
  g->ub_live = std::min<u64>(g->capacity, g->ub_live + n);
  g->recent.emplace_back(g->pushes, (uint64_t)n);
  if (g->recent.size() > 4096) {  // many pushes without a poll: fold the older half into one entry under its LAST push number (a
    uint64_t sum = 0;              // frame that reports an earlier push then counts all of it: the bound stays an upper bound)
    for (size_t i = 0; i < 2048; ++i) sum += g->recent[i].second;
    g->recent[2047] = std::make_pair(g->recent[2047].first, sum);
    g->recent.erase(g->recent.begin(), g->recent.begin() + 2047);
  }
  // segmentation over the live part (its size is known to the device only: the grids cover the host's upper bound)
  const u64 bound64 = g->ub_live;
  const u32 n_bound = (u32)bound64;
  hipLaunchKernelGGL(k_ing_begin, dim3(1), dim3(1), 0, s, g->st, (const uint4*)g->buf[0], (const uint4*)g->buf[1], g->period, g->desc);
  if (n_bound >= 2) {
    const u32 nb = (n_bound + SCAN_BLOCK - 1) / SCAN_BLOCK;
    hipLaunchKernelGGL(k_ing_pause_flags, dim3(nb), dim3(SCAN_BLOCK), 0, s, (const IngestState*)g->st, (const uint4*)g->buf[0],
                       (const uint4*)g->buf[1], (long long)g->cfg.pause_thresh_us, n_bound, g->flags);
    hipLaunchKernelGGL(k_filter_scan_blocks, dim3(nb), dim3(SCAN_BLOCK), 0, s, (const u32*)g->flags, n_bound, g->pos2, g->sums2);
    hipLaunchKernelGGL(k_filter_scan_sums, dim3(1), dim3(SCAN_BLOCK), 0, s, g->sums2, nb, g->n_pauses);
    hipLaunchKernelGGL(k_pause_emit, dim3(nb), dim3(SCAN_BLOCK), 0, s, (const u32*)g->flags, (const u32*)g->pos2, (const u32*)g->sums2,
                       n_bound, g->pauses);
    hipLaunchKernelGGL(k_ing_segment, dim3(1), dim3(BLOCK), 0, s, g->st, (const uint4*)g->buf[0], (const uint4*)g->buf[1],
                       (const u32*)g->pauses, (const u32*)g->n_pauses, g->period, (u32)g->cfg.min_events_per_frame, g->desc,
                       g->key_frame, g->slot, (float* const*)g->d_depth_ring, (uint8_t* const*)g->d_bgr_ring, (u32)g->ring);
    // the frame kernels run on whatever the device cut (FrameDesc in device memory); nothing to do when desc.valid == 0
    const u64 est = g->est_frame_events ? g->est_frame_events : 0;
    int rc = batch_path(h, est) ? ingest_launch_frame<false>(g, bound64, est) : ingest_launch_frame<true>(g, bound64, est);
    if (rc) return rc;
    hipLaunchKernelGGL(k_ing_publish, dim3(1), dim3(64), 0, s, g->st, (const FrameDesc*)g->desc, g->h_status, (u64)g->pushes);
  }
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

int xm_ingest_poll(xm_ingest* g, xm_ingest_frame* out) {
  if (!g || !out) return fail(XM_ERR_INVALID, "NULL argument");
  const int slot = (int)(g->next_seq % (uint64_t)g->ring);
  const IngestStatus* st = g->h_status + slot;
  const uint64_t want = g->next_seq + 1;  // the entry's seq once frame next_seq has been published
  const uint64_t seq = __atomic_load_n(&st->seq, __ATOMIC_ACQUIRE);
  if (seq < want) return 0;  // not there yet
  IngestStatus v;
  memcpy(&v, st, sizeof v);
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  const uint64_t seq2 = __atomic_load_n(&st->seq, __ATOMIC_ACQUIRE);  // did the producer rewrite the entry while it was read?
  const bool lapped = seq > want || seq2 != seq;  // the ring holds a later frame here (or is being rewritten): this one is lost
  memset(out, 0, sizeof *out);
  out->seq = g->next_seq;
  out->lost = lapped ? 1 : 0;
  if (lapped) {
    // Nothing of the entry can be trusted for frame next_seq (no statistics, no images: depth / bgr stay NULL).  The host's bound of
    // the live part is left as it is (an upper bound stays one).  Resume with the oldest frame the ring may still hold intact.
    const uint64_t newest = std::max(seq, seq2);  // >= want + ring - 1
    g->next_seq = std::max<uint64_t>(g->next_seq + 1, newest >= (uint64_t)g->ring ? newest - (uint64_t)g->ring : 0);
    return 1;
  }
  out->n_events = v.n_events;
  out->t_first = v.t_first;
  out->t_last = v.t_last;
  out->n_inliers = v.n_inliers;
  out->n_index_errors = v.n_index_errors;
  out->live_after = v.live_after;
  out->overflow = v.overflow;
  out->depth = g->h_depth[slot];
  out->bgr = g->h_bgr[slot];
  g->est_frame_events = v.n_events;  // the next frames' K1 variant / block size follow the stream's density
  {  // after that frame's cut `live_after` events were left; everything pushed since may have been appended
    uint64_t later = 0;
    size_t keep_from = g->recent.size();
    for (size_t i = g->recent.size(); i-- > 0;) {
      if (g->recent[i].first <= v.push_seq) break;
      later += g->recent[i].second;
      keep_from = i;
    }
    g->recent.erase(g->recent.begin(), g->recent.begin() + keep_from);
    g->ub_live = std::min<uint64_t>(g->capacity, v.live_after + later);
  }
  g->next_seq += 1;
  return 1;
}

int xm_ingest_flush(xm_ingest* g) {
  if (!g) return fail(XM_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(g->h->cfg.device));
  if (g->copy_stream) HIP_TRY(hipStreamSynchronize(g->copy_stream));
  HIP_TRY(hipStreamSynchronize(g->stream));
  return XM_OK;
}

int xm_ingest_reset(xm_ingest* g) {
  if (!g) return fail(XM_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(g->h->cfg.device));
  HIP_TRY(hipStreamSynchronize(g->stream));
  IngestState z;
  HIP_TRY(hipMemcpy(&z, g->st, sizeof z, hipMemcpyDeviceToHost));
  z.buf_start = z.write = 0;  // RobustTriggerFinder.reset(): the buffered events are discarded (trigger_finder.py:116-119)
  HIP_TRY(hipMemcpy(g->st, &z, sizeof z, hipMemcpyHostToDevice));
  g->ub_live = 0;
  g->recent.clear();
  return XM_OK;
}

// ---- N1: X-map construction ----------------------------------------------------------------------------------
int xm_build_x_map(int device, const float* time_map, int height, int width, int x_map_width, int t_px_scale,
                   int x_offset, int num_scanlines, int16_t* x_map_out, float* t_diffs_out) {
  if (!time_map || !x_map_out || height <= 0 || width <= 0 || x_map_width <= 0 || t_px_scale <= 0 || num_scanlines <= 0)
    return fail(XM_ERR_INVALID, "bad argument");
  if (height > 32767 || width + x_offset > 32767) return fail(XM_ERR_INVALID, "indices must fit int16 (x_maps_disparity.py:52-53)");
  if ((size_t)width * sizeof(double) > 150 * 1024) return fail(XM_ERR_INVALID, "time-map row does not fit LDS");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(XM_ERR_HIP, "no HIP device visible");
  HIP_TRY(hipSetDevice(device));
  const size_t n_in = (size_t)height * width, n_out = (size_t)height * x_map_width;
  float* d_in = nullptr;
  int16_t* d_x = nullptr;
  float* d_d = nullptr;
  int rc = XM_OK;
  do {
    hipError_t e;
    if ((e = hipMalloc((void**)&d_in, n_in * 4)) != hipSuccess || (e = hipMalloc((void**)&d_x, n_out * 2)) != hipSuccess ||
        (t_diffs_out && (e = hipMalloc((void**)&d_d, n_out * 4)) != hipSuccess) ||
        (e = hipMemcpy(d_in, time_map, n_in * 4, hipMemcpyHostToDevice)) != hipSuccess) {
      rc = fail(XM_ERR_HIP, "xm_build_x_map: %s", hipGetErrorString(e));
      break;
    }
    const size_t lds = (size_t)width * sizeof(double);
    if (lds > 64 * 1024 &&
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_build_x_map), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds)) != hipSuccess) {
      rc = fail(XM_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
      break;
    }
    hipLaunchKernelGGL(k_build_x_map, dim3(height), dim3(BLOCK), lds, 0, d_in, height, width, x_map_width, t_px_scale,
                       x_offset, 2.0 / (double)num_scanlines, d_x, d_d);
    if ((e = hipGetLastError()) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess ||
        (e = hipMemcpy(x_map_out, d_x, n_out * 2, hipMemcpyDeviceToHost)) != hipSuccess ||
        (t_diffs_out && (e = hipMemcpy(t_diffs_out, d_d, n_out * 4, hipMemcpyDeviceToHost)) != hipSuccess)) {
      rc = fail(XM_ERR_HIP, "xm_build_x_map: %s", hipGetErrorString(e));
      break;
    }
  } while (0);
  if (d_in) (void)hipFree(d_in);
  if (d_x) (void)hipFree(d_x);
  if (d_d) (void)hipFree(d_d);
  return rc;
}

// ---- N4: evaluation metrics -------------------------------------------------------------------------------------
int xm_eval_stats(int device, const float* estimate, const float* groundtruth, int height, int width, int filter,
                  float min_depth, float max_depth, xm_eval_result* out) {
  if (!estimate || !groundtruth || !out || height <= 0 || width <= 0) return fail(XM_ERR_INVALID, "bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(XM_ERR_HIP, "no HIP device visible");
  HIP_TRY(hipSetDevice(device));
  const u64 n = (u64)height * width;
  float *d_e = nullptr, *d_g = nullptr;
  EvalAcc* d_a = nullptr;
  int rc = XM_OK;
  EvalAcc a;
  memset(&a, 0, sizeof a);
  do {
    hipError_t e;
    if ((e = hipMalloc((void**)&d_e, n * 4)) != hipSuccess || (e = hipMalloc((void**)&d_g, n * 4)) != hipSuccess ||
        (e = hipMalloc((void**)&d_a, sizeof(EvalAcc))) != hipSuccess ||
        (e = hipMemcpy(d_e, estimate, n * 4, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(d_g, groundtruth, n * 4, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemset(d_a, 0, sizeof(EvalAcc))) != hipSuccess) {
      rc = fail(XM_ERR_HIP, "xm_eval_stats: %s", hipGetErrorString(e));
      break;
    }
    unsigned grid = grid_for(n, BLOCK * 8);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL((k_eval_stats<1>), dim3(grid), dim3(BLOCK), 0, 0, (const float*)d_e, (const float*)d_g, n, filter, min_depth, max_depth, d_a);
    hipLaunchKernelGGL((k_eval_stats<2>), dim3(grid), dim3(BLOCK), 0, 0, (const float*)d_e, (const float*)d_g, n, filter, min_depth, max_depth, d_a);
    if ((e = hipGetLastError()) != hipSuccess || (e = hipMemcpy(&a, d_a, sizeof a, hipMemcpyDeviceToHost)) != hipSuccess) {
      rc = fail(XM_ERR_HIP, "xm_eval_stats: %s", hipGetErrorString(e));
      break;
    }
  } while (0);
  if (d_e) (void)hipFree(d_e);
  if (d_g) (void)hipFree(d_g);
  if (d_a) (void)hipFree(d_a);
  if (rc) return rc;
  const double hw = (double)n;
  out->margin = 0.01 * a.sum_gt / (double)a.n_gt_pos;
  out->fillrate = ((double)a.n_close - (double)a.n_gt_zero) / (hw - (double)a.n_gt_zero);
  out->rmse = a.n_valid ? std::sqrt(a.sum_sq / (double)a.n_valid) : 0.0;
  out->perc_1 = 100.0 * (double)a.n1 / hw;
  out->perc_5 = 100.0 * (double)a.n5 / hw;
  out->perc_10 = 100.0 * (double)a.n10 / hw;
  out->n_valid = a.n_valid;
  out->n_gt_zero = a.n_gt_zero;
  return XM_OK;
}

// ---- pinned host memory --------------------------------------------------------------------------------------
int xm_host_alloc(xm_handle* h, size_t bytes, void** out) {
  if (!h || !out) return fail(XM_ERR_INVALID, "NULL argument");
  XM_ENTER(h);
  HIP_TRY(hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault));
  return XM_OK;
}
int xm_host_free(xm_handle* h, void* p) {
  if (!h) return fail(XM_ERR_INVALID, "NULL handle");
  XM_ENTER(h);
  if (p) HIP_TRY(hipHostFree(p));
  return XM_OK;
}

// ---- device memory helpers ----------------------------------------------------------------------------------
int xm_dev_alloc(xm_handle* h, size_t bytes, void** out) {
  if (!h || !out) return fail(XM_ERR_INVALID, "NULL argument");
  XM_ENTER(h);
  HIP_TRY(hipMalloc(out, bytes ? bytes : 16));
  return XM_OK;
}
int xm_dev_free(xm_handle* h, void* p) {
  if (!h) return fail(XM_ERR_INVALID, "NULL handle");
  XM_ENTER(h);
  if (p) HIP_TRY(hipFree(p));
  return XM_OK;
}
int xm_dev_upload(xm_handle* h, void* dst_dev, const void* src_host, size_t bytes) {
  if (!h) return fail(XM_ERR_INVALID, "NULL handle");
  XM_ENTER(h);
  if (bytes) HIP_TRY(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
  return XM_OK;
}
int xm_dev_download(xm_handle* h, void* dst_host, const void* src_dev, size_t bytes) {
  if (!h) return fail(XM_ERR_INVALID, "NULL handle");
  XM_ENTER(h);
  if (bytes) HIP_TRY(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
  return XM_OK;
}

}  // extern "C"
