// xm_host.hpp -- error reporting, device scratch, slots, launch workers' queues, the handle, launch / profile macros
// (part of libxmaps_hip.so's host side: included by ../xmaps_hip.hip, one translation unit; see that file for the order)
#pragma once

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

// Variant switches for tests and experiments (XM_COLS, XM_K2_PIPE, XM_OWN_W, ...): set through xm_debug_option(), never read from
// the environment -- the library's behaviour does not depend on what the calling process happens to have exported.  Read when a
// handle (or an ingest) is created, except the few per-call measurement switches ("XM_SHARDED_KEYS", "XM_SHARD_PROFILE").
std::mutex g_opt_mu;
std::map<std::string, const std::string*> g_opts;  // name -> its current text, in g_opt_texts
std::deque<std::string> g_opt_texts;               // every text ever set: never erased, so a pointer handed out stays valid for the
                                                   // life of the process whatever another thread sets or removes meanwhile
const char* dbg_opt(const char* name) {
  std::lock_guard<std::mutex> lk(g_opt_mu);
  auto it = g_opts.find(name);
  return it == g_opts.end() ? nullptr : it->second->c_str();
}

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
  int4* d_k2_tiles[3] = {nullptr, nullptr, nullptr};  // [g]: tiles of 16 << g pixels x 16 rows
  u32* d_k2_pix[3] = {nullptr, nullptr, nullptr};
  uint16_t* d_k2_pix16[3] = {nullptr, nullptr, nullptr};  // the pipelined K2's copy: u16, rows padded to k2_pix_stride
  int k2_pix_stride = 0;
  int k2_consec = -1;  // k_frame_proj_pipe<PPT, true>: PPT consecutive pixels per thread; -1 = where it measured faster (PPT = 4), XM_K2_CONSEC=0|1 forces
  int k2_tile_cap[3] = {K2_TILE_MAX, K2_TILE_MAX, K2_TILE_MAX};  // cells of the largest K2 patch (multiple of 8)
  int k2_pipe_g = 1;  // the pipelined kernel's geometry: tiles of 16 << g pixels x 16 rows (2 / 4 pixels per thread)
  int k2_patch_cols_max = 0;  // widest patch of the 16 x 16 / 32 x 16 tiles (-1: some patch does not fit LDS)
  int k2_force_ppt = 0;                             // XM_K2_PPT=1/2: experiments
  // pipelined K2 of the group launches (xmaps_k2pipe.hpp): table entries kept in LDS, CUs of the device, switch (XM_K2_PIPE=0: off)
  int k2_pipe_nlds = 0, n_cus = 256;
  uint64_t k2_pipe_frames = 0;  // frames finished by the pipelined K2 since xm_create (xm_debug_k2_pipe_frames)
  int k2_per_cu_max = 8;        // "XM_K2_PER_CU": the pipelined K2's blocks per CU at most (what it leaves, the next group's K1 can take)
  // "XM_K2_CHAIN": the pipelined K2 launches of different groups (different streams) wait for each other -- one K2 at a time, at
  // k2_per_cu_max blocks per CU, and the following groups' boundary pass and K1 in what it leaves (profiles/r05_own_tiles.md)
  bool k2_chain = false;
  std::mutex k2_chain_mu;
  hipEvent_t k2_chain_ev[16] = {};
  unsigned k2_chain_n = 0;
  bool k2_pipe_force = false;   // XM_K2_PIPE=2 (tests): also for groups too small for the pipeline to matter
  bool k2_pipe = true, k2_pipe_rig_ok = false;  // (rig_ok: every tile's patch fits the pipelined loader, rect_h % 8 == 0)
  ulonglong2* d_zero16 = nullptr;  // 16 zero bytes: what K2 reads instead of a clean key-frame line
  SlotState* d_states = nullptr;  // n_slots + 1 (last = aux state for stage / shard calls)
  SlotState* aux_st = nullptr;
  u64* d_shard_n = nullptr;  // shards on the column tiles: the piece's own event count (device) + its FrameDesc behind it
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
  int cols_lds_pad = 0;  // XM_COLS_LDS_PAD (experiments): bytes of dynamic LDS a column-tile block asks for beyond its carve-up
  int cols_flags = 0;  // COLS_F_ALL_IN_FRAME when no live (row, column) pair of the rig maps outside the frame
  // owner tiles (xmaps_k1own.hpp): the rig's (row, column) -> cell map is not injective (the reference's own calibration), but
  // every cell's columns lie within a few (<= 7) columns of its first one: cols_ok with own_mode set; tile widths and halos per plan
  bool own_mode = false;
  // up to three plans (own_setup), by falling tile width: ownership per 8-row group on tiles of 20 and of 16 columns -- frames
  // whose tiles fit one event pass of a block --, then ownership per row on tiles of 8 -- denser frames, and rigs whose slant
  // rules the first two out.  All write the same frame (every cell a pair maps to, every frame), so consecutive frames of a slot
  // may take different plans.
  static constexpr int OWN_PLANS = 3;
  struct OwnSet {
    bool ok = false, all_in = false;
    int w = 0, halo = 0, extras = 0;
    int r_lo = 0, hr = 0, hrp = 0, rp = 0, grouped = 0, nxs_max = 0, extra_max = 0;
    uint16_t* d_xmap_own = nullptr;
    uint16_t* d_xmap_extra = nullptr;
    int4* d_tiles = nullptr;
    u32* d_bm = nullptr;
    u32* d_extra_cells = nullptr;
  } own[OWN_PLANS];
  int own_ept_forced = 0;  // "XM_OWN_EPT" (experiments): 4 = four events per thread where a tile then fits one pass
  // XM_FLAG_ADAPTIVE_BATCH: asynchronous device-pointer frames are submitted as GROUPS (multi-frame launches) whenever the GPU
  // is still busy with earlier ones: a frame is launched at once when no group is in flight (an idle GPU -- the 60 Hz live
  // case -- never waits), otherwise it joins the pending list, which goes out as one group with the first call that finds the
  // GPU idle, when it holds ab_max = n_slots / 4 frames, or at the next synchronising call
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


}  // namespace
