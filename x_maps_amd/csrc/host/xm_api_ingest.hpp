// xm_api_ingest.hpp -- C-ABI: device-side ingest (N2): raw camera packets in, frames cut and processed on the device
// (part of libxmaps_hip.so's host side: included by ../xmaps_hip.hip, one translation unit; see that file for the order)
#pragma once

// ---- N2: device-side ingest ----------------------------------------------------------------------------------------

struct xm_ingest {
  xm_handle* h = nullptr;
  xm_ingest_config cfg{};
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;  // H2D of packet k+1 runs beside the kernels of packet k
  u64 capacity = 0, max_packet = 0;   // capacity: a power of two (the request rounded up)
  double period = 0.0;
  long long act_thresh = 0;
  // device
  IngestDev dev{};                    // what every ingest kernel gets by value (ring, pause ring, state, descriptor, result ring)
  u32* first_idx = nullptr;           // activity filter: first event index of the sub-packet per pixel
  u32* keep = nullptr;                // activity filter: keep flags of the (sub-)packet
  float** d_depth_ring = nullptr;
  uint8_t** d_bgr_ring = nullptr;
  // staging (pinned host -> device), a small ring so that the copy of packet k+1 does not wait for packet k's kernels
  static constexpr int STAGE = 16;
  uint4* h_pkt[STAGE] = {};
  uint4* d_pkt[STAGE] = {};
  hipEvent_t copied_ev[STAGE] = {};   // per staging entry: its H2D has finished (the compute stream waits for it)
  uint64_t pkt_push[STAGE] = {};      // number of the push that used the entry last (0: never): free once that push has run
  int pkt_next = 0;
  // results (pinned host, written by the kernels)
  int ring = 0;
  IngestStatus* h_status = nullptr;
  u64* h_pushes_done = nullptr;       // pinned: number of the last push whose kernels have run (written by k_ing_publish)
  std::vector<float*> h_depth;
  std::vector<uint8_t*> h_bgr;
  uint64_t next_seq = 0;     // frames delivered through xm_ingest_poll so far
  uint64_t pushed = 0;       // events handed in
  std::atomic<uint64_t> pushes{0};  // packets whose launches have been issued (by the launch thread, if there is one)
  // The slot's frame tag advances on the device by one per cut frame (<= one per push) and the host never reads it: the slot is
  // cleared (k_reset_slot: tags back to 0, key frame emptied) before the pushes since the last clear can have brought the tag to
  // KEY_MAX_TAG -- the tag field of the packed keys is 19 bits wide, and at 2^20 the shifted tag would leave the 64-bit key
  uint64_t pushes_since_clear = 0, clear_every = KEY_MAX_TAG - 16;
  // Upper bound of the live part of the device buffer (its real size is known to the device only): grows with every push,
  // shrinks when a delivered frame reports how much was left after its cut.  Sizes the grids of the frame kernels.
  uint64_t ub_live = 0;
  std::vector<std::pair<uint64_t, uint64_t>> recent;  // (push number, events) of the pushes a frame may still report on
  uint64_t est_frame_events = 0;
  // host time spent inside xm_ingest_push* (what the calling thread pays per packet), for xm_ingest_host_stats
  double push_host_s = 0.0;
  uint64_t push_calls = 0, stage_waits = 0;
  // The launch thread (default; XM_INGEST_NO_LAUNCH_THREAD turns it off): xm_ingest_push* stages the packet and posts a job, the
  // thread issues the copy and the launches (~10 API calls, 35 us per packet) -- the caller pays ~2 us for a pinned packet.
  // `mu` guards what both sides touch: ub_live, recent, est_frame_events (xm_ingest_poll tightens them, the launches read them).
  struct Job {
    int kind = 0;                       // 0: records, 1: EVT 3.0 words (pinned), 2: stop
    int k = 0;                          // staging entry
    size_t n = 0;                       // events (records) / words
    const void* host = nullptr;         // pinned source (the staging entry or the caller's pinned memory)
    xm_evt3* dec = nullptr;
    uint64_t push_no = 0;
  };
  static constexpr unsigned QCAP = 64;
  Job queue[QCAP];
  std::atomic<unsigned long long> q_head{0}, q_tail{0}, q_done{0};
  std::atomic<int> q_error{0};
  std::string q_error_text;
  std::mutex q_mu, mu;
  std::condition_variable q_cv;
  std::atomic<bool> q_sleeping{false};
  std::thread th;
  bool threaded = false;
  uint64_t posted = 0;                  // pushes accepted so far (the caller's count; `pushes` = issued, the launch side's)
};

namespace {

template <bool DIRECT>
int ingest_launch_frame(xm_ingest* g, u64 n_bound, u64 est_n) {
  xm_handle* h = g->h;
  hipStream_t s = g->stream;
  const FrameDesc* desc = g->dev.desc;
  // K0 over the frame (general path: the cut frame is sorted whenever the camera stream is, but nothing here relies on it)
  {
    unsigned gx = grid_for(n_bound, BLOCK * 4);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL((k_minmax_batch<long long, true, false, 1>), dim3(gx, 1), dim3(BLOCK), 0, s, desc);
  }
  if constexpr (DIRECT) {
    const unsigned gx = grid_for(n_bound, BLOCK);
    if (h->cfg.view == XM_VIEW_PROJECTOR)
      hipLaunchKernelGGL((k_scatter_direct_batch<long long, true, false, 0>), dim3(gx, 1), dim3(BLOCK), 0, s, desc, h->tb, 0);
    else
      hipLaunchKernelGGL((k_scatter_direct_batch<long long, true, false, 1>), dim3(gx, 1), dim3(BLOCK), 0, s, desc, h->tb, 0);
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
      hipLaunchKernelGGL(kern, dim3(gx, 1), dim3(threads), h->k1_lds, s, desc, h->tb, h->w_ts, h->w_x, 0);
      return XM_OK;
    };
    int rc = h->cfg.view == XM_VIEW_PROJECTOR ? launch(std::integral_constant<int, 0>{}) : launch(std::integral_constant<int, 1>{});
    if (rc) return rc;
  }
  if (h->cfg.view == XM_VIEW_PROJECTOR) {
    if (!h->k2_direct) {
      launch_k2_batch<0>(h, s, desc, 1);
    } else {
      return fail(XM_ERR_INVALID, "ingest needs the tiled frame kernel (XM_K2_DIRECT is set)");
    }
  } else {
    const u64 px = (u64)h->tb.cam_w * h->tb.cam_h;
    hipLaunchKernelGGL(k_frame_direct_batch, dim3(grid_for(px, BLOCK), 1), dim3(BLOCK), 0, s, desc, px, h->tb.dlut);
  }
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

inline double ingest_now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// the staging entry's previous packet has been consumed once the push that used it has run (k_ing_publish reports the number of
// a finished push in pinned memory -- every fourth one and every one that cut a frame: no API call, no event)
int ingest_wait_entry(xm_ingest* g, int k) {
  const uint64_t need = g->pkt_push[k];
  if (!need || __atomic_load_n(g->h_pushes_done, __ATOMIC_ACQUIRE) >= need) return XM_OK;
  g->stage_waits += 1;
  while (g->pushes.load(std::memory_order_acquire) < std::min<uint64_t>(g->posted, (need + 3) & ~3ull)) __builtin_ia32_pause();  // (still queued)
  unsigned spins = 0;
  while (__atomic_load_n(g->h_pushes_done, __ATOMIC_ACQUIRE) < need) {
    __builtin_ia32_pause();
    if ((++spins & 0xfff) == 0) {  // make sure the runtime has handed the launches to the GPU; stop once the stream is empty
      hipError_t q = hipStreamQuery(g->stream);
      if (q == hipSuccess && g->pushes.load(std::memory_order_acquire) >= g->posted) break;
      if (q != hipSuccess && q != hipErrorNotReady) HIP_TRY(q);
    }
  }
  return XM_OK;
}

int ingest_process(xm_ingest* g, int k, size_t n, const uint4* hp, const u32* n_dev = nullptr);

// the launches of one packet of records: H2D on the copy stream (beside the previous packets' kernels), then everything else
int ingest_issue_records(xm_ingest* g, int k, size_t n, const uint4* hp) {
  if (n) {
    HIP_TRY(hipMemcpyAsync(g->d_pkt[k], hp, n * 16, hipMemcpyHostToDevice, g->copy_stream));
    HIP_TRY(hipEventRecord(g->copied_ev[k], g->copy_stream));
    HIP_TRY(hipStreamWaitEvent(g->stream, g->copied_ev[k], 0));
  }
  return ingest_process(g, k, n, hp);
}

int evt3_enqueue(xm_evt3* d, const uint16_t* words_host, size_t n_words, bool pinned, uint4* out, size_t out_cap, hipStream_t stream);
int ingest_issue_evt3(xm_ingest* g, xm_evt3* d, int k, const uint16_t* words, size_t n_words, bool pinned);

void ingest_thread_main(xm_ingest* g) {
  (void)hipSetDevice(g->h->cfg.device);
  for (;;) {
    unsigned long long t = g->q_tail.load(std::memory_order_relaxed);
    if (t == g->q_head.load(std::memory_order_acquire)) {  // empty: spin a little, then sleep
      bool got = false;
      for (int i = 0; i < 20000 && !got; ++i) {
        __builtin_ia32_pause();
        got = t != g->q_head.load(std::memory_order_acquire);
      }
      if (!got) {
        std::unique_lock<std::mutex> lk(g->q_mu);
        g->q_sleeping.store(true, std::memory_order_seq_cst);
        g->q_cv.wait(lk, [&] { return t != g->q_head.load(std::memory_order_acquire); });
        g->q_sleeping.store(false, std::memory_order_relaxed);
      }
    }
    const xm_ingest::Job j = g->queue[t % xm_ingest::QCAP];
    g->q_tail.store(t + 1, std::memory_order_release);
    if (j.kind == 2) {
      g->q_done.store(t + 1, std::memory_order_release);
      return;
    }
    const int rc = j.kind == 0 ? ingest_issue_records(g, j.k, j.n, (const uint4*)j.host)
                               : ingest_issue_evt3(g, j.dec, j.k, (const uint16_t*)j.host, j.n, true);
    if (rc != XM_OK && g->q_error.load(std::memory_order_relaxed) == 0) {
      g->q_error_text = g_err;  // thread-local text of this thread
      g->q_error.store(rc, std::memory_order_release);
    }
    g->q_done.store(t + 1, std::memory_order_release);
  }
}

void ingest_post(xm_ingest* g, const xm_ingest::Job& j) {
  const unsigned long long hd = g->q_head.load(std::memory_order_relaxed);
  while (hd - g->q_tail.load(std::memory_order_acquire) >= xm_ingest::QCAP) __builtin_ia32_pause();  // queue full: back-pressure
  g->queue[hd % xm_ingest::QCAP] = j;
  g->q_head.store(hd + 1, std::memory_order_seq_cst);
  if (g->q_sleeping.load(std::memory_order_seq_cst)) {
    std::lock_guard<std::mutex> lk(g->q_mu);
    g->q_cv.notify_one();
  }
}

// wait until the launch thread has issued everything posted so far (the GPU may still be running it); reports a failed job
int ingest_drain(xm_ingest* g) {
  if (g->threaded) {
    const unsigned long long hd = g->q_head.load(std::memory_order_acquire);
    while (g->q_done.load(std::memory_order_acquire) < hd) __builtin_ia32_pause();
  }
  const int e = g->q_error.load(std::memory_order_acquire);
  if (e) {
    g->q_error.store(0, std::memory_order_release);
    return fail(e, "%s (reported by the ingest's launch thread)", g->q_error_text.c_str());
  }
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
  const u64 want_cap = cfg->capacity_events ? cfg->capacity_events : (1u << 21);
  g->capacity = 1;
  while (g->capacity < want_cap) g->capacity <<= 1;  // the ring is indexed by (absolute stream index) & (capacity - 1)
  g->max_packet = cfg->max_packet_events ? cfg->max_packet_events : (1u << 19);
  if (g->capacity >= 0x7fffffffull || g->max_packet * 2 > g->capacity || g->max_packet > (u64)ING_MAX_BLOCKS * ING_EPB) {
    delete g;
    return fail(XM_ERR_INVALID, "capacity must be < 2^31 events and at least twice max_packet_events (itself at most %llu)",
                (unsigned long long)ING_MAX_BLOCKS * ING_EPB);
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
  ING_TRY(hipStreamCreateWithFlags(&g->copy_stream, hipStreamNonBlocking));  // H2D of a packet beside the kernels of the previous one
  for (auto& e : g->copied_ev) ING_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  IngestDev& d = g->dev;
  d.cap = g->capacity;
  d.max_packet = g->max_packet;
  d.mirror = g->capacity / 2;  // frames of up to half the ring are contiguous wherever they start
  d.pcap = g->capacity * 2;    // (a pause per live event + the stale head the trigger finder has not skipped yet)
  ING_TRY(hipMalloc((void**)&d.buf, (d.cap + d.mirror) * 16));
  ING_TRY(hipMalloc((void**)&d.pring, d.pcap * 8));
  ING_TRY(hipMalloc((void**)&d.blk, sizeof(IngBlk) * ING_MAX_BLOCKS));
  if (cfg->activity_filter) {
    ING_TRY(hipMalloc((void**)&g->first_idx, cam_px * 4));
    ING_TRY(hipMalloc((void**)&d.last_ts, cam_px * 8));
    std::vector<long long> init(cam_px, ING_NO_TS);
    ING_TRY(hipMemcpy(d.last_ts, init.data(), cam_px * 8, hipMemcpyHostToDevice));
    ING_TRY(hipMalloc((void**)&g->keep, g->max_packet * 4));
  }
  d.cam_w = h->tb.cam_w;
  d.cam_h = h->tb.cam_h;
  d.pause_thresh = g->cfg.pause_thresh_us;
  d.period = g->period;
  d.min_events = (u32)g->cfg.min_events_per_frame;
  d.ring = (u32)g->ring;
  ING_TRY(hipMalloc((void**)&d.st, sizeof(IngestState)));
  ING_TRY(hipMemset(d.st, 0, sizeof(IngestState)));
  ING_TRY(hipMalloc((void**)&d.desc, sizeof(FrameDesc)));
  ING_TRY(hipMemset(d.desc, 0, sizeof(FrameDesc)));
  ING_TRY(hipMalloc((void**)&d.key_frame, h->key_cells * sizeof(u64)));
  ING_TRY(hipMalloc((void**)&d.slot, sizeof(SlotState)));
  ING_TRY(hipMemset(d.slot, 0, sizeof(SlotState)));
  // (a memset of device memory may return before it has run and g->stream does not wait for the default stream: k_reset_slot
  //  initialises the extrema slots inside these bytes -- seen once as a first frame with a wrong time normalisation)
  ING_TRY(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_reset_slot, dim3(1024), dim3(BLOCK), 0, g->stream, d.slot, d.key_frame, (u64)h->key_cells, (unsigned char*)nullptr);
  ING_TRY(hipGetLastError());
  for (int i = 0; i < xm_ingest::STAGE; ++i) {
    ING_TRY(hipHostMalloc((void**)&g->h_pkt[i], g->max_packet * 16, hipHostMallocDefault));
    ING_TRY(hipMalloc((void**)&g->d_pkt[i], g->max_packet * 16));
  }
  ING_TRY(hipHostMalloc((void**)&g->h_status, sizeof(IngestStatus) * g->ring, hipHostMallocMapped));
  memset(g->h_status, 0, sizeof(IngestStatus) * g->ring);
  ING_TRY(hipHostMalloc((void**)&g->h_pushes_done, 64, hipHostMallocMapped));
  memset(g->h_pushes_done, 0, 64);
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
  d.depth_ring = g->d_depth_ring;
  d.bgr_ring = g->d_bgr_ring;
  ING_TRY(hipDeviceSynchronize());  // (the memsets above ran on the default stream, which the ingest's non-blocking streams do not wait for)
  g->est_frame_events = cfg->expected_events_per_frame;
#undef ING_TRY
  if (!(cfg->flags & XM_INGEST_NO_LAUNCH_THREAD)) {
    g->threaded = true;
    g->th = std::thread(ingest_thread_main, g);
  }
  *out = g;
  return XM_OK;
}

void xm_ingest_destroy(xm_ingest* g) {
  if (!g) return;
  (void)hipSetDevice(g->h->cfg.device);
  if (g->threaded) {
    xm_ingest::Job stop;
    stop.kind = 2;
    ingest_post(g, stop);
    if (g->th.joinable()) g->th.join();
    g->threaded = false;
  }
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  IngestDev& d = g->dev;
  if (d.buf) (void)hipFree(d.buf);
  if (d.pring) (void)hipFree(d.pring);
  if (d.blk) (void)hipFree(d.blk);
  if (d.last_ts) (void)hipFree(d.last_ts);
  if (d.st) (void)hipFree(d.st);
  if (d.desc) (void)hipFree(d.desc);
  if (d.key_frame) (void)hipFree(d.key_frame);
  if (d.slot) (void)hipFree(d.slot);
  if (g->first_idx) (void)hipFree(g->first_idx);
  if (g->keep) (void)hipFree(g->keep);
  if (g->d_depth_ring) (void)hipFree(g->d_depth_ring);
  if (g->d_bgr_ring) (void)hipFree(g->d_bgr_ring);
  for (int i = 0; i < xm_ingest::STAGE; ++i) {
    if (g->h_pkt[i]) (void)hipHostFree(g->h_pkt[i]);
    if (g->d_pkt[i]) (void)hipFree(g->d_pkt[i]);
  }
  if (g->h_status) (void)hipHostFree(g->h_status);
  if (g->h_pushes_done) (void)hipHostFree(g->h_pushes_done);
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
  const double c0 = ingest_now();
  xm_handle* h = g->h;
  if (n > g->max_packet) return fail(XM_ERR_TOO_MANY, "packet of %zu events exceeds max_packet_events %llu", n, (unsigned long long)g->max_packet);
  if (const int e = g->q_error.load(std::memory_order_acquire)) {  // an earlier packet's launches failed
    g->q_error.store(0, std::memory_order_release);
    return fail(e, "%s (reported by the ingest's launch thread)", g->q_error_text.c_str());
  }
  if (!g->threaded) HIP_TRY(hipSetDevice(h->cfg.device));
  const int k = g->pkt_next;
  g->pkt_next = (k + 1) % xm_ingest::STAGE;
  const uint4* hp = pinned ? (const uint4*)eventcd16 : g->h_pkt[k];
  int rc = XM_OK;
  if (n) {
    if ((rc = ingest_wait_entry(g, k))) return rc;
    if (!pinned) memcpy(g->h_pkt[k], eventcd16, n * 16);  // pageable memory: through the pinned staging ring
  }
  g->posted += 1;
  g->pkt_push[k] = g->posted;
  if (g->threaded) {
    xm_ingest::Job j;
    j.kind = 0; j.k = k; j.n = n; j.host = hp; j.push_no = g->posted;
    ingest_post(g, j);
  } else {
    rc = ingest_issue_records(g, k, n, hp);
  }
  g->push_host_s += ingest_now() - c0;
  g->push_calls += 1;
  return rc;
}

}  // extern "C"

namespace {

int ingest_process(xm_ingest* g, int k, size_t n, const uint4* hp, const u32* n_dev) {
  xm_handle* h = g->h;
  hipStream_t s = g->stream;
  if (g->pushes_since_clear >= g->clear_every) {  // (stream-ordered behind every frame cut so far)
    hipLaunchKernelGGL(k_reset_slot, dim3(1024), dim3(BLOCK), 0, s, g->dev.slot, g->dev.key_frame, (u64)h->key_cells, (unsigned char*)nullptr);
    HIP_TRY(hipGetLastError());
    g->pushes_since_clear = 0;
  }
  g->pushes_since_clear += 1;
  const int act = g->cfg.activity_filter && hp ? 1 : 0;
  const int cw = h->tb.cam_w, ch = h->tb.cam_h;
  IngestPush p{};
  p.flags = g->cfg.use_polarity ? ING_F_POLARITY : 0u;
  const auto launch3 = [&](const IngestPush& pp, u32 bound) {  // count, append (blocks of the packet), commit / segment (one block)
    const unsigned nb = (bound + ING_EPB - 1) / ING_EPB;
    if (nb) {
      hipLaunchKernelGGL(k_ing_count, dim3(nb), dim3(ING_THREADS), 0, s, g->dev, pp);
      hipLaunchKernelGGL(k_ing_append, dim3(nb), dim3(ING_THREADS), 0, s, g->dev, pp);
    }
    hipLaunchKernelGGL(k_ing_segment, dim3(1), dim3(ING_THREADS), 0, s, g->dev, pp);
  };
  if (!act) {
    p.src = g->d_pkt[k];
    p.n = (u32)n;
    p.n_dev = n_dev;
    p.flags |= ING_F_SEGMENT;
    launch3(p, (u32)n);
  } else {
    // sub-packets whose time span (max - min) stays within the activity threshold (see xmaps_ingest.hpp); the trigger finder
    // runs once, behind the last one
    size_t a = 0;
    if (n == 0) {
      p.flags |= ING_F_SEGMENT;
      launch3(p, 0);
    }
    while (a < n) {
      long long lo = rec_t_host(hp[a]), hi = lo;
      size_t b = a + 1;
      while (b < n) {
        const long long t = rec_t_host(hp[b]);
        const long long nlo = t < lo ? t : lo, nhi = t > hi ? t : hi;
        if (nhi - nlo > g->act_thresh) break;
        lo = nlo; hi = nhi;
        ++b;
      }
      const u32 m = (u32)(b - a);
      const uint4* dp = g->d_pkt[k] + a;
      HIP_TRY(hipMemsetAsync(g->first_idx, 0xff, (size_t)cw * ch * 4, s));
      hipLaunchKernelGGL(k_ing_first, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, s, dp, m, (int)(p.flags & ING_F_POLARITY), cw, ch, g->first_idx);
      hipLaunchKernelGGL(k_ing_mark, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, s, dp, m, (int)(p.flags & ING_F_POLARITY), 1, g->act_thresh, cw, ch,
                         (const u32*)g->first_idx, (const long long*)g->dev.last_ts, g->keep);
      IngestPush sp = p;
      sp.src = dp;
      sp.keep = g->keep;
      sp.n = m;
      if (b == n) sp.flags |= ING_F_SEGMENT;
      launch3(sp, m);
      a = b;
    }
  }
  g->pushed += n;
  const uint64_t push_no = g->pushes.load(std::memory_order_relaxed) + 1;
  u64 bound64, est;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->ub_live = std::min<u64>(g->capacity, g->ub_live + n);
    g->recent.emplace_back(push_no, (uint64_t)n);
    if (g->recent.size() > 4096) {  // many pushes without a poll: fold the older half into one entry under its LAST push number (a
      uint64_t sum = 0;              // frame that reports an earlier push then counts all of it: the bound stays an upper bound)
      for (size_t i = 0; i < 2048; ++i) sum += g->recent[i].second;
      g->recent[2047] = std::make_pair(g->recent[2047].first, sum);
      g->recent.erase(g->recent.begin(), g->recent.begin() + 2047);
    }
    bound64 = std::min<u64>(g->ub_live, g->dev.mirror);
    est = g->est_frame_events;
  }
  // the frame kernels run on whatever the device cut (FrameDesc in device memory; its size is known to the device only: the
  // grids cover the host's upper bound of the live part); nothing to do when desc.valid == 0
  if (bound64 >= 2) {
    int rc = batch_path(h, est) ? ingest_launch_frame<false>(g, bound64, est) : ingest_launch_frame<true>(g, bound64, est);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_ing_publish, dim3(1), dim3(64), 0, s, g->dev.st, (const FrameDesc*)g->dev.desc, g->h_status, (u64)push_no, g->h_pushes_done);
  HIP_TRY(hipGetLastError());
  g->pushes.store(push_no, std::memory_order_release);
  return XM_OK;
}

}  // namespace

extern "C" {

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
  {  // after that frame's cut `live_after` events were left; everything pushed since may have been appended
    std::lock_guard<std::mutex> lk(g->mu);
    g->est_frame_events = v.n_events;  // the next frames' K1 variant / block size follow the stream's density
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
  int rc = ingest_drain(g);
  if (rc) return rc;
  if (g->copy_stream) HIP_TRY(hipStreamSynchronize(g->copy_stream));
  HIP_TRY(hipStreamSynchronize(g->stream));
  return XM_OK;
}

int xm_ingest_reset(xm_ingest* g) {
  if (!g) return fail(XM_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(g->h->cfg.device));
  int rc = ingest_drain(g);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(g->stream));
  // RobustTriggerFinder.reset(): the buffered events are discarded (trigger_finder.py:116-119).  The stream indices start over
  // (nothing live refers to the old ones); the frame / push counters and the sticky overflow count go on.
  IngestState z;
  HIP_TRY(hipMemcpy(&z, g->dev.st, sizeof z, hipMemcpyDeviceToHost));
  z.start_abs = z.write_abs = z.p_head = z.p_tail = 0;
  z.last_t = 0;
  z.span_ok = 0;
  HIP_TRY(hipMemcpy(g->dev.st, &z, sizeof z, hipMemcpyHostToDevice));
  HIP_TRY(hipDeviceSynchronize());  // (default-stream work: the ingest's non-blocking streams do not wait for it)
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->ub_live = 0;
    g->recent.clear();
  }
  return XM_OK;
}

int xm_ingest_host_stats(xm_ingest* g, uint64_t* pushes, double* host_seconds_in_push, uint64_t* staging_waits) {
  if (!g) return fail(XM_ERR_INVALID, "NULL argument");
  if (pushes) *pushes = g->push_calls;
  if (host_seconds_in_push) *host_seconds_in_push = g->push_host_s;
  if (staging_waits) *staging_waits = g->stage_waits;
  return XM_OK;
}

}  // extern "C"
