// xm_api_ingest.hpp -- C-ABI: device-side ingest (N2): raw camera packets in, frames cut and processed on the device
// (part of libxmaps_hip.so's host side: included by ../xmaps_hip.hip, one translation unit; see that file for the order)
#pragma once

// ---- N2: device-side ingest ----------------------------------------------------------------------------------------

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
  {  // H2D of a packet on a copy stream beside the kernels of the previous one
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
  // (a memset of device memory may return before it has run and g->stream does not wait for the default stream: k_reset_slot
  //  initialises the extrema slots inside these bytes -- seen once as a first frame with a wrong time normalisation)
  ING_TRY(hipDeviceSynchronize());
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
  ING_TRY(hipDeviceSynchronize());  // (the memsets above ran on the default stream, which the ingest's non-blocking streams do not wait for)
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

// everything behind the packet's arrival in d_pkt[k]: filters, append, segmentation, the frame kernels, publish.  hp = the packet
// in host memory (the activity filter splits it into sub-packets by time stamps there); NULL for a packet decoded on the device
static int ingest_process(xm_ingest* g, int k, size_t n, const uint4* hp);

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
  return ingest_process(g, k, n, hp);
}

static int ingest_process(xm_ingest* g, int k, size_t n, const uint4* hp) {
  xm_handle* h = g->h;
  hipStream_t s = g->stream;
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
    if (act && hp) {
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
  HIP_TRY(hipDeviceSynchronize());  // (default-stream work: the ingest's non-blocking streams do not wait for it)
  g->ub_live = 0;
  g->recent.clear();
  return XM_OK;
}


}  // extern "C"
