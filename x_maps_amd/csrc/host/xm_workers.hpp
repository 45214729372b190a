// xm_workers.hpp -- slot rotation, the verified-shortcut verdict and redo (resolve_prev), launch workers, the single-frame entry (process_common)
// (part of libxmaps_hip.so's host side: included by ../xmaps_hip.hip, one translation unit; see that file for the order)
#pragma once

namespace {

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
    // One group = one kernel template (AoS or SoA, the time stamps' type, polarity column or not), chosen from its first frame:
    // a frame of another layout closes the group that is pending and opens its own.
    if (!h->pending.empty()) {
      const EventsView& e0 = h->pending.front().ev;
      if ((e0.aos != nullptr) != (ev.aos != nullptr) || e0.t_dtype != ev.t_dtype || e0.use_p != ev.use_p || (e0.p != nullptr) != (ev.p != nullptr))
        if ((rc = flush_pending(h))) return rc;
    }
    h->pending.push_back(xm_handle::Deferred{ev, depth_out, bgr_out});
    int in_flight = 0;
    for (hipEvent_t e : h->ab_inflight)
      if (e && hipEventQuery(e) == hipErrorNotReady) in_flight += 1;
    (void)hipGetLastError();
    // Launch at once when the GPU is idle (the 60 Hz live pipe: latency), else hold the frame back until a full group is pending: a
    // caller that produces frames faster than the GPU takes them gets groups of ab_max from the second launch on.  ("in_flight < 3"
    // until the end of round 3: every group's submission costs the CALLING thread ~30 us, so a run that started with small groups
    // could stay there -- the thread had no time left to get ahead of the GPU: 37 instead of 140 Gev/s on some boxes.)
    if ((int)h->pending.size() >= h->ab_max || in_flight == 0) return flush_pending(h);
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
