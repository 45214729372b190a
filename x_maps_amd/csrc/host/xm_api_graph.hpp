// xm_api_graph.hpp -- C-ABI: batches captured into a hipGraph (BASELINE config 5)
// (part of libxmaps_hip.so's host side: included by ../xmaps_hip.hip, one translation unit; see that file for the order)
#pragma once

extern "C" {

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
  int rc = graph_capture_batched(h, g, x, y, t, p, t_dtype, offsets_host, n_frames, depth_out, bgr_out);
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
  if (h->try_sorted)
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


}  // extern "C"
