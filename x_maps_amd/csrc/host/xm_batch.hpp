// xm_batch.hpp -- multi-frame launches: one K0 / K1 / K2 launch each for a group of frames (enqueue_batch), frame statistics, host staging
// (part of libxmaps_hip.so's host side: included by ../xmaps_hip.hip, one translation unit; see that file for the order)
#pragma once

namespace {

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
      if (h->own_mode) {  // owner tiles (the rig's X-map is not injective): two boundaries per tile (its first column, the end of its halo), tiles of the frame's plan (own_set): cols_w + halo columns
        const int ept = own_ept(h, n_mean, cols_w, !AOS && vec16);
        auto kern = ept == 4 ? k_scatter_own_batch<AOS, false, 4> : k_scatter_own_batch<AOS, false>;
        if constexpr (!AOS) {
          if (vec16) kern = k_scatter_own_batch<false, true>;
        }
        const xm_handle::OwnSet& os = own_set(h, cols_w);
        DevTables tbo = h->tb;
        own_apply(os, tbo);
        const size_t lds = own_lds_bytes(os);
        rc = h->ensure_lds(reinterpret_cast<const void*>(kern), lds);
        if (rc) return rc;
        prof_slot(0);
        XM_LAUNCH(k_cols_bounds_batch<AOS>, dim3(grid_for(2 * grid_for(h->tb.xmap_w, cols_w) + 1, cols_bounds_per_block(32)), n_frames),
                  dim3(256), 0, stream, d_descs, tbo, cols_w, 0, os.halo);
        prof_slot(1);
        XM_LAUNCH(kern, dim3(grid_for(h->tb.xmap_w, cols_w), n_frames), dim3(cols_threads(h, n_mean, cols_w, ept)), lds, stream, d_descs, tbo,
                  cols_w, os.halo, h->cols_flags | (d_descs_redo ? COLS_F_DEVICE_REDO : 0));
      } else {
      auto kern = k_scatter_cols_batch<AOS, false>;
      if constexpr (!AOS) {
        if (vec16) kern = k_scatter_cols_batch<false, true>;
      }
      const size_t lds = cols_lds_bytes(h, cols_w) + (size_t)h->cols_lds_pad;  // (pad: experiments -- fewer K1 blocks per CU, room for another kernel's)
      rc = h->ensure_lds(reinterpret_cast<const void*>(kern), lds);
      if (rc) return rc;
      prof_slot(0);
      if (cols_w <= 16)  // (16 lanes per boundary: see cols_bounds_per_block)
        XM_LAUNCH((k_cols_bounds_batch<AOS, 16>), dim3(grid_for(grid_for(h->tb.xmap_w, cols_w) + 1, cols_bounds_per_block(16)), n_frames),
                  dim3(256), 0, stream, d_descs, h->tb, cols_w, 0, 0);
      else
        XM_LAUNCH(k_cols_bounds_batch<AOS>, dim3(grid_for(grid_for(h->tb.xmap_w, cols_w) + 1, cols_bounds_per_block(32)), n_frames),
                  dim3(256), 0, stream, d_descs, h->tb, cols_w, 0, 0);
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
    if (key32) XM_LAUNCH(k_frame_cam32_batch, dim3(grid_for(h->tb.cam_w, CAM32_T), grid_for(h->tb.cam_h, CAM32_T), n_frames), dim3(BLOCK), 0, stream, d_descs, h->tb.cam_w, h->tb.cam_h, h->tb.dlut);
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


}  // namespace
