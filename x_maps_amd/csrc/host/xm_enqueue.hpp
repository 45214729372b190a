// xm_enqueue.hpp -- which K1 a frame takes (path selection) and the launches of ONE frame (enqueue_frame)
// (part of libxmaps_hip.so's host side: included by ../xmaps_hip.hip, one translation unit; see that file for the order)
#pragma once

namespace {

// time columns per tile for frames of n events: about cols_target events per tile, within the LDS budget; 0 = not this path
int cols_width(const xm_handle* h, u64 n) {
  if (h->cols_ok && h->own_mode) {  // owner tiles: the widths the ownership tables are built for; not for nearly empty frames
    // the default plan (wide tiles) while a tile's own + halo columns fit ONE event pass of a block (8 events x 512 threads: past
    // that a tile looks its events up again for every row pass); denser frames take the second plan's narrow tiles
    const double per_col = (double)n / (double)h->tb.xmap_w;
    int pick = 0;  // (the mean tile: measured on the ESL-like frames at 94 % of a block's events, the fuller tiles' second pass included)
    while (pick + 1 < xm_handle::OWN_PLANS && h->own[pick + 1].ok &&
           per_col * (h->own[pick].w + h->own[pick].halo) > (double)(COLS_EPT * COLS_MAX_THREADS))
      pick += 1;
    const xm_handle::OwnSet& os = h->own[pick];
    const u64 tiles = grid_for(h->tb.xmap_w, os.w);
    return n >= tiles * 128 && n < (1ull << 28) ? os.w : 0;
  }
  if (!h->cols_ok || h->cols_w_max < 1 || h->tb.xmap_w < 1 || n == 0 || n >= (1ull << 28)) return 0;
  const double per_col = (double)n / (double)h->tb.xmap_w;
  int W = (int)((double)h->cols_target / per_col);
  W = std::max(1, std::min(W, h->cols_w_max));
  if (per_col * W < 1024.0) return 0;  // sparse frames: the band copies and the slot scan would dominate (direct kernel instead)
  return W;
}

// owner tiles: events per thread -- eight; four (twice the waves at half the registers for the same tile: xmaps_k1own.hpp) measured
// the same within the noise (profiles/r05_own_tiles.md) and is kept as an experiment switch ("XM_OWN_EPT" = 4; not for 16-byte SoA loads)
// the plan whose tiles are W columns wide (cols_width picked it)
const xm_handle::OwnSet& own_set(const xm_handle* h, int W) {
  for (int i = 1; i < xm_handle::OWN_PLANS; ++i)
    if (h->own[i].ok && h->own[i].w == W) return h->own[i];
  return h->own[0];
}

int own_ept(const xm_handle* h, u64 n, int W, bool vec16) {
  if (vec16 || h->own_ept_forced != 4) return 8;
  const double per = (double)n / (double)h->tb.xmap_w * (W + own_set(h, W).halo);
  return per * 1.12 <= 4.0 * COLS_MAX_THREADS ? 4 : 8;
}

unsigned cols_threads(const xm_handle* h, u64 n, int W, int ept = COLS_EPT) {
  if (h->own_mode) {  // own + halo columns in one pass where 512 threads hold them
    const double per = (double)n / (double)h->tb.xmap_w * (W + own_set(h, W).halo);
    const unsigned t = ((unsigned)(per * 1.12 / ept) + 63u) / 64u * 64u;
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
  const int split = h->own_mode ? own_set(h, W).halo : 0;  // owner tiles: two boundaries per tile (tile = W columns + a halo behind them)
  const unsigned nb = (split ? 2u : 1u) * grid_for(h->tb.xmap_w, W);
  const bool g16 = !split && W <= 16;  // (16 lanes per boundary on the column tiles: cols_bounds_per_block)
  const unsigned gx = grid_for(nb + 1, cols_bounds_per_block(g16 ? 16 : 32));
  if (ev.aos && g16)
    XM_LAUNCH((k_cols_bounds<true, 16>), dim3(gx), dim3(256), 0, stream, ev.x, (const long long*)ev.t, (const uint4*)ev.aos, (u32)ev.n, h->tb, W, frame16, split);
  else if (ev.aos)
    XM_LAUNCH(k_cols_bounds<true>, dim3(gx), dim3(256), 0, stream, ev.x, (const long long*)ev.t, (const uint4*)ev.aos, (u32)ev.n, h->tb, W, frame16, split);
  else if (g16)
    XM_LAUNCH((k_cols_bounds<false, 16>), dim3(gx), dim3(256), 0, stream, ev.x, (const long long*)ev.t, (const uint4*)ev.aos, (u32)ev.n, h->tb, W, frame16, split);
  else
    XM_LAUNCH(k_cols_bounds<false>, dim3(gx), dim3(256), 0, stream, ev.x, (const long long*)ev.t, (const uint4*)ev.aos, (u32)ev.n, h->tb, W, frame16, split);
}

int launch_scatter_cols(xm_handle* h, const EventsView& ev, SlotState* st, uint16_t* frame16, int W, hipStream_t stream) {
  const bool vec16 = !ev.aos && aligned(ev.x, 16) && aligned(ev.y, 16) && aligned(ev.t, 16);
  if (h->own_mode) {
    const int ept = own_ept(h, ev.n, W, vec16);
    auto kern = k_scatter_own<false, false>;
    if (ev.aos) kern = ept == 4 ? k_scatter_own<true, false, 4> : k_scatter_own<true, false>;
    else if (vec16) kern = k_scatter_own<false, true>;
    else if (ept == 4) kern = k_scatter_own<false, false, 4>;
    const xm_handle::OwnSet& os = own_set(h, W);
    DevTables tbo = h->tb;
    own_apply(os, tbo);
    const size_t lds = own_lds_bytes(os);
    int rc = h->ensure_lds(reinterpret_cast<const void*>(kern), lds);
    if (rc) return rc;
    XM_LAUNCH(kern, dim3(grid_for(h->tb.xmap_w, W)), dim3(cols_threads(h, ev.n, W, ept)), lds, stream, ev.x, ev.y, (const long long*)ev.t,
              (const uint4*)ev.aos, (u32)ev.n, tbo, st, frame16, W, os.halo, h->cols_flags);
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
    XM_LAUNCH(k_frame_cam32, dim3(grid_for(h->tb.cam_w, CAM32_T), grid_for(h->tb.cam_h, CAM32_T)), dim3(BLOCK), 0, stream,
              reinterpret_cast<u32*>(const_cast<u64*>(key_frame)), h->tb.cam_w, h->tb.cam_h, st, h->tb.dlut, depth, bgr);
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
  static const int skip = dbg_opt("XM_SKIP_MASK") ? atoi(dbg_opt("XM_SKIP_MASK")) : 0;  // experiments: 1=K0 2=K1 4=K2
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


}  // namespace
