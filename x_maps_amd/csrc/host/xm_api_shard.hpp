// xm_api_shard.hpp -- C-ABI: one index shard of a frame (extrema, scatter with global indices, decode, finish, band-sharded finish)
// (part of libxmaps_hip.so's host side: included by ../xmaps_hip.hip, one translation unit; see that file for the order)
#pragma once

extern "C" {

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


// ---- shards on the column tiles: every time column processed by one rank, u16 frames merged by SUM (xmaps_k1cols.hpp) ----
int xm_shard_cols_info(xm_handle* h, uint64_t n_frame_events, size_t* frame_bytes, size_t* reduce_u32, size_t* send_bytes, size_t* cap_events) {
  if (!h) return fail(XM_ERR_INVALID, "NULL handle");
  const int W = h->own_mode ? 0 : cols_width(h, n_frame_events);
  if (!h->cols_ok || h->own_mode || W == 0 || h->cfg.view != XM_VIEW_PROJECTOR || h->k2_direct)
    return fail(XM_ERR_INVALID, "this rig / frame density does not take the column tiles (injective X-map, projector view, >= 1024 events per tile)");
  const size_t cells = frame16_cells(h->tb);
  // room for the events of four mean time columns (a shard's last column), a multiple of 8
  const size_t cap = ((size_t)(4.0 * (double)n_frame_events / (double)h->tb.xmap_w) + 1024 + 7) & ~(size_t)7;
  if (frame_bytes) *frame_bytes = cols_frame_bytes(cells, h->tb.xmap_w);
  if (reduce_u32) *reduce_u32 = (cells + 1) / 2;
  if (send_bytes) *send_bytes = sizeof(ShardColsHeader) + cap * 12;
  if (cap_events) *cap_events = cap;
  return XM_OK;
}

int xm_shard_cols_pack(xm_handle* h, const uint16_t* x, const uint16_t* y, const int64_t* t, size_t n, void* send_buf_dev, size_t cap_events) {
  if (!h || !send_buf_dev || (n && (!x || !y || !t))) return fail(XM_ERR_INVALID, "NULL argument");
  XM_ENTER(h);
  const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>(32, (std::min(n, cap_events) + 1023) / 1024));
  hipLaunchKernelGGL(k_shard_cols_pack, dim3(blocks), dim3(256), 0, h->slots[0].stream, x, y, (const long long*)t, (u64)n,
                     (unsigned char*)send_buf_dev, (u64)cap_events);
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

int xm_shard_cols_scatter(xm_handle* h, uint16_t* x, uint16_t* y, int64_t* t, size_t n, uint64_t n_frame_events, const void* gathered_dev,
                          size_t send_bytes, int rank, int world, size_t cap_events, uint16_t* frame16) {
  if (!h || !gathered_dev || !frame16 || !x || !y || !t) return fail(XM_ERR_INVALID, "NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(XM_ERR_INVALID, "bad rank / world");
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)t) & 15) return fail(XM_ERR_INVALID, "x / y / t must be 16-byte aligned");
  XM_ENTER(h);
  const int W = cols_width(h, n_frame_events);
  if (!h->cols_ok || h->own_mode || W == 0) return fail(XM_ERR_INVALID, "this rig / frame density does not take the column tiles");
  if (!h->d_shard_n) HIP_TRY(hipMalloc((void**)&h->d_shard_n, 64 + sizeof(FrameDesc)));
  hipStream_t s = h->slots[0].stream;
  long long* mm = reinterpret_cast<long long*>(h->d_shard_n);  // {tmin, -tmax} of the frame
  FrameDesc* desc = reinterpret_cast<FrameDesc*>(reinterpret_cast<unsigned char*>(h->d_shard_n) + 64);
  hipLaunchKernelGGL(k_shard_cols_prepare, dim3(1), dim3(256), 0, s, x, y, (long long*)t, (u64)n, (const unsigned char*)gathered_dev,
                     (u64)send_bytes, rank, world, h->tb, (u64)cap_events, mm, frame16, h->aux_st, desc);
  const int flags = h->cols_flags | COLS_F_EXT_EXTREMA;
  if (W <= 16)  // (16 lanes per boundary: cols_bounds_per_block)
    hipLaunchKernelGGL((k_cols_bounds_batch<false, 16>), dim3(grid_for(grid_for(h->tb.xmap_w, W) + 1, cols_bounds_per_block(16)), 1), dim3(256), 0, s,
                       (const FrameDesc*)desc, h->tb, W, flags, 0);
  else
    hipLaunchKernelGGL(k_cols_bounds_batch<false>, dim3(grid_for(grid_for(h->tb.xmap_w, W) + 1, cols_bounds_per_block(32)), 1), dim3(256), 0, s,
                       (const FrameDesc*)desc, h->tb, W, flags, 0);
  auto kern = k_scatter_cols_batch<false, true>;  // (the piece starts 8-aligned: 16-byte event loads)
  const size_t lds = cols_lds_bytes(h, W);
  int rc;
  if ((rc = h->ensure_lds(reinterpret_cast<const void*>(kern), lds))) return rc;
  if (dbg_opt("XM_SHARD_PROFILE"))  // (measurement: HIP events tied to THIS dispatch -- xm_shard_cols_last_k1_ms -- beside the three-launch bracket a caller can take itself)
    hipExtLaunchKernelGGL(kern, dim3(grid_for(h->tb.xmap_w, W), 1), dim3(cols_threads(h, n_frame_events, W)), (std::uint32_t)lds, s, h->prof_ev[2],
                          h->prof_ev[3], 0u, (const FrameDesc*)desc, h->tb, W, h->w_x, h->cols_xr_min, flags);
  else
    hipLaunchKernelGGL(kern, dim3(grid_for(h->tb.xmap_w, W), 1), dim3(cols_threads(h, n_frame_events, W)), lds, s, (const FrameDesc*)desc, h->tb, W,
                       h->w_x, h->cols_xr_min, flags);
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

// duration of the column-tile K1 of the last xm_shard_cols_scatter issued while xm_debug_option("XM_SHARD_PROFILE", "1") was set
// (synchronises the handle's stream)
int xm_shard_cols_last_k1_ms(xm_handle* h, float* ms) {
  if (!h || !ms) return fail(XM_ERR_INVALID, "NULL argument");
  XM_ENTER(h);
  HIP_TRY(hipStreamSynchronize(h->slots[0].stream));
  HIP_TRY(hipEventElapsedTime(ms, h->prof_ev[2], h->prof_ev[3]));
  return XM_OK;
}

// did any piece of the frames since the last call object (see xmaps_k1cols.hpp)?  Synchronises the handle's stream.
int xm_shard_cols_failed(xm_handle* h, int* failed) {
  if (!h || !failed) return fail(XM_ERR_INVALID, "NULL argument");
  XM_ENTER(h);
  hipStream_t s = h->slots[0].stream;
  u32 v = 0;
  HIP_TRY(hipMemcpyAsync(&v, &h->aux_st->unsorted_sticky, sizeof v, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  *failed = v ? 1 : 0;
  if (v) {
    HIP_TRY(hipMemsetAsync(&h->aux_st->unsorted_sticky, 0, sizeof(u32), s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  return XM_OK;
}

}  // extern "C"
