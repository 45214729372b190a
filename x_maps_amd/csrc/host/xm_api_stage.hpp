// xm_api_stage.hpp -- C-ABI: per-event debug outputs and the stage-by-stage API with the reference's signatures (A1 .. A7, point cloud)
// (part of libxmaps_hip.so's host side: included by ../xmaps_hip.hip, one translation unit; see that file for the order)
#pragma once

extern "C" {

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


}  // extern "C"
