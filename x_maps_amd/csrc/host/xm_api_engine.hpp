// xm_api_engine.hpp -- C-ABI: lifetime (xm_create / xm_destroy), synchronisation, the fused hot path (frames, groups, adaptive batching), profiling
// (part of libxmaps_hip.so's host side: included by ../xmaps_hip.hip, one translation unit; see that file for the order)
#pragma once

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
  // extrema pass on every frame
  h->try_sorted = !h->time_sorted && !(cfg->flags & XM_FLAG_GENERAL);
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
    if (const char* e = dbg_opt("XM_K2_PPT")) h->k2_force_ppt = atoi(e);
    double mean_cells2 = 0.0;
    bool pipe_ok_g[3] = {false, false, false};
    for (int g = 0; g < 3; ++g) {
      const int ppt = 1 << g;
      const unsigned tiles_x = grid_for(cfg->proj_width, K2_TX * ppt), tiles_y = grid_for(cfg->proj_height, K2_TY);
      XM_TRY_CREATE(hipMalloc((void**)&h->d_k2_tiles[g], (size_t)tiles_x * tiles_y * sizeof(int4)));
      XM_TRY_CREATE(hipMalloc((void**)&h->d_k2_pix[g], (size_t)cfg->proj_width * cfg->proj_height * sizeof(u32)));
      if (g == 0) hipLaunchKernelGGL(k_build_k2_tables<1>, dim3(tiles_x, tiles_y), dim3(K2_TX * K2_TY), 0, 0, h->tb, h->d_k2_tiles[g], h->d_k2_pix[g]);
      else if (g == 1) hipLaunchKernelGGL(k_build_k2_tables<2>, dim3(tiles_x, tiles_y), dim3(K2_TX * K2_TY), 0, 0, h->tb, h->d_k2_tiles[g], h->d_k2_pix[g]);
      else hipLaunchKernelGGL(k_build_k2_tables<4>, dim3(tiles_x, tiles_y), dim3(K2_TX * K2_TY), 0, 0, h->tb, h->d_k2_tiles[g], h->d_k2_pix[g]);
      XM_TRY_CREATE(hipGetLastError());
      if (g > 0) {  // the pipelined kernel's u16 copy
        h->k2_pix_stride = (cfg->proj_width + 7) & ~7;
        const size_t n16 = (size_t)h->k2_pix_stride * cfg->proj_height;
        XM_TRY_CREATE(hipMalloc((void**)&h->d_k2_pix16[g], n16 * sizeof(uint16_t) + 16));
        hipLaunchKernelGGL(k_k2_pix_to_u16, dim3(grid_for(n16, BLOCK)), dim3(BLOCK), 0, 0, h->d_k2_pix[g], h->d_k2_pix16[g], cfg->proj_width,
                           cfg->proj_height, h->k2_pix_stride);
        XM_TRY_CREATE(hipGetLastError());
      }
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
      pipe_ok_g[g] = pipe_ok;
      // Four pixels per thread when the 32 x 16-pixel tiles' patches are small against the tile (a projector image finer than
      // the rectified frame: < 2 patch cells per pixel): an item's fixed costs -- five barriers, the descriptor reads, the tile
      // arithmetic -- then weigh more than its patch, and half as many items carry the same pixels.  (Eight per thread --
      // 128 x 16 tiles -- was built and measured on the ESL-like rig: 94 VGPRs, five blocks per CU, K2 7.3-7.9 against 5.9-6.7 us
      // per frame; removed.)
      if (g == 2) {
        const char* e4 = dbg_opt("XM_K2_PIPE_PPT");  // experiments / tests: 2 / 4
        const bool small = mean_cells2 < 2.0 * (2 * K2_TX * K2_TY);
        const int want = e4 ? atoi(e4) : small ? 4 : 2;
        h->k2_pipe_g = h->k2_pipe_rig_ok && want >= 4 && pipe_ok_g[2] ? 2 : 1;
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
    const char* e32 = dbg_opt("XM_KEY32");
    // (camera view: (event index + 1) << 12 | disparity on the camera frame -- only the disparity range matters)
    h->key32_ok = (cfg->view != XM_VIEW_PROJECTOR || (cfg->rect_height & 3) == 0) && max_disp < (1l << KEY32_DISP_BITS) &&
                  !(e32 && e32[0] == '0');
    // the pipelined K2 keeps the per-disparity table in LDS: every disparity an event of this rig can have (<= 4096 entries, 32 KB)
    // (at most 2048 entries = 16 KB: with the patch and the staging rows a block then stays under 27 KB of LDS, six blocks per CU;
    //  a larger disparity -- the reference's ESL calibration allows 3808 through LUT entries far outside the frame, no rendered
    //  frame comes near -- reads the global table)
    int nlds_max = 2048;
    if (const char* e = dbg_opt("XM_K2_NLDS_MAX")) nlds_max = std::max(1, std::min(4096, atoi(e)));  // experiments
    h->k2_pipe_nlds = (int)std::max<long>(1, std::min<long>(std::min<long>(max_disp + 1, 65536), nlds_max));
    if (const char* e = dbg_opt("XM_K2_PIPE")) {
      h->k2_pipe = e[0] != '0';
      h->k2_pipe_force = e[0] == '2';
    }
    if (const char* e = dbg_opt("XM_K2_CHAIN")) h->k2_chain = e[0] != '0';
    if (const char* e = dbg_opt("XM_K2_PER_CU")) {
      h->k2_per_cu_max = std::max(1, atoi(e));
    }
    if (const char* e = dbg_opt("XM_COLS_LDS_PAD")) h->cols_lds_pad = std::max(0, std::min(64 * 1024, atoi(e)));
    if (const char* e = dbg_opt("XM_K2_CONSEC")) h->k2_consec = e[0] != '0' ? 1 : 0;  // experiments / tests: the strided pixel assignment
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
    const char* ec = dbg_opt("XM_COLS");
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
    const char* e1 = dbg_opt("XM_K1_DIRECT");
    const char* e2 = dbg_opt("XM_K2_DIRECT");
    h->k1_direct = e1 && e1[0] == '1';
    h->k2_direct = e2 && e2[0] == '1';
    if (const char* e3 = dbg_opt("XM_K2_FLAGS")) h->k2_flags = e3[0] == '1';
    // C-1M needs 44 KB (w_ts = 5, w_x = 16): three blocks per CU beside K2's 12 KB blocks
    size_t budget = 76 * 1024;
    int w_ts = 5, w_x = 16;  // 5 time columns, 16 camera columns: 70 KB at C-1M
    if (const char* e = dbg_opt("XM_K1_WX")) w_x = std::max(2, std::min(64, atoi(e)));  // experiments: the LUT band's width in camera columns
    if (w_ts > 64) w_ts = 64;
    auto need = [&](int wt, int wx) {
      // must mirror the carve-up at the top of k_scatter_tiled (uint4 units, +1 uint4 of alignment slack per band)
      const size_t win_words = cfg->view == XM_VIEW_PROJECTOR ? (size_t)wt * xmap_h : (size_t)wx * cfg->cam_height;
      constexpr size_t slack = 64;  // LDS-direct band loads write whole waves: one wave of slack behind each band
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
  if (const char* e = dbg_opt("XM_ABLATE")) {
    int v = atoi(e);
    XM_TRY_CREATE(hipMemcpyToSymbol(HIP_SYMBOL(xm::g_ablate), &v, sizeof v));
  }
#endif
  XM_TRY_CREATE(hipMalloc((void**)&h->d_states, sizeof(SlotState) * (n_slots + 1)));
  XM_TRY_CREATE(hipMemset(h->d_states, 0, sizeof(SlotState) * (n_slots + 1)));  // host_flags = NULL
  // (a memset of device memory may return before it has run, and the slots' non-blocking streams do not wait for the default one:
  //  k_reset_slot below initialises the extrema slots in the same bytes -- the memset must have landed first)
  XM_TRY_CREATE(hipDeviceSynchronize());
  h->aux_st = h->d_states + n_slots;
  h->slots.resize(n_slots);
  for (int i = 0; i < n_slots; ++i) {
    Slot& s = h->slots[i];
    {
      // The slots' streams get their own hardware queues: HIP multiplexes all streams of one priority onto
      // GPU_MAX_HW_QUEUES (4) hardware queues, the application's default stream included, and how the eight slot streams
      // happened to interleave with it cost up to 17 % of the pipelined frame rate (first engine of a process: 64 Gev/s,
      // second: 75; tools/engine_order_probe.py).  Streams of another priority live in another queue pool.
      int lo = 0, hi = 0;
      XM_TRY_CREATE(hipDeviceGetStreamPriorityRange(&lo, &hi));  // lo = least, hi = greatest priority (numerically lowest)
      // one stream per hardware queue; slots beyond that share them (more streams than queues is where the runtime's
      // stream -> queue assignment starts to matter, and it only added buffering, no overlap)
      static const int hw_q = getenv("GPU_MAX_HW_QUEUES") && atoi(getenv("GPU_MAX_HW_QUEUES")) > 0 ? atoi(getenv("GPU_MAX_HW_QUEUES")) : 4;
      const int n_streams = hw_q;
      if (n_streams > 0 && i >= n_streams) {
        s.stream = h->slots[i % n_streams].stream;
        s.owns_stream = false;
      } else if (cfg->flags & XM_FLAG_DEFAULT_STREAMS)
        XM_TRY_CREATE(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
      else XM_TRY_CREATE(hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, hi));
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
    if (h->try_sorted) {
      XM_TRY_CREATE(hipHostMalloc((void**)&s.h_flags, 64, hipHostMallocMapped));
      s.h_flags[0] = s.h_flags[1] = 0;
      u32* d_flags = nullptr;
      XM_TRY_CREATE(hipHostGetDevicePointer((void**)&d_flags, s.h_flags, 0));
      XM_TRY_CREATE(hipMemcpy(&s.st->host_flags, &d_flags, sizeof d_flags, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_reset_slot, dim3(1024), dim3(BLOCK), 0, s.stream, s.st, s.key_frame, (u64)h->key_cells, s.dirty);
    XM_TRY_CREATE(hipGetLastError());
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
  // (the memsets and table builders above ran on the default stream; the slots' streams are non-blocking: nothing of a first frame
  //  may overtake them)
  XM_TRY_CREATE(hipDeviceSynchronize());
  {  // launch workers: one per distinct slot stream (XM_FLAG_LAUNCH_WORKERS; off: launches stay in the calling thread)
    const char* we = dbg_opt("XM_WORKERS");  // overrides the flag either way
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
  if (!h->pending.empty()) (void)flush_pending(h);  // XM_FLAG_ADAPTIVE_BATCH: frames still held back are submitted, not dropped
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
  if (h->d_shard_n) (void)hipFree(h->d_shard_n);
  for (auto gs : h->gstreams) if (gs) (void)hipStreamDestroy(gs);
  for (auto& e : h->prof_ev) if (e) (void)hipEventDestroy(e);
  if (h->fork_ev) (void)hipEventDestroy(h->fork_ev);
  for (auto& e : h->join_ev) if (e) (void)hipEventDestroy(e);
  if (h->stage_frame) (void)hipFree(h->stage_frame);
  if (h->d_states) (void)hipFree(h->d_states);
  if (h->d_lut) (void)hipFree(h->d_lut);
  if (h->d_xmap) (void)hipFree(h->d_xmap);
  for (xm_handle::OwnSet& os : h->own) {
    if (os.d_xmap_own) (void)hipFree(os.d_xmap_own);
    if (os.d_tiles) (void)hipFree(os.d_tiles);
    if (os.d_xmap_extra) (void)hipFree(os.d_xmap_extra);
    if (os.d_bm) (void)hipFree(os.d_bm);
    if (os.d_extra_cells) (void)hipFree(os.d_extra_cells);
  }
  for (hipEvent_t e : h->k2_chain_ev)
    if (e) (void)hipEventDestroy(e);
  if (h->d_pmap) (void)hipFree(h->d_pmap);
  if (h->d_dlut) (void)hipFree(h->d_dlut);
  for (int g = 0; g < 3; ++g) {
    if (h->d_k2_tiles[g]) (void)hipFree(h->d_k2_tiles[g]);
    if (h->d_k2_pix[g]) (void)hipFree(h->d_k2_pix[g]);
    if (h->d_k2_pix16[g]) (void)hipFree(h->d_k2_pix16[g]);
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
    const xm_handle::OwnSet& os = h->own[0];  // (the plan frames take by default; [10], [11]: width and halo of the one for denser frames)
    info[1] = os.w;
    info[2] = os.halo;
    info[3] = os.nxs_max;
    info[4] = h->tb.shear_m;
    info[5] = h->tb.shear_extra;
    info[6] = os.r_lo;
    info[7] = os.hr;
    info[8] = os.extras;
    info[9] = os.extra_max;
    for (int i = 1; i < xm_handle::OWN_PLANS; ++i)  // (the narrowest plan: the one the densest frames take)
      if (h->own[i].ok) {
        info[10] = h->own[i].w;
        info[11] = h->own[i].halo;
      }
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
  OwnPlan pls[OWN_PLANS];
  own_plans(cfg, xmap_h, xr_min, pls);
  const OwnPlan& pl = pls[0];  // (the plan frames take by default)
  if (!pl.ok) return XM_OK;
  info[0] = 2; info[1] = pl.W; info[2] = pl.halo; info[3] = pl.nxs_max; info[4] = pl.m; info[5] = pl.extra_cols; info[6] = pl.r_lo;
  info[7] = pl.hr; info[8] = (int)pl.extra_flat.size() - 1; info[9] = pl.extra_max; info[10] = pl.delta_max;
  info[11] = (int)own_plan_lds_bytes(pl.nxs_max, pl.rp, pl.hrp, pl.extra_max, pl.grouped);
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
  if (h->try_sorted) {  // frames whose shortcut failed are redone now, then waited for
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
    if (bad) HIP_TRY(hipDeviceSynchronize());  // (a memset of device memory may return early; the slots' streams do not wait for the default one)
    if (bad) return fail(XM_ERR_UNSORTED, "XM_FLAG_TIME_SORTED: %u wavefront(s) saw events outside [t[0], t[n-1]] -- a frame "
                         "processed since the last xm_sync was not time-sorted, its output is invalid", bad);
  }
  return XM_OK;
}

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

// tests: the u16 disparity frame the column / owner tiles left in the slot of the handle's last frame (A3's output before K2),
// un-sheared, as [rect_h][rect_w] row-major like the reference's disp_map.  Meaningful only if that frame took the tiles
// (xm_path_counts) and was not redone.
int xm_debug_last_disp_frame(xm_handle* h, uint16_t* out_host) {
  if (!h || !out_host) return fail(XM_ERR_INVALID, "NULL argument");
  XM_ENTER(h);
  Slot& s = h->slots[h->last_slot];
  if (!s.frame16) return fail(XM_ERR_INVALID, "this handle has no column-tile frames");
  if (s.pending_batch_ev) {
    HIP_TRY(hipEventSynchronize(s.pending_batch_ev));
    s.pending_batch_ev = nullptr;
  }
  HIP_TRY(hipStreamSynchronize(s.stream));
  const size_t cells = frame16_cells(h->tb);
  std::vector<uint16_t> raw(cells);
  HIP_TRY(hipMemcpy(raw.data(), s.frame16, cells * sizeof(uint16_t), hipMemcpyDeviceToHost));
  const int rw = h->tb.rect_w, rh = h->tb.rect_h;
  for (int r = 0; r < rh; ++r)
    for (int x = 0; x < rw; ++x) out_host[(size_t)r * rw + x] = raw[(size_t)frame16_col(h->tb, x, r) * rh + r];
  return XM_OK;
}

// tests: frames finished by the software-pipelined K2 (k_frame_proj_pipe) since xm_create
int xm_debug_k2_pipe_frames(xm_handle* h, uint64_t* count) {
  if (!h || !count) return fail(XM_ERR_INVALID, "NULL argument");
  XM_ENTER(h);
  *count = h->k2_pipe_frames;
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


}  // extern "C"
