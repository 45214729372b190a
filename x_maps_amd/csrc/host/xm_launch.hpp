// xm_launch.hpp -- kernel dispatch: K0 / K1 / K2 launch helpers of every variant (general, tiled, column tiles, owner tiles, pipelined K2)
// (part of libxmaps_hip.so's host side: included by ../xmaps_hip.hip, one translation unit; see that file for the order)
#pragma once

namespace {

// ---- kernel dispatch ---------------------------------------------------------------------------------
template <typename T, bool AOS, bool HAS_P>
void launch_minmax_t(const EventsView& ev, SlotState* st, u32 tag_override, hipStream_t stream) {
  const u64 n = ev.n;
  const bool vec2 = !AOS && sizeof(T) == 8 && std::is_same<T, long long>::value && aligned(ev.t, 16) &&
                    (!HAS_P || aligned(ev.p, 4));
  // ~2048 events per thread-block iteration keeps every CU busy without drowning the 32 atomic slots
  const unsigned per_block = BLOCK * (vec2 ? 2 * K0_UN : 4);  // VEC2: K0_UN loads x 2 events per thread per sweep
  unsigned grid = grid_for(n, per_block);
  if (grid > 1024) grid = 1024;
  if constexpr (std::is_same<T, long long>::value && !AOS) {
    if (vec2) {
      XM_LAUNCH((k_minmax<T, false, HAS_P, 2>), dim3(grid), dim3(BLOCK), 0, stream, (const T*)ev.t, ev.p,
                (const uint4*)nullptr, n, st, tag_override);
      return;
    }
  }
  XM_LAUNCH((k_minmax<T, AOS, HAS_P, 1>), dim3(grid), dim3(BLOCK), 0, stream, (const T*)ev.t, ev.p,
            (const uint4*)ev.aos, n, st, tag_override);
}

void launch_minmax(const EventsView& ev, SlotState* st, u32 tag_override, hipStream_t stream) {
  if (ev.aos) {
    if (ev.use_p) launch_minmax_t<long long, true, true>(ev, st, tag_override, stream);
    else launch_minmax_t<long long, true, false>(ev, st, tag_override, stream);
    return;
  }
  const bool hp = ev.use_p;
  switch (ev.t_dtype) {
    case XM_T_INT64:
      hp ? launch_minmax_t<long long, false, true>(ev, st, tag_override, stream)
         : launch_minmax_t<long long, false, false>(ev, st, tag_override, stream);
      break;
    case XM_T_FLOAT32:
      hp ? launch_minmax_t<float, false, true>(ev, st, tag_override, stream)
         : launch_minmax_t<float, false, false>(ev, st, tag_override, stream);
      break;
    default:
      hp ? launch_minmax_t<double, false, true>(ev, st, tag_override, stream)
         : launch_minmax_t<double, false, false>(ev, st, tag_override, stream);
  }
}

struct ScatterArgs {
  xm_handle* h;
  const EventsView* ev;
  const DevTables* tb;
  int view;
  SlotState* st;
  u32 tag_override;
  u64 idx_offset, mm_lo, mm_hi;
  const void* mm_ext;  // sharded mode: {tmin, -tmax} in device memory (NULL: mm_lo / mm_hi)
  u64* frame;
  unsigned char* dirty;
  hipStream_t stream;
  int w_ts, w_x;
  size_t lds;
  bool direct;
  bool sorted;
  bool key32;
};

template <typename T, bool AOS, bool HAS_P, int VIEW>
int launch_scatter_tv(const ScatterArgs& a) {
  const EventsView& ev = *a.ev;
  const u64 n = ev.n;
  const bool vec16 = !AOS && aligned(ev.x, 16) && aligned(ev.y, 16) && aligned(ev.t, 16) && (!HAS_P || aligned(ev.p, 16));
  const bool vec = !AOS && aligned(ev.x, 8) && aligned(ev.y, 8) && aligned(ev.t, 16) && (!HAS_P || aligned(ev.p, 8));
  // The tiled kernel pays a fixed price per block (copy the bands, clear + scan w_ts * xmap_h slots), so it needs blocks
  // of >= 1024 events whose time slice still fits the LDS window of w_ts X-map columns.  A frame of n events spreads over
  // xmap_w columns: a block of E events spans about E * xmap_w / n of them.  Dense frames (C-1M: 1 M events / 640
  // columns) get 4096-event blocks; sparse ones (ESL-like: 150 K events / 1080 columns, < 1 event per slot, nothing to
  // de-duplicate) go to the one-thread-per-event kernel, whose cost is proportional to n.
  const double max_ev = a.tb->xmap_w > 0 ? (a.w_ts - 1.5) * (double)n / (double)a.tb->xmap_w : 0.0;
  if (!a.direct && a.w_ts > 0 && a.w_x > 0 && max_ev >= 1024.0) {
    // vector-load variant: 16-byte aligned SoA columns with int64 t (the EventCD time type); everything else takes the
    // lane-strided loads (any alignment)
    constexpr bool kHasVec = !AOS && std::is_same<T, long long>::value;
    auto kern = k_scatter_tiled<T, AOS, HAS_P, VIEW, false>;
    if constexpr (kHasVec) {
      if (vec16) kern = k_scatter_tiled<T, AOS, HAS_P, VIEW, true>;
    }
    if (a.key32) {  // compact key frame (a.frame points at it)
      kern = k_scatter_tiled<T, AOS, HAS_P, VIEW, false, true>;
      if constexpr (kHasVec) {
        if (vec16) kern = k_scatter_tiled<T, AOS, HAS_P, VIEW, true, true>;
      }
    }
    // raise the kernel's dynamic-LDS cap once per (handle = device, kernel instantiation); gfx950: 160 KB / CU
    {
      int rc_lds = a.h->ensure_lds(reinterpret_cast<const void*>(kern), a.lds);
      if (rc_lds) return rc_lds;
    }
    unsigned threads = TILE_THREADS;
    while (threads > 1024 / TILE_EPT && (double)(threads * TILE_EPT) > max_ev) threads >>= 1;  // smallest block: 1024 events
    XM_LAUNCH(kern, dim3(grid_for(n, threads * TILE_EPT)), dim3(threads), a.lds, a.stream, ev.x, ev.y,
              (const T*)ev.t, ev.p, (const uint4*)ev.aos, n, a.idx_offset, *a.tb, a.st, a.tag_override,
              a.mm_lo, a.mm_hi, a.mm_ext, a.frame, a.dirty, a.w_ts, a.w_x, a.sorted ? 1 : 0);
    return XM_OK;
  }
  if constexpr (AOS) {
    XM_LAUNCH((k_scatter<T, true, HAS_P, 1, VIEW>), dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, a.stream,
              (const uint16_t*)nullptr, (const uint16_t*)nullptr, (const T*)nullptr, (const int16_t*)nullptr,
              (const uint4*)ev.aos, n, a.idx_offset, *a.tb, a.st, a.tag_override, a.mm_lo, a.mm_hi, a.mm_ext, a.frame, a.dirty,
              a.sorted ? 1 : 0);
  } else if (vec) {
    XM_LAUNCH((k_scatter<T, false, HAS_P, 4, VIEW>), dim3(grid_for(n, BLOCK * 4)), dim3(BLOCK), 0, a.stream,
              ev.x, ev.y, (const T*)ev.t, ev.p, (const uint4*)nullptr, n, a.idx_offset, *a.tb, a.st,
              a.tag_override, a.mm_lo, a.mm_hi, a.mm_ext, a.frame, a.dirty, a.sorted ? 1 : 0);
  } else {
    XM_LAUNCH((k_scatter<T, false, HAS_P, 1, VIEW>), dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, a.stream, ev.x,
              ev.y, (const T*)ev.t, ev.p, (const uint4*)nullptr, n, a.idx_offset, *a.tb, a.st, a.tag_override,
              a.mm_lo, a.mm_hi, a.mm_ext, a.frame, a.dirty, a.sorted ? 1 : 0);
  }
  return XM_OK;
}

template <typename T, bool AOS, bool HAS_P>
int launch_scatter_t(const ScatterArgs& a) {
  return a.view == XM_VIEW_PROJECTOR ? launch_scatter_tv<T, AOS, HAS_P, 0>(a) : launch_scatter_tv<T, AOS, HAS_P, 1>(a);
}

int launch_scatter(xm_handle* h, const EventsView& ev, SlotState* st, u32 tag_override, u64 idx_offset, u64 mm_lo,
                   u64 mm_hi, u64* frame, unsigned char* dirty, hipStream_t stream, bool sorted = false,
                   const void* mm_ext = nullptr, bool key32 = false) {
  ScatterArgs a{h, &ev, &h->tb, h->cfg.view, st, tag_override, idx_offset, mm_lo, mm_hi, mm_ext, frame, dirty, stream,
                h->w_ts, h->w_x, h->k1_lds, h->k1_direct, sorted, key32};
  if (ev.aos) return ev.use_p ? launch_scatter_t<long long, true, true>(a) : launch_scatter_t<long long, true, false>(a);
  switch (ev.t_dtype) {
    case XM_T_INT64: return ev.use_p ? launch_scatter_t<long long, false, true>(a) : launch_scatter_t<long long, false, false>(a);
    case XM_T_FLOAT32: return ev.use_p ? launch_scatter_t<float, false, true>(a) : launch_scatter_t<float, false, false>(a);
    default: return ev.use_p ? launch_scatter_t<double, false, true>(a) : launch_scatter_t<double, false, false>(a);
  }
}

// ---- column-tile K1 (xmaps_k1cols.hpp) ------------------------------------------------------------------------------------------
// kmode of a frame: 0 = 64-bit key frame (general), 1 = compact 32-bit key frame, 2 = column tiles + plain u16 frame
enum { KM_KEY64 = 0, KM_KEY32 = 1, KM_COLS = 2 };

// ---- K2 launches (tiled frame kernel, projector view) -----------------------------------------------------------------------
// Pixels per thread of a launch over n_frames frames: two (32 x 16-pixel tiles) when the launch fills the chip several times
// over -- a K2 wave is a chain of dependent round trips, what it costs there is resident waves x lifetime, so each wave carries
// two pixels through the chain --, one (16 x 16) for a lone small frame, where the chain's length IS the kernel's duration and
// twice the blocks start at once (C-1M, one frame: 7.3 us with one pixel per thread, 12.3 with two).
int k2_ppt(const xm_handle* h, int n_frames) {
  if (h->k2_force_ppt == 1 || h->k2_force_ppt == 2) return h->k2_force_ppt;
  const u64 blocks2 = (u64)grid_for(h->tb.proj_w, 2 * K2_TX) * grid_for(h->tb.proj_h, K2_TY) * (u64)std::max(n_frames, 1);
  // measured at C-1M (600 blocks of 32 x 16 pixels per frame, tools/ppt_threshold.sh), K2 us per launch with one / two pixels per
  // thread: 1 frame 5.3 / 7.2, 2 frames 8.7 / 8.6, 3: 11.1 / 10.6, 8: 23.4 / 21.3, 16: 47.5 / 41.1 -- the crossover is at about
  // four blocks per CU
  return blocks2 >= 1024 ? 2 : 1;
}

// K2's dynamic LDS: the tile's patch of u16 disparities (the row maxima replace it in place) + the overrun of its last read
size_t k2_lds_bytes(const xm_handle* h, int ppt) {
  return (size_t)(h->k2_tile_cap[ppt - 1] + 32) * sizeof(uint16_t);
}

template <int FMT>
void launch_k2(xm_handle* h, hipStream_t stream, const u64* key_frame, SlotState* st, u32 tag_override, const unsigned char* dirty,
               float* depth, uint8_t* bgr, bool unsheared = false, int col_lo = 0, int col_hi = 0) {
  const int ppt = k2_ppt(h, 1);
  const dim3 grid(grid_for(h->tb.proj_w, K2_TX * ppt), grid_for(h->tb.proj_h, K2_TY));
  DevTables tb = h->tb;
  if (unsheared) tb.shear_m = tb.shear_bias = tb.shear_extra = 0;  // a plain [rect_w][rect_h] u16 frame (shards), not a slot's frame16
  if (ppt == 1)
    XM_LAUNCH((k_frame_proj_tiled<FMT, 1>), grid, dim3(K2_TX * K2_TY), k2_lds_bytes(h, 1), stream, key_frame, tb, st, tag_override,
              dirty, (const ulonglong2*)h->d_zero16, depth, bgr, h->k2_tile_cap[0], col_lo, col_hi);
  else
    XM_LAUNCH((k_frame_proj_tiled<FMT, 2>), grid, dim3(K2_TX * K2_TY), k2_lds_bytes(h, 2), stream, key_frame, tb, st, tag_override,
              dirty, (const ulonglong2*)h->d_zero16, depth, bgr, h->k2_tile_cap[1], col_lo, col_hi);
}

// the software-pipelined K2 (persistent blocks walking (frame, tile) items): groups on the plain u16 frame, two pixels per thread
size_t k2_pipe_lds_bytes(const xm_handle* h, int g) {
  return (size_t)((h->k2_tile_cap[g] + 32 + 7) & ~7) * sizeof(uint16_t) + (size_t)h->k2_pipe_nlds * sizeof(uint2);
}

template <int COND = 0>
bool launch_k2_pipe(xm_handle* h, hipStream_t stream, const FrameDesc* d_descs, int n_frames) {
  if (!h->k2_pipe || !h->k2_pipe_rig_ok || h->k2_pipe_nlds < 1 || (k2_ppt(h, n_frames) != 2 && !h->k2_pipe_force)) return false;
  const int g = h->k2_pipe_g, ppt = 1 << g;
  const u32 gx = grid_for(h->tb.proj_w, K2_TX * ppt), gy = grid_for(h->tb.proj_h, K2_TY);
  const u64 total = (u64)gx * gy * (u64)n_frames;
  const size_t lds = k2_pipe_lds_bytes(h, g);
  const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>((size_t)h->k2_per_cu_max, (160 * 1024) / (lds + 2048)));
  unsigned blocks = (unsigned)h->n_cus * per_cu / 8 * 8;
  if (total < 3ull * blocks && !h->k2_pipe_force) return false;  // too few items per block for the pipeline to matter: one block per tile
  blocks = (unsigned)std::min<u64>(blocks, std::max<u64>(total, 1));
  // divmod(tile, gx) by a multiply in the kernel: exact while tile * gx < 2^32
  const u32 gx_magic = gx > 1 && (u64)gx * gy * gx < (1ull << 32) ? (u32)(((1ull << 32) + gx - 1) / gx) : 0u;
  const bool cs = (h->k2_consec < 0 ? g >= 2 : h->k2_consec != 0) && h->d_k2_pix16[g];  // default: consecutive pixels on the 64 x 16 tiles
  const void* fn = g == 2 ? (cs ? reinterpret_cast<const void*>(k_frame_proj_pipe<4, true, COND>) : reinterpret_cast<const void*>(k_frame_proj_pipe<4, false, COND>))
                          : (cs ? reinterpret_cast<const void*>(k_frame_proj_pipe<2, true, COND>) : reinterpret_cast<const void*>(k_frame_proj_pipe<2, false, COND>));
  if (h->ensure_lds(fn, lds) != XM_OK) return false;
  K2PipeArgs pa;
  pa.proj_w = h->tb.proj_w; pa.proj_h = h->tb.proj_h; pa.rect_w = h->tb.rect_w; pa.rect_h = h->tb.rect_h;
  pa.shear_m = h->tb.shear_m; pa.shear_bias = h->tb.shear_bias;
#define XM_K2P_LAUNCH(P, C)                                                                                                          \
  XM_LAUNCH((k_frame_proj_pipe<P, C, COND>), dim3(blocks), dim3(K2_TX * K2_TY), lds, stream, d_descs, (const int4*)h->d_k2_tiles[g],      \
            (const u32*)h->d_k2_pix[g], (const uint16_t*)h->d_k2_pix16[g], h->k2_pix_stride, h->tb.dlut, pa, h->k2_tile_cap[g],      \
            (u32)n_frames, gx, gy, h->k2_pipe_nlds, gx_magic)
  std::unique_lock<std::mutex> chain_lock(h->k2_chain_mu, std::defer_lock);
  if (h->k2_chain && COND == 0) {  // one K2 at a time: wait for the one launched last (whatever its stream)
    chain_lock.lock();
    if (h->k2_chain_n) (void)hipStreamWaitEvent(stream, h->k2_chain_ev[(h->k2_chain_n - 1) % 16], 0);
  }
  if (g == 2) {
    if (cs) XM_K2P_LAUNCH(4, true);
    else XM_K2P_LAUNCH(4, false);
  } else {
    if (cs) XM_K2P_LAUNCH(2, true);
    else XM_K2P_LAUNCH(2, false);
  }
#undef XM_K2P_LAUNCH
  if (chain_lock.owns_lock()) {
    hipEvent_t& ev = h->k2_chain_ev[h->k2_chain_n % 16];
    if (!ev) (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (ev && hipEventRecord(ev, stream) == hipSuccess) h->k2_chain_n += 1;
  }
  h->k2_pipe_frames += (uint64_t)n_frames;
  return true;
}

template <int FMT, int COND = 0>
void launch_k2_batch(xm_handle* h, hipStream_t stream, const FrameDesc* d_descs, int n_frames) {
  if constexpr (FMT == 2 && (COND == 0 || COND == 2)) {  // (COND = 2: the attempt's K2 node of a captured batch)
    if (launch_k2_pipe<COND>(h, stream, d_descs, n_frames)) return;
  }
  const int ppt = k2_ppt(h, n_frames);
  dim3 grid(grid_for(h->tb.proj_w, K2_TX * ppt), grid_for(h->tb.proj_h, K2_TY), n_frames);
  if (COND == 1) grid = dim3(std::min(grid.x * grid.y, 32u), 1, n_frames);  // redo node: a few blocks per frame walk its tiles
  if (ppt == 1)
    XM_LAUNCH((k_frame_proj_tiled_batch<FMT, COND, 1>), grid, dim3(K2_TX * K2_TY), k2_lds_bytes(h, 1), stream, d_descs, h->tb,
              (const ulonglong2*)h->d_zero16, h->k2_tile_cap[0]);
  else
    XM_LAUNCH((k_frame_proj_tiled_batch<FMT, COND, 2>), grid, dim3(K2_TX * K2_TY), k2_lds_bytes(h, 2), stream, d_descs, h->tb,
              (const ulonglong2*)h->d_zero16, h->k2_tile_cap[1]);
}

size_t cols_lds_bytes(const xm_handle* h, int W) {  // mirrors the carve-up at the top of scatter_cols_body
  const size_t lut_q = ((size_t)h->w_x * h->tb.cam_h + 3) / 4 + 1 + 64, xm_q = ((size_t)W * h->tb.xmap_h + 7) / 8 + 1 + 64,
               slot_q = ((size_t)W * h->tb.xmap_h + 3) / 4;
  return 16 * (lut_q + xm_q + slot_q);
}

size_t own_plan_lds_bytes(int nxs_max, int rp, int hrp, int extra_max, bool grouped) {  // mirrors the carve-up at the top of scatter_own_body
  return (size_t)4 * nxs_max * rp + (size_t)4 * extra_max + (size_t)4 * own_tab_words(hrp, grouped);
}
size_t own_lds_bytes(const xm_handle::OwnSet& os) {
  return own_plan_lds_bytes(os.nxs_max, os.rp, os.hrp, os.extra_max, os.grouped != 0);
}


}  // namespace
