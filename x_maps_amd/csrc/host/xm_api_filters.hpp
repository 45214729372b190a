// xm_api_filters.hpp -- C-ABI: the frame event filters (N3) and pause detection (N2)
// (part of libxmaps_hip.so's host side: included by ../xmaps_hip.hip, one translation unit; see that file for the order)
#pragma once

extern "C" {

// ---- N3: frame event filters ------------------------------------------------------------------------------------
int xm_frame_event_filter(xm_handle* h, int filter, int intended_semantics, const void* eventcd16_in, size_t n,
                          const int16_t* xp_i16, int map_height, int map_width, void* eventcd16_out, size_t* n_out) {
  if (!h || !n_out || (n && !eventcd16_in)) return fail(XM_ERR_INVALID, "NULL argument");
  if (filter < FILTER_FIRST_PER_YT || filter > FILTER_MEAN_PER_XY) return fail(XM_ERR_INVALID, "unknown filter %d", filter);
  if (filter == FILTER_FIRST_PER_YT && n && !xp_i16) return fail(XM_ERR_INVALID, "FirstEventPerYT needs xp_i16");
  if (n >= 0xffffffffull) return fail(XM_ERR_TOO_MANY, "too many events");
  *n_out = 0;
  if (n == 0 || map_height <= 0 || map_width <= 0) return XM_OK;
  XM_ENTER(h);
  Slot& s = h->slots[0];
  const size_t cells = (size_t)map_height * map_width;
  if (cells >= 0x7fffffffull) return fail(XM_ERR_INVALID, "map too large");
  if (!eventcd16_out) return fail(XM_ERR_INVALID, "NULL output");
  const u32 n_blocks = (u32)((cells + SCAN_BLOCK - 1) / SCAN_BLOCK);
  int rc;
  if ((rc = stage_in(s.ev_aos, eventcd16_in, n * 16, s.stream))) return rc;
  if (filter == FILTER_FIRST_PER_YT && (rc = stage_in(s.ev_p, xp_i16, n * 2, s.stream))) return rc;
  // scratch: first[cells] last[cells] pos[cells] sums[n_blocks] total[1]
  if ((rc = s.dbg[0].reserve((3 * cells + n_blocks + 4) * sizeof(u32)))) return rc;
  if ((rc = s.dbg[1].reserve(cells * 16))) return rc;
  u32* first = (u32*)s.dbg[0].p;
  u32* last = first + cells;
  u32* pos = last + cells;
  u32* sums = pos + cells;
  u32* total = sums + n_blocks;
  HIP_TRY(hipMemsetAsync(first, 0xff, cells * sizeof(u32), s.stream));
  HIP_TRY(hipMemsetAsync(last, 0, cells * sizeof(u32), s.stream));
  if ((rc = rearm_aux(h, s.stream, nullptr, 0))) return rc;
  hipLaunchKernelGGL(k_filter_scatter, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, s.stream, (const uint4*)s.ev_aos.p,
                     (const int16_t*)s.ev_p.p, (u64)n, filter == FILTER_FIRST_PER_YT ? 1 : 0, map_height, map_width, first,
                     last, &h->aux_st->cnt[0][0][CNT_OOB]);
  hipLaunchKernelGGL(k_filter_scan_blocks, dim3(n_blocks), dim3(SCAN_BLOCK), 0, s.stream, last, (u32)cells, pos, sums);
  hipLaunchKernelGGL(k_filter_scan_sums, dim3(1), dim3(SCAN_BLOCK), 0, s.stream, sums, n_blocks, total);
  hipLaunchKernelGGL(k_filter_emit, dim3(n_blocks), dim3(SCAN_BLOCK), 0, s.stream, (const uint4*)s.ev_aos.p, first, last, pos,
                     sums, (u32)cells, map_width, filter, intended_semantics, (uint4*)s.dbg[1].p);
  HIP_TRY(hipGetLastError());
  u32 cnt = 0;
  HIP_TRY(hipMemcpyAsync(&cnt, total, sizeof cnt, hipMemcpyDeviceToHost, s.stream));
  if ((rc = read_oob(h, s.stream, "frame event filter"))) return rc;  // synchronises the stream
  if (cnt) HIP_TRY(hipMemcpy(eventcd16_out, s.dbg[1].p, (size_t)cnt * 16, hipMemcpyDeviceToHost));
  *n_out = cnt;
  return XM_OK;
}

// ---- N2: pause detection -----------------------------------------------------------------------------------------
int xm_find_pauses(xm_handle* h, const int64_t* t, const void* eventcd16, size_t n, int mem, int64_t thresh_us,
                   uint32_t* idx_out, size_t idx_capacity, size_t* n_out) {
  if (!h || !n_out || (!t == !eventcd16 && n)) return fail(XM_ERR_INVALID, "give exactly one of t / eventcd16");
  if (n >= 0x7fffffffull) return fail(XM_ERR_TOO_MANY, "too many events");
  *n_out = 0;
  if (n < 2) return XM_OK;
  XM_ENTER(h);
  Slot& s = h->slots[0];
  int rc;
  const long long* d_t = (const long long*)t;
  const uint4* d_aos = (const uint4*)eventcd16;
  if (mem == XM_MEM_HOST) {
    if (t) {
      if ((rc = stage_in(s.ev_t, t, n * 8, s.stream))) return rc;
      d_t = (const long long*)s.ev_t.p;
    } else {
      if ((rc = stage_in(s.ev_aos, eventcd16, n * 16, s.stream))) return rc;
      d_aos = (const uint4*)s.ev_aos.p;
    }
  } else if (mem != XM_MEM_DEVICE) {
    return fail(XM_ERR_INVALID, "mem must be XM_MEM_HOST or XM_MEM_DEVICE");
  }
  const u32 n_blocks = (u32)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
  // scratch: flags[n] pos[n] out[n] sums[n_blocks] total[1]
  if ((rc = s.dbg[0].reserve((3 * n + n_blocks + 4) * sizeof(u32)))) return rc;
  u32* flags = (u32*)s.dbg[0].p;
  u32* pos = flags + n;
  u32* out = pos + n;
  u32* sums = out + n;
  u32* total = sums + n_blocks;
  hipLaunchKernelGGL(k_pause_flags, dim3(n_blocks), dim3(SCAN_BLOCK), 0, s.stream, d_t, d_aos, (u32)n, (long long)thresh_us, flags);
  hipLaunchKernelGGL(k_filter_scan_blocks, dim3(n_blocks), dim3(SCAN_BLOCK), 0, s.stream, flags, (u32)n, pos, sums);
  hipLaunchKernelGGL(k_filter_scan_sums, dim3(1), dim3(SCAN_BLOCK), 0, s.stream, sums, n_blocks, total);
  hipLaunchKernelGGL(k_pause_emit, dim3(n_blocks), dim3(SCAN_BLOCK), 0, s.stream, flags, pos, sums, (u32)n, out);
  HIP_TRY(hipGetLastError());
  u32 cnt = 0;
  HIP_TRY(hipMemcpyAsync(&cnt, total, sizeof cnt, hipMemcpyDeviceToHost, s.stream));
  HIP_TRY(hipStreamSynchronize(s.stream));
  *n_out = cnt;
  const size_t ncopy = cnt < idx_capacity ? cnt : idx_capacity;
  if (ncopy && idx_out) HIP_TRY(hipMemcpy(idx_out, out, ncopy * sizeof(u32), hipMemcpyDeviceToHost));
  return XM_OK;
}


}  // extern "C"
