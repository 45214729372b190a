// xm_api_misc.hpp -- C-ABI: X-map construction (N1), evaluation metrics (N4), pinned host / device memory helpers
// (part of libxmaps_hip.so's host side: included by ../xmaps_hip.hip, one translation unit; see that file for the order)
#pragma once

extern "C" {

// value == NULL removes the option; name == NULL removes all of them
int xm_debug_option(const char* name, const char* value) {
  std::lock_guard<std::mutex> lk(g_opt_mu);
  if (!name) g_opts.clear();
  else if (!value) g_opts.erase(name);
  else {
    auto it = g_opts.find(name);
    if (it != g_opts.end() && *it->second == value) return XM_OK;  // (unchanged: no new text)
    g_opt_texts.emplace_back(value);
    g_opts[name] = &g_opt_texts.back();  // (deque: growing it never moves an element)
  }
  return XM_OK;
}


// ---- N1: X-map construction ----------------------------------------------------------------------------------
int xm_build_x_map(int device, const float* time_map, int height, int width, int x_map_width, int t_px_scale,
                   int x_offset, int num_scanlines, int16_t* x_map_out, float* t_diffs_out) {
  if (!time_map || !x_map_out || height <= 0 || width <= 0 || x_map_width <= 0 || t_px_scale <= 0 || num_scanlines <= 0)
    return fail(XM_ERR_INVALID, "bad argument");
  if (height > 32767 || width + x_offset > 32767) return fail(XM_ERR_INVALID, "indices must fit int16 (x_maps_disparity.py:52-53)");
  if ((size_t)width * sizeof(double) > 150 * 1024) return fail(XM_ERR_INVALID, "time-map row does not fit LDS");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(XM_ERR_HIP, "no HIP device visible");
  HIP_TRY(hipSetDevice(device));
  const size_t n_in = (size_t)height * width, n_out = (size_t)height * x_map_width;
  float* d_in = nullptr;
  int16_t* d_x = nullptr;
  float* d_d = nullptr;
  int rc = XM_OK;
  do {
    hipError_t e;
    if ((e = hipMalloc((void**)&d_in, n_in * 4)) != hipSuccess || (e = hipMalloc((void**)&d_x, n_out * 2)) != hipSuccess ||
        (t_diffs_out && (e = hipMalloc((void**)&d_d, n_out * 4)) != hipSuccess) ||
        (e = hipMemcpy(d_in, time_map, n_in * 4, hipMemcpyHostToDevice)) != hipSuccess) {
      rc = fail(XM_ERR_HIP, "xm_build_x_map: %s", hipGetErrorString(e));
      break;
    }
    int wp = 2;
    while (wp < width) wp <<= 1;
    const char* es = dbg_opt("XM_XMAP_SCAN");  // tests: XM_XMAP_SCAN=1 takes the exhaustive scan (read at every call)
    const bool brute = es && es[0] == '1';
    if (wp <= 8192 && !brute) {  // rows of up to 8192 columns: the sorted-row kernel (96 KB of LDS at most)
      const size_t lds = (size_t)wp * sizeof(u64) + (size_t)width * sizeof(float);
      if (lds > 64 * 1024 &&
          (e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_build_x_map_sorted), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds)) != hipSuccess) {
        rc = fail(XM_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        break;
      }
      hipLaunchKernelGGL(k_build_x_map_sorted, dim3(height), dim3(BLOCK), lds, 0, d_in, height, width, wp, x_map_width, t_px_scale,
                         x_offset, 2.0 / (double)num_scanlines, d_x, d_d);
    } else {
      const size_t lds = (size_t)width * sizeof(double);
      if (lds > 64 * 1024 &&
          (e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_build_x_map), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds)) != hipSuccess) {
        rc = fail(XM_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        break;
      }
      hipLaunchKernelGGL(k_build_x_map, dim3(height), dim3(BLOCK), lds, 0, d_in, height, width, x_map_width, t_px_scale,
                         x_offset, 2.0 / (double)num_scanlines, d_x, d_d);
    }
    if ((e = hipGetLastError()) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess ||
        (e = hipMemcpy(x_map_out, d_x, n_out * 2, hipMemcpyDeviceToHost)) != hipSuccess ||
        (t_diffs_out && (e = hipMemcpy(t_diffs_out, d_d, n_out * 4, hipMemcpyDeviceToHost)) != hipSuccess)) {
      rc = fail(XM_ERR_HIP, "xm_build_x_map: %s", hipGetErrorString(e));
      break;
    }
  } while (0);
  if (d_in) (void)hipFree(d_in);
  if (d_x) (void)hipFree(d_x);
  if (d_d) (void)hipFree(d_d);
  return rc;
}

// ---- N4: evaluation metrics -------------------------------------------------------------------------------------
int xm_eval_stats(int device, const float* estimate, const float* groundtruth, int height, int width, int filter,
                  float min_depth, float max_depth, xm_eval_result* out) {
  if (!estimate || !groundtruth || !out || height <= 0 || width <= 0) return fail(XM_ERR_INVALID, "bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(XM_ERR_HIP, "no HIP device visible");
  HIP_TRY(hipSetDevice(device));
  const u64 n = (u64)height * width;
  float *d_e = nullptr, *d_g = nullptr;
  EvalAcc* d_a = nullptr;
  int rc = XM_OK;
  EvalAcc a;
  memset(&a, 0, sizeof a);
  do {
    hipError_t e;
    if ((e = hipMalloc((void**)&d_e, n * 4)) != hipSuccess || (e = hipMalloc((void**)&d_g, n * 4)) != hipSuccess ||
        (e = hipMalloc((void**)&d_a, sizeof(EvalAcc))) != hipSuccess ||
        (e = hipMemcpy(d_e, estimate, n * 4, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(d_g, groundtruth, n * 4, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemset(d_a, 0, sizeof(EvalAcc))) != hipSuccess) {
      rc = fail(XM_ERR_HIP, "xm_eval_stats: %s", hipGetErrorString(e));
      break;
    }
    unsigned grid = grid_for(n, BLOCK * 8);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL((k_eval_stats<1>), dim3(grid), dim3(BLOCK), 0, 0, (const float*)d_e, (const float*)d_g, n, filter, min_depth, max_depth, d_a);
    hipLaunchKernelGGL((k_eval_stats<2>), dim3(grid), dim3(BLOCK), 0, 0, (const float*)d_e, (const float*)d_g, n, filter, min_depth, max_depth, d_a);
    if ((e = hipGetLastError()) != hipSuccess || (e = hipMemcpy(&a, d_a, sizeof a, hipMemcpyDeviceToHost)) != hipSuccess) {
      rc = fail(XM_ERR_HIP, "xm_eval_stats: %s", hipGetErrorString(e));
      break;
    }
  } while (0);
  if (d_e) (void)hipFree(d_e);
  if (d_g) (void)hipFree(d_g);
  if (d_a) (void)hipFree(d_a);
  if (rc) return rc;
  const double hw = (double)n;
  out->margin = 0.01 * a.sum_gt / (double)a.n_gt_pos;
  out->fillrate = ((double)a.n_close - (double)a.n_gt_zero) / (hw - (double)a.n_gt_zero);
  out->rmse = a.n_valid ? std::sqrt(a.sum_sq / (double)a.n_valid) : 0.0;
  out->perc_1 = 100.0 * (double)a.n1 / hw;
  out->perc_5 = 100.0 * (double)a.n5 / hw;
  out->perc_10 = 100.0 * (double)a.n10 / hw;
  out->n_valid = a.n_valid;
  out->n_gt_zero = a.n_gt_zero;
  return XM_OK;
}

// ---- pinned host memory --------------------------------------------------------------------------------------
int xm_host_alloc(xm_handle* h, size_t bytes, void** out) {
  if (!h || !out) return fail(XM_ERR_INVALID, "NULL argument");
  XM_ENTER(h);
  HIP_TRY(hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault));
  return XM_OK;
}
int xm_host_free(xm_handle* h, void* p) {
  if (!h) return fail(XM_ERR_INVALID, "NULL handle");
  XM_ENTER(h);
  if (p) HIP_TRY(hipHostFree(p));
  return XM_OK;
}

// ---- device memory helpers ----------------------------------------------------------------------------------
int xm_dev_alloc(xm_handle* h, size_t bytes, void** out) {
  if (!h || !out) return fail(XM_ERR_INVALID, "NULL argument");
  XM_ENTER(h);
  HIP_TRY(hipMalloc(out, bytes ? bytes : 16));
  return XM_OK;
}
int xm_dev_free(xm_handle* h, void* p) {
  if (!h) return fail(XM_ERR_INVALID, "NULL handle");
  XM_ENTER(h);
  if (p) HIP_TRY(hipFree(p));
  return XM_OK;
}
int xm_dev_upload(xm_handle* h, void* dst_dev, const void* src_host, size_t bytes) {
  if (!h) return fail(XM_ERR_INVALID, "NULL handle");
  XM_ENTER(h);
  if (bytes) HIP_TRY(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
  return XM_OK;
}
int xm_dev_download(xm_handle* h, void* dst_host, const void* src_dev, size_t bytes) {
  if (!h) return fail(XM_ERR_INVALID, "NULL handle");
  XM_ENTER(h);
  if (bytes) HIP_TRY(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
  return XM_OK;
}


}  // extern "C"
