// xm_own_plan.hpp -- owner tiles: the rig's ownership tables, worked out once on the host (own_plan, pure host code) and uploaded (own_setup)
// (part of libxmaps_hip.so's host side: included by ../xmaps_hip.hip, one translation unit; see that file for the order)
#pragma once

namespace {

// ---- owner tiles (xmaps_k1own.hpp): the rig's ownership tables, worked out once on the host -------------------------------------
struct OwnPlan {  // what own_plan() works out (host memory) and own_setup() uploads
  bool ok = false, all_in = true;
  int W = 0, halo = 0, r_lo = 0, hr = 0, hrp = 0, rp = 0, nxs_max = 0, extra_max = 0, m = 0, bias = 0, extra_cols = 0, delta_max = 0;
  std::vector<uint16_t> packed, xextra, masks;
  std::vector<int4> tiles;
  std::vector<int16_t> bases;
  std::vector<u32> extra_flat;
};

// Pure host code (no device needed: xm_own_plan_info runs it for the CPU tests).  pl.ok says whether the rig qualifies.
void own_plan(const xm_config* cfg, int xmap_h, int xr_min, OwnPlan& pl) {
  const int xmap_w = cfg->xmap_width, rect_w = cfg->rect_width, rect_h = cfg->rect_height, x_off = cfg->x_offset;
  const int rows = std::min(xmap_h - 1, rect_h);
  if (rows <= 0 || rect_h < xmap_h - 1 || xr_min <= -x_off) return;  // (an undefined X-map cell, 0, must read as dead)
  int yr_min = 32767, yr_max = -32768;
  const size_t cam_px = (size_t)cfg->cam_width * cfg->cam_height;
  for (size_t i = 0; i < cam_px; ++i) {
    yr_min = std::min<int>(yr_min, cfg->cam_mapy_i16[i]);
    yr_max = std::max<int>(yr_max, cfg->cam_mapy_i16[i]);
  }
  const int r_lo = std::max(0, yr_min) & ~7, r_hi = std::min(yr_max, rows - 1);
  if (r_hi < r_lo) return;
  const int hr = r_hi - r_lo + 1, hrp = (hr + 7) & ~7;
  int W = 8;
  if (const char* e = dbg_opt("XM_OWN_W")) W = atoi(e);
  W = std::max(OWN_BW, std::min(W, 32)) / OWN_BW * OWN_BW;  // (K0b: half a wave per boundary computes the thresholds up to the next one)
  // 1. owner column of every cell, row by row: delta = column - first column of the row that maps to the same cell
  std::vector<uint16_t>& packed = pl.packed;  // [c][row], as tb.xmap
  packed.assign((size_t)xmap_w * xmap_h, 0);
  std::vector<int> first(rect_w, -1), fcs((size_t)xmap_w);
  int delta_max = 0;
  bool all_in = true;
  for (int r = r_lo; r <= r_hi; ++r) {
    const int16_t* X = cfg->proj_x_map + (size_t)r * xmap_w;
    for (int c = 0; c < xmap_w; ++c) {
      fcs[c] = -1;
      const int xp = X[c], fu = xp - x_off;
      if (fu < xr_min) continue;  // dead: no LUT entry gives disp >= 0
      if (xp < 0 || xp >= (1 << OWN_XP_BITS)) return;
      int fc = (int)(short)fu;
      if (fc < 0) fc += rect_w;  // NumPy's negative wrap
      const bool in = fc >= 0 && fc < rect_w;
      if (!in || fu < 0) all_in = false;  // the kernel's lean path takes fu as the column
      int delta = 0;
      if (in) {
        if (first[fc] < 0) first[fc] = c;
        delta = c - first[fc];
        fcs[c] = fc;
      }
      if (delta > OWN_MAX_DELTA) return;
      delta_max = std::max(delta_max, delta);
      packed[(size_t)c * xmap_h + r] = (uint16_t)(xp | (delta << OWN_XP_BITS));
    }
    for (int c = 0; c < xmap_w; ++c)
      if (fcs[c] >= 0) first[fcs[c]] = -1;
  }
  if (delta_max == 0) return;  // an injective X-map: the plain column tiles' business
  const int halo = delta_max;  // (a tile reads its own columns and `halo` behind them: K0b finds that boundary too)
  if (halo >= W) return;
  // 2. the shear: slope of the cell column against the row along the middle time columns (least squares over the live entries)
  double slope = 0.0;
  {
    double sn = 0, sx = 0, sy = 0, sxx = 0, sxy = 0;
    for (int c = xmap_w / 4; c < xmap_w; c += std::max(1, xmap_w / 4)) {
      sn = sx = sy = sxx = sxy = 0;
      for (int r = r_lo; r <= r_hi; ++r) {
        const int fu = cfg->proj_x_map[(size_t)r * xmap_w + c] - x_off;
        if (fu < xr_min || fu < 0 || fu >= rect_w) continue;
        sn += 1; sx += r; sy += fu; sxx += (double)r * r; sxy += (double)r * fu;
      }
      if (sn >= 16 && sn * sxx - sx * sx > 0) {
        slope = (sn * sxy - sx * sy) / (sn * sxx - sx * sx);
        if (c >= xmap_w / 2) break;  // prefer the middle column
      }
    }
  }
  int m = 0;
  if (std::fabs(slope) * hr >= 24.0) m = (int)std::lround(-slope * 8.0 * 4096.0);
  if (const char* e = dbg_opt("XM_OWN_SHEAR")) m = atoi(e);  // experiments
  int sh_min = 0, sh_max = 0;
  for (int g = 0; g <= (rect_h - 1) >> 3; ++g) {
    const int sh = (g * m) >> 12;
    sh_min = std::min(sh_min, sh);
    sh_max = std::max(sh_max, sh);
  }
  const int bias = -sh_min, extra = sh_max - sh_min;
  if (rect_w + extra > 32767) return;
  // 3. per (tile, row): where its cells lie in the sheared frame.  The band of a row = the window of NX frame columns that
  //    holds most of the row's owner cells (a tile's cells of one row are a short run; the run moves with the row by what the
  //    frame's shear leaves of the X-map's slant); owner cells outside it are "extras" (where the rectified
  //    time map replicates its border the X-map jumps by hundreds of columns: first / last tile of the ESL rig).  NX = the
  //    narrowest band that leaves (almost) no more extras than the widest one.
  const int nt = (xmap_w + W - 1) / W, ng = hrp;  // (one band position per row)
  const auto owner_col = [&](int r, int c, int& xs) {  // owner pairs only: the cell's column in the sheared frame
    const uint16_t pk = packed[(size_t)c * xmap_h + r];
    if (!pk || (pk >> OWN_XP_BITS) != 0) return false;
    int fc = (int)(short)((int)(pk & ((1 << OWN_XP_BITS) - 1)) - x_off);
    if (fc < 0) fc += rect_w;
    if (fc < 0 || fc >= rect_w) return false;
    xs = fc + bias + (((r >> 3) * m) >> 12);
    return true;
  };
  std::vector<std::vector<int>> cells((size_t)nt * ng);  // sorted sheared columns of every (tile, row)'s owner cells
  for (int r = r_lo; r <= r_hi; ++r)
    for (int c = 0; c < xmap_w; ++c) {
      int xs;
      if (owner_col(r, c, xs)) cells[(size_t)(c / W) * ng + (r - r_lo)].push_back(xs);
    }
  for (auto& v : cells) std::sort(v.begin(), v.end());
  const auto best_window = [](const std::vector<int>& v, int nx, int& start) {  // most cells inside [start, start + nx)
    size_t best = 0, j = 0;
    start = v.empty() ? 0 : v[0];
    for (size_t i = 0; i < v.size(); ++i) {
      if (i && v[i] == v[i - 1]) continue;
      while (j < v.size() && v[j] < v[i] + nx) ++j;
      if (j - i > best) {
        best = j - i;
        start = v[i];
      }
    }
    return best;
  };
  size_t extras_at[OWN_MAX_NXS + 1] = {};
  for (int nx = 1; nx <= OWN_MAX_NXS; ++nx)
    for (const auto& v : cells) {
      int st;
      extras_at[nx] += v.size() - best_window(v, nx, st);
    }
  int NX = OWN_MAX_NXS;
  while (NX > 1 && extras_at[NX - 1] <= extras_at[OWN_MAX_NXS] + extras_at[OWN_MAX_NXS] / 8 + 64) NX -= 1;
  std::vector<int4>& tiles = pl.tiles;
  std::vector<int16_t>& bases = pl.bases;
  tiles.assign(nt, make_int4(0, 0, 0, 0));
  bases.assign((size_t)nt * ng, 0);
  for (int t = 0; t < nt; ++t)
    for (int g = 0; g < ng; ++g) {
      int st;
      best_window(cells[(size_t)t * ng + g], NX, st);
      bases[(size_t)t * ng + g] = (int16_t)st;
    }
  std::vector<uint16_t>&masks = pl.masks, &xextra = pl.xextra;
  masks.assign((size_t)nt * hrp, 0);
  xextra.assign((size_t)xmap_w * xmap_h, 0);
  std::vector<std::vector<u32>> extra_cells(nt);
  for (int r = r_lo; r <= r_hi; ++r)
    for (int c = 0; c < xmap_w; ++c) {
      int xs;
      if (!owner_col(r, c, xs)) continue;
      const int t = c / W, k = xs - bases[(size_t)t * ng + (r - r_lo)];
      if (k >= 0 && k < NX) {
        masks[(size_t)t * hrp + (r - r_lo)] |= (uint16_t)(1u << k);
        tiles[t].x = std::max(tiles[t].x, k + 1);
      } else {
        extra_cells[t].push_back((u32)xs * (u32)rect_h + (u32)r);
        if (extra_cells[t].size() > 4096) return;  // (a wild X-map: the packed keys stay)
        xextra[(size_t)c * xmap_h + r] = (uint16_t)extra_cells[t].size();
      }
    }
  int nxs_max = 1, extra_max = 0;
  std::vector<u32>& extra_flat = pl.extra_flat;
  extra_flat.clear();
  for (int t = 0; t < nt; ++t) {
    tiles[t].y = (int)extra_flat.size();
    tiles[t].z = (int)extra_cells[t].size();
    extra_flat.insert(extra_flat.end(), extra_cells[t].begin(), extra_cells[t].end());
    nxs_max = std::max(nxs_max, tiles[t].x);
    extra_max = std::max(extra_max, tiles[t].z);
  }
  extra_max = (extra_max + 3) & ~3;
  // rows per pass of the tile's LDS slots: the fewest passes (<= 4) that leave room for OWN_LDS_TARGET-sized blocks -- the
  // kernel is a chain of dependent round trips, what hides them is the number of tiles a CU holds at once
  int passes = 1;
  const auto rows_per_pass = [&](int P) { return ((hrp / 8 + P - 1) / P) * 8; };
  while (passes < OWN_MAX_ROW_PASSES && own_plan_lds_bytes(nxs_max, rows_per_pass(passes), hrp, extra_max) > OWN_LDS_TARGET) passes += 1;
  if (const char* e = dbg_opt("XM_OWN_ROW_PASSES")) passes = std::max(1, std::min(atoi(e), OWN_MAX_ROW_PASSES));
  {  // the kernel finds a row's pass as row * ceil(2^20 / rp) >> 20
    const auto magic_ok = [&](int rp_) {
      const u32 inv = ((1u << 20) + (u32)rp_ - 1u) / (u32)rp_;
      for (int r = 0; r < hrp; ++r)
        if ((int)(((u32)r * inv) >> 20) != r / rp_) return false;
      return true;
    };
    while (passes > 1 && !magic_ok(rows_per_pass(passes))) passes -= 1;
  }
  const int rp = rows_per_pass(passes);
  if (own_plan_lds_bytes(nxs_max, rp, hrp, extra_max) > 60 * 1024) return;  // LDS per block
  extra_flat.push_back(0);
  pl.W = W; pl.halo = halo; pl.r_lo = r_lo; pl.hr = hr; pl.hrp = hrp; pl.rp = rp; pl.nxs_max = nxs_max; pl.extra_max = extra_max;
  pl.m = m; pl.bias = bias; pl.extra_cols = extra; pl.delta_max = delta_max; pl.all_in = all_in;
  pl.ok = true;
}

// Returns XM_OK whether or not the rig qualifies (h->own_mode says); an error only for HIP failures.
int own_setup(xm_handle* h, const xm_config* cfg, int xr_min) {
  OwnPlan pl;
  own_plan(cfg, h->tb.xmap_h, xr_min, pl);
  if (!pl.ok) return XM_OK;
  const auto up = [](auto** dst, const auto& v) -> hipError_t {
    typedef typename std::remove_reference<decltype(v)>::type::value_type E;
    hipError_t e = hipMalloc((void**)dst, v.size() * sizeof(E) + 64);
    return e != hipSuccess ? e : hipMemcpy(*dst, v.data(), v.size() * sizeof(E), hipMemcpyHostToDevice);
  };
  HIP_TRY(up(&h->d_xmap_own, pl.packed));
  HIP_TRY(up(&h->d_xmap_extra, pl.xextra));
  HIP_TRY(up(&h->d_own_tiles, pl.tiles));
  {  // band position | ownership mask << 16 per (tile, row): one table, read by 16-byte loads at the head of every tile
    std::vector<u32> bm(pl.bases.size());
    for (size_t i = 0; i < bm.size(); ++i) {  // (the band's origin BEFORE the frame's shear: an event's band column = cell column - origin)
      const int row = (int)(i % (size_t)pl.hrp) + pl.r_lo;
      const int org = (int)pl.bases[i] - pl.bias - (((row >> 3) * pl.m) >> 12);
      bm[i] = (u32)(uint16_t)(int16_t)org | ((u32)pl.masks[i] << 16);
    }
    HIP_TRY(up(&h->d_own_bm, bm));
  }
  HIP_TRY(up(&h->d_own_extra_cells, pl.extra_flat));
  h->own_extras = (int)pl.extra_flat.size() - 1;
  h->tb.xmap_own = h->d_xmap_own;
  h->tb.xmap_extra = h->d_xmap_extra;
  h->tb.own_tiles = h->d_own_tiles;
  h->tb.own_bm = h->d_own_bm;
  h->tb.own_extra_cells = h->d_own_extra_cells;
  h->tb.own_r_lo = pl.r_lo;
  h->tb.own_hr = pl.hr;
  h->tb.own_hrp = pl.hrp;
  h->tb.own_rp = pl.rp;
  h->tb.own_nxs_max = pl.nxs_max;
  h->tb.own_extra_max = pl.extra_max;
  h->tb.shear_m = pl.m;
  h->tb.shear_bias = pl.bias;
  h->tb.shear_extra = pl.extra_cols;
  h->own_mode = true;
  if (const char* e = dbg_opt("XM_OWN_EPT")) h->own_ept_forced = atoi(e);
  h->own_w = pl.W;
  h->own_halo = pl.halo;
  if (pl.all_in) h->cols_flags |= COLS_F_ALL_IN_FRAME;
  return XM_OK;
}


}  // namespace
