// xm_own_plan.hpp -- owner tiles: the rig's ownership tables, worked out once on the host (own_plan, pure host code) and uploaded (own_setup)
// (part of libxmaps_hip.so's host side: included by ../xmaps_hip.hip, one translation unit; see that file for the order)
#pragma once

namespace {

// ---- owner tiles (xmaps_k1own.hpp): the rig's ownership tables, worked out once on the host -------------------------------------
struct OwnPlan {  // what own_plan() works out (host memory) and own_setup() uploads
  bool ok = false, all_in = true, grouped = false;
  int W = 0, halo = 0, r_lo = 0, hr = 0, hrp = 0, rp = 0, nxs_max = 0, extra_max = 0, m = 0, bias = 0, extra_cols = 0, delta_max = 0;
  std::vector<uint16_t> packed, xextra;
  std::vector<u32> masks;        // [tiles][entries] band columns the tile owns (entry = a row, or an 8-row group)
  std::vector<int4> tiles;
  std::vector<int16_t> bases;    // [tiles][entries] first (sheared) frame column of the band
  std::vector<u32> extra_flat;
};

// Pure host code (no device needed: xm_own_plan_info runs it for the CPU tests).  pl.ok says whether the rig qualifies.
//
// Two ways to own a cell.  PER ROW: the owner column of cell (x, row) = the first time column of that row that maps to it; a tile
// owns an arbitrary subset of every row's band (a mask per row) and flushes it in 2-byte lanes.  PER 8-ROW GROUP (grouped): the
// owner column of (x, rows 8 g .. 8 g + 7) = the first time column of ANY of the eight rows that maps to column x -- a tile then
// owns whole 16-byte pieces of the frame (8 rows of one column: the frame's shear is constant there) and flushes them as such, a
// quarter of the texture addresser's cycles of the 2-byte flush (profiles/r05_own_tiles.md).  The price: delta = column - owner
// column grows by the X-map's slant over 8 rows, and with it the halo a tile reads.  Rigs whose grouped delta exceeds the packed
// X-map's 3 bits (or the tile) keep the per-row form.
void own_plan_as(const xm_config* cfg, int xmap_h, int xr_min, bool grouped, int W, OwnPlan& pl) {
  pl = OwnPlan{};
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
  const int sh = grouped ? 3 : 0, n_ent = hrp >> sh;  // entries of a tile's band table: rows, or 8-row groups
  W = std::max(OWN_BW, std::min(W, 32)) / OWN_BW * OWN_BW;  // (K0b: half a wave per boundary computes the thresholds up to the next one)
  // the frame column of a pair, as the kernel works it out: -1 = dead or outside the frame
  bool all_in = true, bad_xp = false;
  const auto cell_col = [&](int r, int c, bool& dead) {
    const int xp = cfg->proj_x_map[(size_t)r * xmap_w + c], fu = xp - x_off;
    dead = fu < xr_min;  // no LUT entry gives disp >= 0
    if (dead) return -1;
    if (xp < 0 || xp >= (1 << OWN_XP_BITS)) {
      bad_xp = true;
      return -1;
    }
    int fc = (int)(short)fu;
    if (fc < 0) fc += rect_w;  // NumPy's negative wrap
    const bool in = fc >= 0 && fc < rect_w;
    if (!in || fu < 0) all_in = false;  // the kernel's lean path takes fu as the column
    return in ? fc : -1;
  };
  // 1. owner column of every cell: delta = column - first column of the row (of the 8-row group) that maps to the same frame column
  std::vector<uint16_t>& packed = pl.packed;  // [c][row], as tb.xmap
  packed.assign((size_t)xmap_w * xmap_h, 0);
  std::vector<int> first(rect_w, -1);
  int delta_max = 0;
  const int unit = grouped ? 8 : 1;
  for (int r0 = r_lo; r0 <= r_hi; r0 += unit) {
    const int r1 = std::min(r0 + unit - 1, r_hi);
    for (int r = r0; r <= r1; ++r)
      for (int c = 0; c < xmap_w; ++c) {
        bool dead;
        const int fc = cell_col(r, c, dead);
        if (fc >= 0 && (first[fc] < 0 || c < first[fc])) first[fc] = c;
      }
    if (bad_xp) return;
    for (int r = r0; r <= r1; ++r)
      for (int c = 0; c < xmap_w; ++c) {
        bool dead;
        const int fc = cell_col(r, c, dead);
        if (dead) continue;
        const int delta = fc >= 0 ? c - first[fc] : 0;
        if (delta > OWN_MAX_DELTA) return;
        delta_max = std::max(delta_max, delta);
        packed[(size_t)c * xmap_h + r] = (uint16_t)(cfg->proj_x_map[(size_t)r * xmap_w + c] | (delta << OWN_XP_BITS));
      }
    for (int r = r0; r <= r1; ++r)
      for (int c = 0; c < xmap_w; ++c) {
        bool dead;
        const int fc = cell_col(r, c, dead);
        if (fc >= 0) first[fc] = -1;
      }
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
    const int shv = (g * m) >> 12;
    sh_min = std::min(sh_min, shv);
    sh_max = std::max(sh_max, shv);
  }
  const int bias = -sh_min, extra = sh_max - sh_min;
  if (rect_w + extra > 32767) return;
  // 3. per (tile, entry): where its cells lie in the sheared frame.  The band of an entry = the window of NX frame columns that
  //    holds most of the columns the tile owns there (a short run; it moves from entry to entry by what the frame's shear leaves of
  //    the X-map's slant); cells outside it are "extras" (where the rectified time map replicates its border the X-map jumps by
  //    hundreds of columns: first / last tile of the ESL rig).  NX = the narrowest band that leaves (almost) no more extras than
  //    the widest one.
  const int nt = (xmap_w + W - 1) / W;
  const int nx_cap = grouped ? 32 : OWN_MAX_NXS;  // (band columns: the bits of an entry's ownership word)
  const auto pair_cell = [&](int r, int c, int& t, int& xs) {  // a live pair inside the frame: its owner tile, its column in the sheared frame
    const uint16_t pk = packed[(size_t)c * xmap_h + r];
    if (!pk) return false;
    int fc = (int)(short)((int)(pk & ((1 << OWN_XP_BITS) - 1)) - x_off);
    if (fc < 0) fc += rect_w;
    if (fc < 0 || fc >= rect_w) return false;
    t = (c - (int)(pk >> OWN_XP_BITS)) / W;
    xs = fc + bias + (((r >> 3) * m) >> 12);
    return true;
  };
  std::vector<std::vector<int>> cols((size_t)nt * n_ent);  // sorted, unique: the sheared columns a tile owns in an entry
  for (int r = r_lo; r <= r_hi; ++r)
    for (int c = 0; c < xmap_w; ++c) {
      int t, xs;
      if (pair_cell(r, c, t, xs)) cols[(size_t)t * n_ent + ((r - r_lo) >> sh)].push_back(xs);
    }
  for (auto& v : cols) {
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
  }
  const auto best_window = [](const std::vector<int>& v, int nx, int& start) {  // most columns inside [start, start + nx)
    size_t best = 0, j = 0;
    start = v.empty() ? 0 : v[0];
    for (size_t i = 0; i < v.size(); ++i) {
      while (j < v.size() && v[j] < v[i] + nx) ++j;
      if (j - i > best) {
        best = j - i;
        start = v[i];
      }
    }
    return best;
  };
  std::vector<size_t> extras_at(nx_cap + 1, 0);
  for (int nx = 1; nx <= nx_cap; ++nx)
    for (const auto& v : cols) {
      int st;
      extras_at[nx] += v.size() - best_window(v, nx, st);
    }
  int NX = nx_cap;
  while (NX > 1 && extras_at[NX - 1] <= extras_at[nx_cap] + extras_at[nx_cap] / 8 + (grouped ? 8 : 64)) NX -= 1;
  std::vector<int4>& tiles = pl.tiles;
  std::vector<int16_t>& bases = pl.bases;
  tiles.assign(nt, make_int4(0, 0, 0, 0));
  bases.assign((size_t)nt * n_ent, 0);
  for (size_t i = 0; i < cols.size(); ++i) {
    int st;
    best_window(cols[i], NX, st);
    bases[i] = (int16_t)st;
  }
  std::vector<u32>& masks = pl.masks;
  std::vector<uint16_t>& xextra = pl.xextra;
  masks.assign((size_t)nt * n_ent, 0);
  xextra.assign((size_t)xmap_w * xmap_h, 0);  // at EVERY pair whose cell is an extra: the cell's slot + 1
  std::vector<std::vector<u32>> extra_cells(nt);
  std::vector<std::map<u32, uint16_t>> extra_ids(nt);
  for (int r = r_lo; r <= r_hi; ++r)
    for (int c = 0; c < xmap_w; ++c) {
      int t, xs;
      if (!pair_cell(r, c, t, xs)) continue;
      const size_t e = (size_t)t * n_ent + ((r - r_lo) >> sh);
      const int k = xs - bases[e];
      if (k >= 0 && k < NX) {
        masks[e] |= 1u << k;
        tiles[t].x = std::max(tiles[t].x, k + 1);
      } else {
        const u32 cell = (u32)xs * (u32)rect_h + (u32)r;
        auto it = extra_ids[t].find(cell);
        if (it == extra_ids[t].end()) {
          extra_cells[t].push_back(cell);
          if (extra_cells[t].size() > 4096) return;  // (a wild X-map: the packed keys stay)
          it = extra_ids[t].emplace(cell, (uint16_t)extra_cells[t].size()).first;
        }
        xextra[(size_t)c * xmap_h + r] = it->second;
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
  // (wide tiles run as blocks of 512 threads, two per CU by their registers: 60 KB each is no limit -- and every further pass costs them 5-8 %)
  const size_t lds_target = grouped && W > 8 ? (size_t)60 * 1024 : OWN_LDS_TARGET;
  while (passes < OWN_MAX_ROW_PASSES && own_plan_lds_bytes(nxs_max, rows_per_pass(passes), hrp, extra_max, grouped) > lds_target) passes += 1;
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
  if (own_plan_lds_bytes(nxs_max, rp, hrp, extra_max, grouped) > 60 * 1024) return;  // LDS per block
  extra_flat.push_back(0);
  pl.W = W; pl.halo = halo; pl.r_lo = r_lo; pl.hr = hr; pl.hrp = hrp; pl.rp = rp; pl.nxs_max = nxs_max; pl.extra_max = extra_max;
  pl.m = m; pl.bias = bias; pl.extra_cols = extra; pl.delta_max = delta_max; pl.all_in = all_in; pl.grouped = grouped;
  pl.ok = true;
}

// The rig's plans by falling tile width (a frame takes the first whose mean tile fits one event pass of a block: cols_width);
// !ok entries behind the last one.  Default: ownership per 8-row group at 20 and at 16 columns (the halo of 4-7 columns costs
// 1.35 x / 1.44 x event reads, against 1.9 x at 8) where the rig allows it, then ownership per row at 8 columns.  "XM_OWN_W" /
// "XM_OWN_GROUPED=0" (experiments / tests): one plan, that width / per row only.
constexpr int OWN_PLANS = xm_handle::OWN_PLANS;
void own_plans(const xm_config* cfg, int xmap_h, int xr_min, OwnPlan (&out)[OWN_PLANS]) {
  for (OwnPlan& p : out) p = OwnPlan{};
  const char* eg = dbg_opt("XM_OWN_GROUPED");
  const char* ew = dbg_opt("XM_OWN_W");
  const bool try_grouped = !(eg && eg[0] == '0');
  if (ew) {
    if (try_grouped) own_plan_as(cfg, xmap_h, xr_min, true, atoi(ew), out[0]);
    if (!out[0].ok) own_plan_as(cfg, xmap_h, xr_min, false, atoi(ew), out[0]);
    return;
  }
  int n = 0;
  if (try_grouped)
    for (int W : {20, 16, 12}) {
      if (n >= OWN_PLANS - 1 || (n == 1 && W < 16)) break;  // (12 columns: only as the widest that fits; 1.58 x event reads lose to per row)
      own_plan_as(cfg, xmap_h, xr_min, true, W, out[n]);
      if (out[n].ok) n += 1;
    }
  own_plan_as(cfg, xmap_h, xr_min, false, 8, out[n]);
  if (out[n].ok) n += 1;
  else if (n == 0 && try_grouped) own_plan_as(cfg, xmap_h, xr_min, true, 8, out[0]);  // (no per-row plan: the narrow grouped one, if any)
}

// a plan's device tables and geometry into the kernel argument
void own_apply(const xm_handle::OwnSet& os, DevTables& tb) {
  tb.xmap_own = os.d_xmap_own;
  tb.xmap_extra = os.d_xmap_extra;
  tb.own_tiles = os.d_tiles;
  tb.own_bm = os.d_bm;
  tb.own_extra_cells = os.d_extra_cells;
  tb.own_r_lo = os.r_lo;
  tb.own_hr = os.hr;
  tb.own_hrp = os.hrp;
  tb.own_rp = os.rp;
  tb.own_grouped = os.grouped;
  tb.own_nxs_max = os.nxs_max;
  tb.own_extra_max = os.extra_max;
}

// Returns XM_OK whether or not the rig qualifies (h->own_mode says); an error only for HIP failures.
int own_setup(xm_handle* h, const xm_config* cfg, int xr_min) {
  OwnPlan pls[OWN_PLANS];
  own_plans(cfg, h->tb.xmap_h, xr_min, pls);
  if (!pls[0].ok) return XM_OK;
  const auto up = [](auto** dst, const auto& v) -> hipError_t {
    typedef typename std::remove_reference<decltype(v)>::type::value_type E;
    hipError_t e = hipMalloc((void**)dst, v.size() * sizeof(E) + 64);
    return e != hipSuccess ? e : hipMemcpy(*dst, v.data(), v.size() * sizeof(E), hipMemcpyHostToDevice);
  };
  for (int i = 0; i < OWN_PLANS; ++i) {
    const OwnPlan& pl = pls[i];
    if (!pl.ok) continue;
    xm_handle::OwnSet& os = h->own[i];
    HIP_TRY(up(&os.d_xmap_own, pl.packed));
    HIP_TRY(up(&os.d_xmap_extra, pl.xextra));
    HIP_TRY(up(&os.d_tiles, pl.tiles));
    {  // per tile ONE table, read by 16-byte loads at the head of the tile.  Per-row ownership: per row the band's first column before
       // the frame's shear (an event's band column = cell column - that) | the row's ownership mask << 16.  Per 8-row group: per
       // group that column | the band's first column in the sheared frame << 16, then per group the band columns the tile owns.
      const int n_ent = pl.hrp >> (pl.grouped ? 3 : 0), words = own_tab_words(pl.hrp, pl.grouped), nt = (int)pl.tiles.size();
      std::vector<u32> tab((size_t)nt * words, 0);
      for (int t = 0; t < nt; ++t)
        for (int e = 0; e < n_ent; ++e) {
          const size_t k = (size_t)t * n_ent + e;
          const int row = (pl.grouped ? e * 8 : e) + pl.r_lo;
          const int org = (int)pl.bases[k] - pl.bias - (((row >> 3) * pl.m) >> 12);
          if (pl.grouped) {
            tab[(size_t)t * words + e] = (u32)(uint16_t)(int16_t)org | ((u32)(uint16_t)pl.bases[k] << 16);
            tab[(size_t)t * words + n_ent + e] = pl.masks[k];
          } else {
            tab[(size_t)t * words + e] = (u32)(uint16_t)(int16_t)org | (pl.masks[k] << 16);
          }
        }
      HIP_TRY(up(&os.d_bm, tab));
    }
    HIP_TRY(up(&os.d_extra_cells, pl.extra_flat));
    os.extras = (int)pl.extra_flat.size() - 1;
    os.w = pl.W; os.halo = pl.halo; os.all_in = pl.all_in;
    os.r_lo = pl.r_lo; os.hr = pl.hr; os.hrp = pl.hrp; os.rp = pl.rp; os.grouped = pl.grouped ? 1 : 0;
    os.nxs_max = pl.nxs_max; os.extra_max = pl.extra_max;
    os.ok = true;
  }
  // (the frame's shear is the rig's: the same in every plan)
  h->tb.shear_m = pls[0].m;
  h->tb.shear_bias = pls[0].bias;
  h->tb.shear_extra = pls[0].extra_cols;
  own_apply(h->own[0], h->tb);  // (K2 and the other kernels read none of these fields; a launch of K0b / K1 applies its frame's plan)
  h->own_mode = true;
  if (const char* e = dbg_opt("XM_OWN_EPT")) h->own_ept_forced = atoi(e);
  if (pls[0].all_in) h->cols_flags |= COLS_F_ALL_IN_FRAME;
  return XM_OK;
}


}  // namespace
