/*
 * xmaps_oracle.c -- CPU oracle in plain C for the X-maps hot path.  TEST INFRASTRUCTURE ONLY: used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg; never linked into or called from the product.
 *
 * One fused scalar pass per event + row-parallel frame stages; OpenMP optional (-fopenmp) = "what the
 * reference could reach if every stage used all host cores like its Numba prange kernels do".
 * Restates (file:line relative to /root/reference/python):
 *   rectify LUT gather            cam_proj_calibration.py:277-281
 *   time normalise, X-map gather, int16 disparity, inlier masks     x_maps_disparity.py:9-32
 *   last-writer-wins scatter      cam_proj_calibration.py:299-303 (projector view), 312-317 (camera view)
 *   7x7 dilate + nearest remap    disp_to_depth.py:76-97 (OpenCV semantics restated; parity unpinned)
 *   disparity -> depth            disp_to_depth.py:46-63
 *   clip/normalise u8, Turbo BGR + white mask     disp_to_depth.py:7-43
 * Pinned against the golden vectors and the NumPy oracle in tests/test_oracle_c.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  int cam_w, cam_h, proj_w, proj_h, rect_w, rect_h, xmap_w, xmap_h, x_offset, camera_view;
  double p03;
  float z_near, z_far;
  const int16_t *mapx, *mapy, *xmap, *pmapxy;
  const uint8_t* turbo_bgr; /* 256 x 3 */
} xmo_tables;

int xmo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* bench.py's all-core leg: one thread per PHYSICAL core (256 SMT threads spinning between six short parallel regions per frame
 * ran anywhere between 11 and 108 Mev/s on the same box).  No-op without OpenMP. */
void xmo_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* scratch of the frame stages, kept between calls (a fresh 9-28 MB malloc per frame is a page fault per 4 KB inside the parallel
 * regions); the checker has one caller at a time */
static float* g_scratch[2] = {NULL, NULL};
static int64_t g_scratch_cells[2] = {0, 0};
static float* scratch(int which, int64_t cells) {
  if (g_scratch_cells[which] < cells) {
    free(g_scratch[which]);
    g_scratch[which] = (float*)malloc(sizeof(float) * (size_t)cells);
    g_scratch_cells[which] = g_scratch[which] ? cells : 0;
  }
  return g_scratch[which];
}

/* returns 0, or -1 if an index fell outside a table/frame (NumPy would raise IndexError).
 * Outputs (any may be NULL): key_frame u64 [H][W] scratch supplied by the caller (zeroed here),
 * disp_map f32 [H][W] (rect frame or camera frame), depth f32 / bgr u8 of the OUTPUT frame,
 * mask u8 [n], disp_ev int16 [n] (0 where masked), n_inliers. */
int xmo_process_frame(const xmo_tables* tb, const uint16_t* x, const uint16_t* y, const int64_t* t, int64_t n,
                      uint64_t* key_frame, float* disp_map, float* depth, uint8_t* bgr, uint8_t* mask,
                      int16_t* disp_ev, int64_t* n_inliers) {
  const int fw = tb->camera_view ? tb->cam_w : tb->rect_w, fh = tb->camera_view ? tb->cam_h : tb->rect_h;
  const int64_t cells = (int64_t)fw * fh;
  int64_t tmin = INT64_MAX, tmax = INT64_MIN, inl = 0;
  int err = 0;
#pragma omp parallel for reduction(min : tmin) reduction(max : tmax) schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    if (t[i] < tmin) tmin = t[i];
    if (t[i] > tmax) tmax = t[i];
  }
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < cells; ++c) key_frame[c] = 0;
  const double den = (double)(tmax - tmin), S = (double)(tb->xmap_w - 1);
#pragma omp parallel for reduction(+ : inl) reduction(| : err) schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    if (mask) mask[i] = 0;
    if (disp_ev) disp_ev[i] = 0;
    if (x[i] >= tb->cam_w || y[i] >= tb->cam_h) { err |= 1; continue; }
    const int64_t px = (int64_t)y[i] * tb->cam_w + x[i];
    const int xr = tb->mapx[px], yr = tb->mapy[px];
    int ts = 0;
    if (tmax != tmin) ts = (int)(int16_t)(int)rint(((double)(t[i] - tmin) / den) * S);
    if (yr < 0 || yr >= tb->xmap_h - 1) continue;
    const int xp = tb->xmap[(int64_t)yr * tb->xmap_w + ts];
    const int disp = (int16_t)(xp - xr - tb->x_offset);
    if (disp < 0) continue;
    int64_t cell;
    if (tb->camera_view) {
      cell = px;
    } else {
      int col = (int16_t)(xr + disp);
      if (col < 0) col += fw;
      if (col < 0 || col >= fw || yr >= fh) { err |= 1; continue; }
      cell = (int64_t)yr * fw + col;
    }
    if (mask) mask[i] = 1;
    if (disp_ev) disp_ev[i] = (int16_t)disp;
    ++inl;
    /* last writer wins == largest event index wins; atomic max keeps that exact under OpenMP */
    const uint64_t key = ((uint64_t)(i + 1) << 16) | (uint64_t)(uint16_t)disp;
    uint64_t cur = __atomic_load_n(&key_frame[cell], __ATOMIC_RELAXED);
    while (cur < key && !__atomic_compare_exchange_n(&key_frame[cell], &cur, key, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
  }
  if (n_inliers) *n_inliers = inl;

  const int ow = tb->camera_view ? tb->cam_w : tb->proj_w, oh = tb->camera_view ? tb->cam_h : tb->proj_h;
  float* frame = disp_map ? disp_map : scratch(0, cells);
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < cells; ++c) frame[c] = key_frame[c] ? (float)(key_frame[c] & 0xffff) : 0.0f;

  float* rowmax = NULL;
  if (!tb->camera_view) { /* separable 7x7 max, then nearest gather */
    rowmax = scratch(1, cells);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < fh; ++r)
      for (int c = 0; c < fw; ++c) {
        float m = 0.0f;
        const int c0 = c - 3 < 0 ? 0 : c - 3, c1 = c + 3 >= fw ? fw - 1 : c + 3;
        for (int k = c0; k <= c1; ++k) m = frame[(int64_t)r * fw + k] > m ? frame[(int64_t)r * fw + k] : m;
        rowmax[(int64_t)r * fw + c] = m;
      }
  }
  const float range = tb->z_far - tb->z_near;
#pragma omp parallel for schedule(static)
  for (int v = 0; v < oh; ++v)
    for (int u = 0; u < ow; ++u) {
      const int64_t o = (int64_t)v * ow + u;
      float d = 0.0f;
      if (tb->camera_view) {
        d = frame[o];
      } else {
        const int mx = tb->pmapxy[2 * o], my = tb->pmapxy[2 * o + 1];
        if (mx >= 0 && mx < fw && my >= 0 && my < fh) {
          const int r0 = my - 3 < 0 ? 0 : my - 3, r1 = my + 3 >= fh ? fh - 1 : my + 3;
          for (int r = r0; r <= r1; ++r) d = rowmax[(int64_t)r * fw + mx] > d ? rowmax[(int64_t)r * fw + mx] : d;
        }
      }
      float z = 0.0f;
      if (d != 0.0f) {
        double q = tb->p03 / (double)d;
        z = (float)(q > 1e-9 ? q : 1e-9);
      }
      if (depth) depth[o] = z;
      if (bgr) {
        unsigned u8 = 0;
        if (z != 0.0f) {
          float c = z < tb->z_far ? z : tb->z_far;
          c = c > tb->z_near ? c : tb->z_near;
          const float qn = (c - tb->z_near) / range;
          u8 = (unsigned)(int)((double)qn * 255.0) & 0xff;
        }
        if (u8 == 0) {
          bgr[3 * o] = bgr[3 * o + 1] = bgr[3 * o + 2] = 255;
        } else {
          memcpy(bgr + 3 * o, tb->turbo_bgr + 3 * u8, 3);
        }
      }
    }
  return err ? -1 : 0;
}

/* ---- activity-noise filter: the sequential definition of oracle/ingest_oracle.py:ActivityFilterOracle.process, in C so that
 * streams of ESL size (and bench.py's cpu legs) can be judged in milliseconds; pinned against the Python form in
 * tests/test_oracle_ingest.py.  rec: 16-byte EventCD records (x:u16@0, y:u16@2, p:i16@4, t:i64@8); every record takes part.
 * last / has: the per-pixel history the caller keeps between packets (int64 / uint8, h x w).  Returns the number kept. */
int64_t xmo_activity_filter2(const void* rec, int64_t n, int w, int h, int64_t thresh, int64_t* last, uint8_t* has, uint8_t* keep, int include_self) {
  const uint8_t* r = (const uint8_t*)rec;
  int64_t kept = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int x = *(const uint16_t*)(r + 16 * i), y = *(const uint16_t*)(r + 16 * i + 2);
    const int64_t t = *(const int64_t*)(r + 16 * i + 8);
    uint8_t k = 0;
    const int y0 = y > 0 ? y - 1 : 0, y1 = y + 1 < h ? y + 1 : h - 1, x0 = x > 0 ? x - 1 : 0, x1 = x + 1 < w ? x + 1 : w - 1;
    for (int yy = y0; yy <= y1; ++yy)
      for (int xx = x0; xx <= x1; ++xx)
        if ((include_self || yy != y || xx != x) && has[(int64_t)yy * w + xx] && t - last[(int64_t)yy * w + xx] <= thresh) k = 1;
    keep[i] = k;
    kept += k;
    const int64_t c = (int64_t)y * w + x;
    if (!has[c] || t > last[c]) last[c] = t;
    has[c] = 1;
  }
  return kept;
}
int64_t xmo_activity_filter(const void* rec, int64_t n, int w, int h, int64_t thresh, int64_t* last, uint8_t* has, uint8_t* keep) {
  return xmo_activity_filter2(rec, n, w, h, thresh, last, has, keep, 0);  /* (the rule as defined: the own pixel does not count) */
}
