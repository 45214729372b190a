/* A host that is neither Python nor C++: plain C99 against include/xmaps.h, linked with libxmaps_hip.so only (no HIP headers, no
 * torch in the process: the library runs on ROCm's own HIP runtime and looks librccl up itself).  tests/test_gpu_c_host.py writes
 * the rig's tables and one frame into a blob, this program runs the frame
 *   1. as one rank of a world of one through the library's shard communicator (xm_shard_comm_*: columns merge, then packed keys),
 *   2. through xm_create_sharded over device 0 (host columns in, host frames out), then over 2 / 4 / 8 virtual ranks on it,
 *   3. through xm_process_frame (the plain single-GPU entry),
 * and writes the depth / BGR frames of each back; the test compares them with the oracle's.
 *
 * blob: int32 header[16] = {cam_w, cam_h, proj_w, proj_h, rect_w, rect_h, xmap_w, xmap_h, x_offset, n_events, ...0},
 *       double p03, float z_near, float z_far, then int16 cam_mapx[cam_h*cam_w], cam_mapy[..], proj_x_map[xmap_h*xmap_w],
 *       disp_proj_mapxy[proj_h*proj_w*2], then uint16 x[n], uint16 y[n], int64 t[n]. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "xmaps.h"

#define CHECK(call)                                                                   \
  do {                                                                                \
    int rc_ = (call);                                                                 \
    if (rc_ != XM_OK) {                                                               \
      fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #call, rc_, xm_last_error()); \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

static void* read_exact(FILE* f, size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (!p || fread(p, 1, bytes, f) != bytes) {
    fprintf(stderr, "short read (%zu bytes)\n", bytes);
    exit(2);
  }
  return p;
}

static int write_frames(const char* path, const float* depth, const uint8_t* bgr, size_t px) {
  FILE* f = fopen(path, "wb");
  if (!f) return 1;
  fwrite(depth, 4, px, f);
  fwrite(bgr, 1, px * 3, f);
  fclose(f);
  return 0;
}

int main(int argc, char** argv) {
  if (argc != 3) {
    fprintf(stderr, "usage: %s blob out_prefix\n", argv[0]);
    return 2;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t* hd = (int32_t*)read_exact(f, 16 * sizeof(int32_t));
  double* p03 = (double*)read_exact(f, 8);
  float* zz = (float*)read_exact(f, 8);
  const size_t cam = (size_t)hd[0] * hd[1], proj = (size_t)hd[2] * hd[3], xm = (size_t)hd[6] * hd[7], n = (size_t)hd[9];
  int16_t* mapx = (int16_t*)read_exact(f, cam * 2);
  int16_t* mapy = (int16_t*)read_exact(f, cam * 2);
  int16_t* xmap = (int16_t*)read_exact(f, xm * 2);
  int16_t* pmap = (int16_t*)read_exact(f, proj * 2 * 2);
  uint16_t* x = (uint16_t*)read_exact(f, n * 2);
  uint16_t* y = (uint16_t*)read_exact(f, n * 2);
  int64_t* t = (int64_t*)read_exact(f, n * 8);
  fclose(f);

  xm_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg;
  cfg.device = 0;
  cfg.cam_width = hd[0]; cfg.cam_height = hd[1];
  cfg.proj_width = hd[2]; cfg.proj_height = hd[3];
  cfg.rect_width = hd[4]; cfg.rect_height = hd[5];
  cfg.xmap_width = hd[6]; cfg.xmap_height = hd[7];
  cfg.x_offset = hd[8];
  cfg.view = XM_VIEW_PROJECTOR;
  cfg.n_slots = 1;
  cfg.flags = XM_FLAG_TIME_SORTED;
  cfg.p03 = *p03;
  cfg.z_near = zz[0]; cfg.z_far = zz[1];
  cfg.cam_mapx_i16 = mapx; cfg.cam_mapy_i16 = mapy; cfg.proj_x_map = xmap; cfg.disp_proj_mapxy_i16 = pmap;

  float* depth = (float*)malloc(proj * 4);
  uint8_t* bgr = (uint8_t*)malloc(proj * 3);
  char path[1024];
  xm_handle* h = NULL;
  CHECK(xm_create(&cfg, &h));

  /* 1. one rank of a world of one: the library's own RCCL communicator */
  {
    unsigned char id[XM_SHARD_COMM_ID_BYTES];
    xm_shard_comm* c = NULL;
    int cols = 0, failed = 0;
    size_t cap = 0;
    void *dx, *dy, *dt, *dd, *db;
    CHECK(xm_shard_comm_id(id));
    CHECK(xm_shard_comm_create(h, id, 0, 1, n, &c));
    CHECK(xm_shard_comm_info(c, &cols, &cap, NULL, NULL));
    const size_t hr = cols ? cap + 8 : 0; /* headroom in front of the shard, 8 events of slack behind */
    CHECK(xm_dev_alloc(h, (hr + n + 8) * 2, &dx));
    CHECK(xm_dev_alloc(h, (hr + n + 8) * 2, &dy));
    CHECK(xm_dev_alloc(h, (hr + n + 8) * 8, &dt));
    CHECK(xm_dev_alloc(h, proj * 4, &dd));
    CHECK(xm_dev_alloc(h, proj * 3, &db));
    uint16_t *ox = (uint16_t*)dx + hr, *oy = (uint16_t*)dy + hr;
    int64_t* ot = (int64_t*)dt + hr;
    CHECK(xm_dev_upload(h, ox, x, n * 2));
    CHECK(xm_dev_upload(h, oy, y, n * 2));
    CHECK(xm_dev_upload(h, ot, t, n * 8));
    if (cols) {
      CHECK(xm_shard_comm_frame(c, ox, oy, ot, n, (float*)dd, (uint8_t*)db));
      CHECK(xm_shard_comm_failed(c, &failed)); /* (synchronises) */
      CHECK(xm_dev_download(h, depth, dd, proj * 4));
      CHECK(xm_dev_download(h, bgr, db, proj * 3));
      snprintf(path, sizeof path, "%s.comm_columns", argv[2]);
      if (failed || write_frames(path, depth, bgr, proj)) return 3;
    }
    CHECK(xm_shard_comm_frame_keys(c, ox, oy, ot, NULL, n, XM_T_INT64, 0, (float*)dd, (uint8_t*)db));
    CHECK(xm_sync(h));
    CHECK(xm_dev_download(h, depth, dd, proj * 4));
    CHECK(xm_dev_download(h, bgr, db, proj * 3));
    snprintf(path, sizeof path, "%s.comm_keys", argv[2]);
    if (write_frames(path, depth, bgr, proj)) return 3;
    printf("comm: columns=%d cap=%zu\n", cols, cap);
    xm_shard_comm_destroy(c);
    CHECK(xm_dev_free(h, dx)); CHECK(xm_dev_free(h, dy)); CHECK(xm_dev_free(h, dt)); CHECK(xm_dev_free(h, dd)); CHECK(xm_dev_free(h, db));
  }

  /* 2. xm_create_sharded over device 0: host columns in, host frames out */
  {
    xm_sharded* s = NULL;
    const int dev = 0;
    int nd = 0, rccl = 0;
    uint64_t fc = 0, fk = 0, fr = 0;
    xm_frame_stats st;
    CHECK(xm_create_sharded(&dev, 1, &cfg, &s));
    CHECK(xm_sharded_info(s, &nd, &rccl, NULL));
    CHECK(xm_sharded_process_frame(s, x, y, t, NULL, n, XM_T_INT64, depth, bgr, &st));
    CHECK(xm_sharded_stats(s, &fc, &fk, &fr));
    snprintf(path, sizeof path, "%s.sharded", argv[2]);
    if (write_frames(path, depth, bgr, proj)) return 3;
    printf("sharded: n_dev=%d rccl=%d columns=%llu keys=%llu redone=%llu t=[%.0f, %.0f]\n", nd, rccl, (unsigned long long)fc,
           (unsigned long long)fk, (unsigned long long)fr, st.t_min, st.t_max);
    xm_sharded_destroy(s);
  }

  /* 2b. the same entry with 2 / 4 / 8 VIRTUAL ranks on device 0 (xm_debug_option: test switch of the library): the N > 1
   *     orchestration -- shard bounds, gathered headers, predecessor columns, merge -- without a second GPU */
  {
    const int worlds[3] = {2, 4, 8};
    for (int k = 0; k < 3; ++k) {
      xm_sharded* s = NULL;
      const int dev = 0;
      int nd = 0, rccl = 0;
      uint64_t fc = 0, fk = 0, fr = 0;
      char w[8];
      snprintf(w, sizeof w, "%d", worlds[k]);
      CHECK(xm_debug_option("XM_SHARD_FAKE_RANKS", w));
      CHECK(xm_create_sharded(&dev, 1, &cfg, &s));
      CHECK(xm_debug_option("XM_SHARD_FAKE_RANKS", NULL));
      CHECK(xm_sharded_info(s, &nd, &rccl, NULL));
      CHECK(xm_sharded_process_frame(s, x, y, t, NULL, n, XM_T_INT64, depth, bgr, NULL));
      CHECK(xm_sharded_stats(s, &fc, &fk, &fr));
      snprintf(path, sizeof path, "%s.sharded_w%d", argv[2], worlds[k]);
      if (write_frames(path, depth, bgr, proj)) return 3;
      printf("virtual: n_dev=%d rccl=%d columns=%llu keys=%llu redone=%llu\n", nd, rccl, (unsigned long long)fc, (unsigned long long)fk,
             (unsigned long long)fr);
      xm_sharded_destroy(s);
    }
  }

  /* 3. the plain entry */
  {
    xm_frame_stats st;
    CHECK(xm_process_frame(h, x, y, t, NULL, n, XM_T_INT64, XM_MEM_HOST, depth, bgr, &st));
    snprintf(path, sizeof path, "%s.single", argv[2]);
    if (write_frames(path, depth, bgr, proj)) return 3;
    printf("single: inliers=%llu\n", (unsigned long long)st.n_inliers);
  }
  xm_destroy(h);
  return 0;
}
