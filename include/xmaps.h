/*
 * xmaps.h -- C-ABI of the MI355X-native X-maps hot path (libxmaps_hip.so).
 *
 * Plain C, plain pointers and sizes, no torch / C++ types.  Every entry point returns XM_OK (0) or a
 * negative error code and never throws; xm_last_error() gives the text for the calling thread.
 *
 * What it replaces (all file:line relative to the reference repo, python/ directory):
 *   xm_create            <- table setup consumed by the hot path: CamProjMaps LUTs
 *                           (cam_proj_calibration.py:246-270), XMapsDisparity.proj_x_map
 *                           (x_maps_disparity.py:44-67), DisparityToDepth (disp_to_depth.py:66-74)
 *   xm_process_frame*    <- DepthReprojectionPipe.process_ev_frame (depth_reprojection_pipe.py:121-167):
 *                           rectify_cam_coords_i16 -> compute_event_disparity ->
 *                           compute_disp_map_{projector,camera}_view -> remap_rectified_disp_map_to_proj
 *                           -> colorize_depth_from_disp, fused into three kernels
 *   xm_stage_*           <- the same stages one by one, with the reference's per-stage signatures
 *                           (cam_proj_calibration.py:277-281,299-317; x_maps_disparity.py:9-32,69-82;
 *                            disp_to_depth.py:46-63,76-115)
 *   xm_shard_*           <- (new) the event buffer sharded by index over several GPUs; the caller
 *                           max-reduces the packed key frame between xm_shard_scatter and
 *                           xm_shard_finish (RCCL all-reduce, MAX on int64)
 *
 * Threading: one handle = one device + its own HIP streams; a handle is not thread-safe.
 * Ownership: the caller owns every pointer it passes; tables are copied to the device in xm_create.
 */
#ifndef XMAPS_H
#define XMAPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XM_API_VERSION 5  /* 5: xm_activity_set_rule, XM_INGEST_ACT_SELF (round 6) */

/* error codes */
#define XM_OK 0
#define XM_ERR_INVALID (-1)  /* bad argument / config                                              */
#define XM_ERR_HIP (-2)      /* a HIP runtime call failed (xm_last_error has the hipError string)   */
#define XM_ERR_NOMEM (-3)
#define XM_ERR_INDEX (-4)    /* >= 1 event indexed outside a table/frame: where NumPy raises
                                IndexError in the reference.  The frame is still produced with the
                                offending events dropped; stats.n_index_errors counts them.        */
#define XM_ERR_TOO_MANY (-5) /* more events in one frame than the key's index field can hold       */
#define XM_ERR_UNSORTED (-6) /* XM_FLAG_TIME_SORTED was set but a frame processed asynchronously since the
                                last xm_sync() was not sorted by t: its output is invalid (see the flag)  */

/* xm_config.flags */
#define XM_FLAG_TIME_SORTED 1u /* The caller declares every frame sorted by t -- true for the frames the trigger
                                  finder cuts out of the camera stream (trigger_finder.py:172) unless a frame event
                                  filter re-ordered them.  Then (tmin, tmax) = (t[0], t[n-1]) and the extrema pass
                                  (K0, 8 B/event re-read) is skipped.  The declaration is VERIFIED on the device for
                                  every frame: synchronous calls (XM_MEM_HOST) transparently redo an unsorted frame on
                                  the general path (result always exact); asynchronous calls report it through
                                  xm_frame_stats.n_unsorted and xm_sync() -> XM_ERR_UNSORTED.  Ignored when a polarity
                                  column is given. */

#define XM_FLAG_TRY_SORTED 2u  /* THE DEFAULT since API version 2 (the bit is accepted and ignored).  No promise from the caller:
                                  every frame is first run with (tmin, tmax) = (t[0], t[n-1]) (no extrema pass, 8 B/event
                                  less HBM traffic, one launch less) while the device checks that every event lies inside that
                                  range; a frame for which that fails is redone on the general path automatically -- inside the
                                  call for XM_MEM_HOST, otherwise when its slot comes round again (n_slots calls later) or in
                                  xm_sync(), whichever is first.  Results are always exact.  Contract for asynchronous
                                  calls: a frame's input and output buffers stay untouched by the caller until n_slots
                                  further frames have been submitted or xm_sync() has returned (the redo reads the inputs
                                  again and rewrites the outputs).  Not used when a polarity column is given, inside
                                  xm_graph_create, and for frames too sparse for the tiled kernel: those run the extrema pass. */
#define XM_FLAG_GENERAL 16u    /* Run the extrema pass (K0) on every frame instead of the verified shortcut above: one more
                                  launch and 8 B/event more traffic, but no frame is ever run twice -- for streams that are
                                  known to be unsorted (raster-ordered frame filters). */

#define XM_FLAG_DEFAULT_STREAMS 4u /* Create the slots' streams at the default priority.  By default they are created at the
                                     highest priority, which gives them hardware queues of their own (HIP multiplexes all
                                     streams of one priority, the application's included, onto 4 queues; sharing them cost
                                     up to 17 % of the pipelined frame rate).  hipGraph replays (xm_graph_*) however lose
                                     their branch concurrency when the process owns non-default-priority streams: set this
                                     flag on handles that replay graphs. */

#define XM_FLAG_ADAPTIVE_BATCH 32u /* Asynchronous device-pointer frames (XM_MEM_DEVICE, stats == NULL) are submitted as GROUPS -- one
                                    * set of multi-frame launches, the path xm_process_batch takes -- whenever the GPU is still busy:
                                    * a frame is launched at once when no group is in flight (an idle GPU, e.g. the 60 Hz live pipe,
                                    * never waits); otherwise it is held back and goes out together with the frames that follow --
                                    * with the first call that finds the GPU idle, when n_slots / 4 (at most 32) frames are held, or
                                    * at the next synchronising call (xm_sync, any synchronous call, xm_destroy).  Needs n_slots >= 8.
                                    * A group holds frames of ONE layout (AoS or SoA, one t_dtype, polarity column or not): a frame of
                                    * another layout closes the pending group first.  Contract of asynchronous calls on such a handle:
                                    * inputs and outputs of a frame stay untouched until xm_sync() has returned or n_slots + n_slots / 4
                                    * further frames have been submitted (a slot's previous frame is resolved -- and, if its time-sorted
                                    * shortcut failed, redone from its inputs -- when the GROUP that reuses the slot is launched, and up
                                    * to n_slots / 4 - 1 frames can be held in front of it).  Held frames are not on any stream yet:
                                    * synchronising xm_stream() alone does not launch them, xm_sync() does. */
#define XM_FLAG_LAUNCH_WORKERS 8u /* One launch thread per slot stream: asynchronous device-pointer calls (XM_MEM_DEVICE) only post
                                    a job (~5 us per call instead of ~11.5 us for the three kernel launches); everything else
                                    waits for the workers to be idle first, so ordering and results are unchanged.  Does not
                                    raise the frame rate (the GPU bounds it); for hosts whose calling thread has other work. */

/* view (RuntimeParams.camera_perspective, depth_reprojection_processor.py:34) */
#define XM_VIEW_PROJECTOR 0
#define XM_VIEW_CAMERA 1

/* where the event / output buffers of a call live */
#define XM_MEM_HOST 0   /* host pointers: staged H2D/D2H inside the call, call returns synchronised  */
#define XM_MEM_DEVICE 1 /* device pointers: everything is enqueued on the handle's stream; xm_sync() */
#define XM_MEM_HOST_PINNED 2 /* PINNED host pointers (xm_host_alloc / hipHostMalloc / hipHostRegister): asynchronous like
                                XM_MEM_DEVICE -- H2D copies, the three kernels and the D2H copies are enqueued on the next
                                slot's stream and the call returns; with n_slots > 1 the copies of one frame overlap the
                                kernels of another.  Buffers must stay valid and untouched until xm_sync(). */

/* dtype of the time column: int64 microseconds (Metavision EventCD) or an already-normalised
 * float time surface (python/eval/compute_depth_x_maps.py:89-96) */
#define XM_T_INT64 0
#define XM_T_FLOAT32 1
#define XM_T_FLOAT64 2

/* packed last-writer-wins key:  [63]=0 | tag:19 | event index:28 | disparity:16 */
#define XM_KEY_DISP_BITS 16
#define XM_KEY_IDX_BITS 28
#define XM_KEY_TAG_BITS 19
#define XM_KEY_MAX_EVENTS (1ull << XM_KEY_IDX_BITS)

typedef struct xm_handle xm_handle;

typedef struct xm_config {
  uint32_t struct_size; /* = sizeof(xm_config), for forward compatibility */
  int32_t device;       /* HIP device ordinal */
  int32_t cam_width, cam_height;
  int32_t proj_width, proj_height;
  int32_t rect_width, rect_height; /* rectified frame (calib.rect_image_{width,height})            */
  int32_t xmap_width;              /* X_MAP_WIDTH; T_PX_SCALE = xmap_width - 1 (xmd:58-59)         */
  int32_t xmap_height;             /* rows of proj_x_map (0 => rect_height)                        */
  int32_t x_offset;                /* X_OFFSET = 4242 (xmd:49); must fit int16                     */
  int32_t view;                    /* XM_VIEW_*                                                    */
  int32_t n_slots;                 /* frames that may be in flight at once (>=1; own stream each)  */
  uint32_t flags;                  /* XM_FLAG_*                                                    */
  double p03;                      /* P2[0,3] (calib:207, d2d:104-107)                              */
  float z_near, z_far;             /* d2d:71-72                                                     */
  /* host tables, reference layouts (row-major int16); copied + re-packed for the device in xm_create */
  const int16_t* cam_mapx_i16;        /* [cam_height][cam_width]   disp_cam_mapx_i16 (calib:253)    */
  const int16_t* cam_mapy_i16;        /* [cam_height][cam_width]   disp_cam_mapy_i16 (calib:254)    */
  const int16_t* proj_x_map;          /* [xmap_height][xmap_width] proj_x_map (xmd:61-67)           */
  const int16_t* disp_proj_mapxy_i16; /* [proj_height][proj_width][2] (x,y) (calib:270); may be NULL
                                         for XM_VIEW_CAMERA                                         */
} xm_config;

typedef struct xm_frame_stats {
  uint64_t n_events;       /* events handed in                                                       */
  uint64_t n_used;         /* after the polarity mask (== n_events when p == NULL)                    */
  uint64_t n_inliers;      /* events that passed both inlier masks (xmd:23,29) and were scattered    */
  uint64_t n_index_errors; /* events that would raise IndexError in the reference                    */
  double t_min, t_max;     /* frame extrema of t (over the used events), as double                   */
  float gpu_ms[4];         /* HIP-event time of {minmax, scatter, frame kernel, start of first .. end of
                              last}; filled only by xm_profile_frame, else 0                           */
  uint64_t n_unsorted;     /* XM_FLAG_TIME_SORTED: > 0 if the frame was NOT sorted (wavefronts that saw an event
                              outside [t[0], t[n-1]]); synchronous calls have already redone such a frame    */
} xm_frame_stats;

/* ---- lifetime ---------------------------------------------------------------------------------- */
/* Variant switches for tests and experiments ("XM_COLS", "XM_K2_PIPE", "XM_K2_PIPE_PPT", "XM_K2_CONSEC", "XM_K2_NLDS_MAX",
 * "XM_K2_PPT", "XM_K2_FLAGS", "XM_K1_DIRECT", "XM_K2_DIRECT", "XM_KEY32", "XM_OWN_W", "XM_OWN_SHEAR", "XM_OWN_GROUPED",
 * "XM_OWN_ROW_PASSES", "XM_OWN_EPT", "XM_K2_PER_CU", "XM_K2_CHAIN", "XM_WORKERS", "XM_XMAP_SCAN",
 * "XM_INGEST_CLEAR_EVERY", "XM_INGEST_TRACE", "XM_INGEST_OUT_PIECE", "XM_INGEST_OUT_SERIAL", "XM_INGEST_OUT_INLINE",
 * "XM_INGEST_OUT_NO_QUERY", "XM_INGEST_EVT3_OUT_STREAM", "XM_INGEST_OWN_STREAMS", "XM_INGEST_PRIOS", "XM_SHARDED_KEYS"; values as text).  Process-wide, read when a handle / an ingest is created (XM_XMAP_SCAN:
 * at every call).  The library never reads them from the environment.  value == NULL removes an option, name == NULL all. */
int xm_debug_option(const char* name, const char* value);
int xm_api_version(void);
const char* xm_last_error(void);
int xm_create(const xm_config* cfg, xm_handle** out);
void xm_destroy(xm_handle* h);
int xm_sync(xm_handle* h); /* wait for everything enqueued on every slot of the handle; XM_ERR_UNSORTED see above */
/* Frames redone on the general path because a time-sorted shortcut (XM_FLAG_TIME_SORTED in synchronous calls,
 * XM_FLAG_TRY_SORTED always) did not hold, since xm_create. */
int xm_sorted_fallbacks(xm_handle* h, uint64_t* count);
/* Frames enqueued per K1 variant since xm_create: counts[0] general (K0 + 64-bit packed keys), counts[1] verified-sorted
 * shortcut on the 64-bit key frame, counts[2] compact 32-bit key frame, counts[3] column tiles + plain u16 disparity frame
 * (no atomics; needs an injective X-map -> frame-cell relation, checked in xm_create).  A redone frame counts twice. */
int xm_path_counts(xm_handle* h, uint64_t counts[4]);
/* Which no-atomics K1 the rig qualified for in xm_create, and its geometry: info[0] = 0 none (sparse / wild tables: packed
 * keys only), 1 column tiles (injective X-map), 2 owner tiles (several time columns per frame cell -- the reference's own
 * calibration, python/cam_proj_calibration.py:299-303 with X_MAP_WIDTH = projector_width, python/x_maps_disparity.py:58-59);
 * owner tiles: info[1] = time columns per tile, [2] = halo columns read behind them, [3] = widest cell band of a tile
 * (sheared frame columns), [4] = shear per 8-row group in 1/4096 columns, [5] = extra frame columns of the sheared
 * u16 frame, [6] = first row / [7] = rows the rectify LUT can reach, [8] = owner cells outside their tile's band ("extras":
 * one slot each, flushed one by one), [9] = the most extras of one tile -- all of the plan frames take by default (ownership per
 * 8-row group on wide tiles where the rig allows it); [10], [11] = tile width and halo of the second plan (ownership per row, for
 * frames too dense for the first plan's tiles), 0 = none. */
int xm_cols_info(xm_handle* h, int32_t info[12]);
/* The owner-tile analysis of xm_create on its own (host code, no device needed): would this rig's tables qualify?  info as
 * xm_cols_info (info[0] = 2 or 0), plus [10] = the largest distance of a time column from its cell's first column, [11] = LDS
 * bytes per tile.  (The injective case, info[0] = 1, is decided on the device in xm_create.) */
int xm_own_plan_info(const xm_config* cfg, int32_t info[12]);

/* ---- the fused hot path: one projector frame of events -> depth frame (+ BGR) -------------------- */
/*
 * SoA events x[n], y[n], t[n] (dtype t_dtype), optional p[n] (NULL = all events used; otherwise only
 * events with p == 1 belong to the frame, which is what PolarityFilterAlgorithm(1) does upstream,
 * depth_reprojection_pipe.py:43,114).  Outputs (either may be NULL):
 *   depth_out f32 [H][W]      -- disparity_to_depth_rectified of the final disparity frame (A5)
 *   bgr_out   u8  [H][W][3]   -- the array process_ev_frame hands to frame_callback (A6+A7)
 * with H x W = projector size (XM_VIEW_PROJECTOR) or camera size (XM_VIEW_CAMERA).
 * mem == XM_MEM_HOST  : synchronous; stats (may be NULL) filled; returns XM_ERR_INDEX if any event
 *                       indexed out of range (frame still written).
 * mem == XM_MEM_DEVICE: asynchronous on the next slot's stream; the buffers must stay valid until
 *                       xm_sync(); stats of the most recent frame via xm_last_frame_stats() after it.
 * n == 0 and t_max == t_min are defined: an empty frame, resp. every event in X-map column 0.
 */
int xm_process_frame(xm_handle* h, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p,
                     size_t n, int t_dtype, int mem, float* depth_out, uint8_t* bgr_out, xm_frame_stats* stats);

/* Same for Metavision's 16-byte EventCD records {u16 x; u16 y; i16 p; (pad) ; i64 t} (AoS), the
 * layout `evs` has when trigger_finder.py:172 calls process_ev_frame.  use_polarity != 0 applies p==1. */
int xm_process_frame_aos(xm_handle* h, const void* eventcd16, size_t n, int use_polarity, int mem,
                         float* depth_out, uint8_t* bgr_out, xm_frame_stats* stats);

int xm_last_frame_stats(xm_handle* h, xm_frame_stats* stats);

/* HIP-event time of an EMPTY event pair on slot 0's stream (milliseconds, median of `reps`): what every
 * gpu_ms[] interval of xm_profile_frame contains on top of the kernel itself.  bench.py subtracts it. */
int xm_profile_event_overhead(xm_handle* h, int reps, float* ms_out);

/* Instrumented run of one frame (device-resident SoA buffers): each of the three kernels is launched with
 * hipExtLaunchKernelGGL, i.e. with a start and a stop HIP event attached to its own dispatch packet on the
 * stream it runs on (the timestamps rocprofv3 --kernel-trace reports); synchronous; stats->gpu_ms filled.
 * Used by bench.py for the roofline line. */
int xm_profile_frame(xm_handle* h, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p,
                     size_t n, int t_dtype, float* depth_out, uint8_t* bgr_out, xm_frame_stats* stats);

/* ---- a group of frames in ONE set of multi-frame launches (grid = frames x tiles) ------------------- */
/* All pointers are device pointers; frame f reads events [offsets_host[f], offsets_host[f+1]) of the SoA columns and
 * writes depth_out + f*H*W (and bgr_out + f*H*W*3); n_frames <= n_slots (every frame of the group owns a slot = a key
 * frame + state).  Asynchronous like XM_MEM_DEVICE calls: successive groups rotate over the handle's streams, so with
 * n_slots >= 2 * n_frames the tail of one group overlaps the head of the next; xm_sync() to wait.  XM_FLAG_TRY_SORTED /
 * XM_FLAG_TIME_SORTED apply per frame exactly as for single frames (a frame whose shortcut failed is redone when its
 * slot comes round again or in xm_sync()).  Frames too sparse for the tiled kernels are run one by one on the group's
 * stream.  Why: a single C-1M frame is 245 K1 blocks for 256 CUs, each a ~10 us dependent chain -- its launches leave
 * the chip half idle while they ramp up and drain; a group's launch keeps every CU fed. */
int xm_process_batch(xm_handle* h, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p, int t_dtype,
                     const uint64_t* offsets_host, int n_frames, float* depth_out, uint8_t* bgr_out);
/* The same for Metavision EventCD records (16-byte AoS, `python/frame_event_filter.py:33-37`) resident in device memory: frame f
 * = records [offsets_host[f], offsets_host[f+1]); every event is used (no polarity selection).  What an offline replay of a
 * recording feeds: `DepthReprojectionPipe.process_ev_frames` in x_maps_amd/depth_reprojection_pipe.py stages a list of host
 * frames and calls this. */
int xm_process_batch_aos(xm_handle* h, const void* eventcd16, const uint64_t* offsets_host, int n_frames, float* depth_out,
                         uint8_t* bgr_out);
/* The same group, synchronously, with start / stop events attached to the dispatch packets of its launches (the time
 * stamps rocprofv3 --kernel-trace reports): gpu_ms[0] = the extrema pass K0 (general path) or the boundary pass K0b
 * (column tiles), 0 when neither ran; [1] = K1; [2] = K2; [3] = start of the first .. end of the last. */
int xm_profile_batch(xm_handle* h, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p, int t_dtype,
                     const uint64_t* offsets_host, int n_frames, float* depth_out, uint8_t* bgr_out, float gpu_ms[4]);

/* ---- a batch of frames captured once into a hipGraph and replayed (BASELINE config 5) ------------ */
/* The capture uses the multi-frame launches above: with n_slots >= n_frames the whole batch is three kernel nodes;
 * otherwise groups of n_slots / 2 frames alternate between two branches of the graph. */
/* All pointers are device pointers that stay valid for the graph's lifetime.  Frame f reads events
 * [offsets[f], offsets[f+1]) of the SoA columns and writes depth_out + f*H*W (and bgr_out + f*H*W*3). */
typedef struct xm_graph xm_graph;
int xm_graph_create(xm_handle* h, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p,
                    int t_dtype, const uint64_t* offsets_host, int n_frames, float* depth_out, uint8_t* bgr_out,
                    xm_graph** out);
int xm_graph_launch(xm_graph* g); /* asynchronous; xm_sync(handle) to wait */
void xm_graph_destroy(xm_graph* g);

/* ---- per-event debug view: every intermediate of A1/A2 for bit-exact tests ------------------------ */
/* Host or device SoA in (mem), host or device out (same mem).  Any output may be NULL.
 *   xr, yr  : rectify_cam_coords_i16            ts : X-map column (t_scaled, xmd:19)
 *   disp    : xp - xr - x_offset (int16 wrap), defined where the y-mask holds, else 0
 *   mask    : final inlier mask (u8 0/1), y-mask & disp >= 0 (& p == 1 when p given)              */
/* tests: the column-tile path's exact integer time thresholds thr[0 .. xmap_width] of a frame whose first / last stamps are
 * t_first / t_last: thr[c] = the smallest a in [0, t_last - t_first + 1] with column(t_first + a) >= c, column() being
 * `rint(((t - tmin) / (tmax - tmin)) * (xmap_width - 1))` in float64 exactly as python/x_maps_disparity.py:16-19 computes it. */
int xm_debug_cols_thresholds(xm_handle* h, long long t_first, long long t_last, uint32_t* out_host);
/* tests: A3's output of the handle's last frame when it took the column / owner tiles (xm_path_counts; not after a redo): the
 * u16 disparity frame as [rect_height][rect_width] row-major (python/cam_proj_calibration.py:299-303's disp_map, 0 = empty). */
int xm_debug_last_disp_frame(xm_handle* h, uint16_t* out_host);
/* tests: frames finished by the software-pipelined K2 (groups on the u16 frame) since xm_create; XM_K2_PIPE=2 in the environment
 * sends every group there, XM_K2_PIPE=0 none. */
int xm_debug_k2_pipe_frames(xm_handle* h, uint64_t* count);
int xm_debug_event_outputs(xm_handle* h, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p,
                           size_t n, int t_dtype, int mem, int16_t* xr, int16_t* yr, int16_t* ts, int16_t* disp,
                           uint8_t* mask);

/* ---- the reference's stages one by one (host pointers, synchronous) -------------------------------- */
/* A1  CamProjMaps.rectify_cam_coords_i16(events) -> (xr, yr) */
int xm_stage_rectify(xm_handle* h, const uint16_t* x, const uint16_t* y, size_t n, int16_t* xr, int16_t* yr);
/* A1f CamProjMaps.rectify_cam_coords_f32(events) (cam_proj_calibration.py:272-275): gather from the caller's float
 *     rectify maps, row-major [cam_h][cam_w]; the offline evaluation caller needs it for the point cloud
 *     (eval/compute_depth_x_maps.py:99).  Out-of-range coordinates -> XM_ERR_INDEX. */
int xm_stage_rectify_f32(xm_handle* h, const float* mapx_f32, const float* mapy_f32, const uint16_t* x,
                         const uint16_t* y, size_t n, float* xr, float* yr);
/* CamProjMaps.construct_point_cloud(xpr, ypr, disp) (cam_proj_calibration.py:319-331): Q is the 4x4 float64
 *     reprojection matrix (row-major; cast to float32 like the reference), cloud is float32 [n][3]. */
int xm_stage_point_cloud(xm_handle* h, const double* Q, const float* xpr, const float* ypr, const float* disp, size_t n,
                         float* cloud);
/* A2  compute_disparity(xr, yr, t, X, T_PX_SCALE, X_OFFSET): full-length disp[n] (0 where masked out)
 *     and mask[n]; the caller compacts disp[mask] to get the reference's first return value */
int xm_stage_event_disparity(xm_handle* h, const int16_t* xr, const int16_t* yr, const void* t, size_t n,
                             int t_dtype, int16_t* disp, uint8_t* mask);
/* A3  compute_disp_map_projector_view: full-length xr, yr, disp + mask -> f32 [rect_h][rect_w] */
int xm_stage_disp_map_projector_view(xm_handle* h, const int16_t* xr, const int16_t* yr, const int16_t* disp,
                                     const uint8_t* mask, size_t n, float* disp_map);
/* A3' compute_disp_map_camera_view: x, y, disp + mask -> f32 [cam_h][cam_w] */
int xm_stage_disp_map_camera_view(xm_handle* h, const uint16_t* x, const uint16_t* y, const int16_t* disp,
                                  const uint8_t* mask, size_t n, float* disp_map);
/* A4  DisparityToDepth.remap_rectified_disp_map_to_proj: f32 [rect_h][rect_w] -> f32 [proj_h][proj_w] */
int xm_stage_remap_rectified_disp_map_to_proj(xm_handle* h, const float* rect_disp, float* proj_disp);
/* A5  disparity_to_depth_rectified(disp, P2): f32 [h][w] -> f32 [h][w] */
int xm_stage_disparity_to_depth(xm_handle* h, const float* disp, int height, int width, float* depth);
/* A5+A6+A7 DisparityToDepth.colorize_depth_from_disp: f32 disparity frame -> BGR u8 [h][w][3] */
int xm_stage_colorize_depth_from_disp(xm_handle* h, const float* disp, int height, int width, uint8_t* bgr);

/* ---- multi-GPU: one shard of a frame ---------------------------------------------------------------- */
/* All pointers are DEVICE pointers.  key_frame is a caller-owned u64 [H][W] buffer (H x W = rect size
 * for the projector view, camera size for the camera view), e.g. a torch.int64 tensor, so that the
 * caller can all-reduce it (MAX) between scatter and finish.  Calls are enqueued on slot 0's stream
 * and are asynchronous unless noted. */
/* extrema of this shard's t (over p == 1 events when p given), written as 2 values of t's dtype to
 * minmax_out_host; synchronous.  An empty shard returns (+max, -max) sentinels of the dtype. */
int xm_shard_minmax(xm_handle* h, const void* t, const int16_t* p, size_t n, int t_dtype, void* minmax_out_host);
/* The same without a host round trip: the shard's extrema are left in a 16-byte DEVICE buffer as {tmin, -tmax} -- int64
 * for XM_T_INT64, float64 for the float dtypes (exact) -- so that ONE MIN all-reduce of that buffer over the ranks gives
 * the frame's extrema; an empty shard writes {+max, +max} resp. {+inf, +inf} (neutral).  Asynchronous on slot 0's stream. */
int xm_shard_minmax_device(xm_handle* h, const void* t, const int16_t* p, size_t n, int t_dtype, void* mm_dev);
/* xm_shard_scatter with the frame's {tmin, -tmax} read from that device buffer when the kernel runs (after the
 * all-reduce, ordered on slot 0's stream): minmax -> all-reduce -> scatter -> all-reduce -> finish without a single
 * host synchronisation. */
int xm_shard_scatter_device(xm_handle* h, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p, size_t n,
                            int t_dtype, uint64_t idx_offset, const void* frame_mm_dev, uint32_t tag, uint64_t* key_frame);
/* zero the key frame (once per buffer, or when the tag wraps) */
int xm_shard_clear(xm_handle* h, uint64_t* key_frame);
/* scatter this shard's events; idx_offset = global index of the shard's first event; frame_minmax_host
 * = the FRAME's (tmin, tmax) in t's dtype; tag in [1, 2^19) must grow from frame to frame */
int xm_shard_scatter(xm_handle* h, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p, size_t n,
                     int t_dtype, uint64_t idx_offset, const void* frame_minmax_host, uint32_t tag,
                     uint64_t* key_frame);
/* key frame (after the reduce) -> depth / BGR device buffers */
int xm_shard_finish(xm_handle* h, const uint64_t* key_frame, uint32_t tag, float* depth_out, uint8_t* bgr_out);
/* Cheaper exchange for many ranks: instead of all-reducing the whole 8-byte key frame, REDUCE-SCATTER it (MAX; rank r gets
 * cells [r*C, (r+1)*C) reduced), decode the own chunk to u16 disparities (xm_shard_decode_u16: 0 where the tag differs),
 * ALL-GATHER the u16 chunks (2 bytes per cell) and run the frame kernel on the plain disparity frame
 * (xm_shard_finish_u16; same cell order as the key frame).  Per rank (W-1)/W * (8 + 2) bytes per cell cross the links
 * instead of (W-1)/W * 16.  All asynchronous on slot 0's stream. */
int xm_shard_decode_u16(xm_handle* h, const uint64_t* key_cells, size_t n_cells, uint32_t tag, uint16_t* disp_out);
int xm_shard_finish_u16(xm_handle* h, const uint16_t* disp_frame, float* depth_out, uint8_t* bgr_out);
/* Band-sharded finish ("reduce-scatter by bands + halo, K2 band-sharded, gather the projector frame"): after the
 * reduce-scatter a rank holds the merged disparities of a band of frame columns; with a halo of xm_k2_patch_cols_max() columns
 * from either neighbour it can finish every projector tile whose patch is centred on one of its columns [col_lo, col_hi).
 * disp_frame is a full-size u16 frame ([rect_w][rect_h], column-major) of which only the band and its halos need to be valid;
 * depth_out / bgr_out are full-size projector frames: pixels of other ranks' tiles are left as they were (zero them first; a MAX
 * all-reduce over the ranks -- depth >= 0, an owner's BGR >= the zeros of the others -- assembles the frame).  Projector view. */
int xm_shard_finish_u16_band(xm_handle* h, const uint16_t* disp_frame, int col_lo, int col_hi, float* depth_out, uint8_t* bgr_out);

/* ---- shards on the column tiles (time-sorted int64 frames on rigs whose X-map is injective: the path xm_process_batch takes) ----
 * A shard is a contiguous index range of the sorted stream = whole time columns, except that its last column may go on in the
 * next shard.  Each rank but the last leaves the events of its last column to its successor; then every column is processed by
 * ONE rank, the ranks' plain u16 disparity frames are disjoint and merge by SUM (2 bytes per cell on the wire instead of the
 * 8-byte packed keys; no atomics, no extrema pass).  Per frame and rank, all enqueued on xm_stream(h, 0), TWO collectives:
 *   xm_shard_cols_pack      send_buf <- {t[0], t[n-1], n, count | x[cap] | y[cap] | t[cap]}: the shard's first / last stamp and its last
 *                           min(n, cap) events                               then ALL-GATHER of the send buffers (send_bytes each)
 *   xm_shard_cols_scatter   the frame's extrema out of the gathered headers; the own part ends where the own last column starts
 *                           (not on the last rank); the predecessor's last column out of its buffer in front of the own events
 *                           -- x / y / t (16-byte aligned) MUST have cap_events + 8 events of writable headroom in front of them
 *                           and 8 behind --; boundary pass + column-tile K1 into frame16
 *                                                                            then SUM all-reduce of reduce_u32 uint32 of frame16
 *   xm_shard_finish_u16     the frame kernel on the merged frame
 * xm_shard_cols_info: sizes for frames of n_frame_events events (frame16 holds frame_bytes: the frame + the boundary pass'
 * scratch behind it; XM_ERR_INVALID when the rig / density does not qualify).  A piece that cannot be handled (a shard without
 * events or inside one column, a last column beyond cap_events, events out of order, >= 2^32 us) raises a sticky flag instead
 * of producing a wrong frame: xm_shard_cols_failed (synchronises) reports and clears it -- MAX-all-reduce it over the ranks and
 * redo the frames since the last check with the packed keys (xm_shard_scatter_device ...). */
int xm_shard_cols_info(xm_handle* h, uint64_t n_frame_events, size_t* frame_bytes, size_t* reduce_u32, size_t* send_bytes,
                       size_t* cap_events);
int xm_shard_cols_pack(xm_handle* h, const uint16_t* x, const uint16_t* y, const int64_t* t, size_t n, void* send_buf_dev,
                       size_t cap_events);
int xm_shard_cols_scatter(xm_handle* h, uint16_t* x, uint16_t* y, int64_t* t, size_t n, uint64_t n_frame_events,
                          const void* gathered_dev, size_t send_bytes, int rank, int world, size_t cap_events, uint16_t* frame16);
int xm_shard_cols_failed(xm_handle* h, int* failed);
/* measurement: milliseconds of the column-tile K1 alone in the last xm_shard_cols_scatter issued while
 * xm_debug_option("XM_SHARD_PROFILE", "1") was set (HIP events tied to that dispatch; synchronises the handle's stream) */
int xm_shard_cols_last_k1_ms(xm_handle* h, float* ms);

/* ---- one frame over several GPUs of ONE process (xm_create_sharded owns the RCCL communicators) ----------------
 * The entry for hosts that are not Python / torch.distributed (x_maps_amd/sharded.py is the multi-process form of the same
 * exchange).  dev_ids[n_dev]: distinct HIP devices; the tables of cfg are uploaded to each (cfg->device is ignored).  Per frame
 * the event buffer (HOST memory, pageable or pinned) is split by index -- device g takes [g n / n_dev, (g + 1) n / n_dev) --,
 * one host thread per device enqueues on its device's stream: H2D, the shard's extrema, ncclAllReduce(MIN) of {tmin, -tmax},
 * the scatter with GLOBAL event indices in the packed keys, ncclAllReduce(MAX, uint64) of the key frame, and device dev_ids[0]
 * runs the frame kernel and copies depth / BGR to the caller's host buffers (either may be NULL).  Synchronous.  Results equal
 * xm_process_frame on one device bit for bit (the largest global index wins a cell = NumPy's last-writer-wins).
 * RCCL is looked up at run time (the copy already loaded into the process, else ROCm's librccl); without it only n_dev = 1 works.
 * stats (may be NULL): n_events, t_min / t_max, gpu_ms[0] / gpu_ms[1] = the two all-reduces on dev_ids[0]'s stream. */
typedef struct xm_sharded xm_sharded;
int xm_create_sharded(const int* dev_ids, int n_dev, const xm_config* cfg, xm_sharded** out);
int xm_sharded_process_frame(xm_sharded* s, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p, size_t n,
                             int t_dtype, float* depth_out, uint8_t* bgr_out, xm_frame_stats* stats);
int xm_sharded_info(xm_sharded* s, int* n_dev, int* uses_rccl, uint64_t* key_frame_bytes);
/* Which exchange the frames so far took: time-sorted int64 frames without a polarity column on rigs whose X-map is injective go
 * through the columns exchange (xm_shard_cols_*: all-gather of the shards' last events + SUM all-reduce of plain u16 frames, 2
 * bytes per cell on the wire); a frame one of whose pieces objected is redone with the packed keys (frames_redone); everything
 * else takes the keys (frames_keys).  Results are the same bit for bit. */
int xm_sharded_stats(xm_sharded* s, uint64_t* frames_columns, uint64_t* frames_keys, uint64_t* frames_redone);
void xm_sharded_destroy(xm_sharded* s);

/* ---- one rank of a frame sharded over several PROCESSES (one per GPU), the library driving RCCL itself (new, round 4) --------
 * The xm_shard_* calls above leave the collectives to the host (x_maps_amd/sharded.py issues them through torch.distributed:
 * five calls and two collectives from Python per frame, ~64 us of host time).  Here the library owns the communicator: ONE
 * call per frame enqueues kernels and collectives on xm_stream(h, 0) (~15 us of host time), and a C / C++ host needs no Python.
 *   xm_shard_comm_id        rank 0: XM_SHARD_COMM_ID_BYTES opaque bytes (an RCCL unique id) for the host to hand to every rank
 *                           by whatever transport it has (MPI_Bcast, a file, torch.distributed.broadcast ...)
 *   xm_shard_comm_create    every rank, collective: the communicator on h's device + the exchange buffers for frames of
 *                           n_frame_events events.  Destroy it BEFORE h.  Several communicators per process (one per handle:
 *                           frames in flight on different streams) are fine -- each from an id of its own.
 *   xm_shard_comm_frame     the columns merge (see xm_shard_cols_*): pack -> ncclAllGather -> prepare + boundary pass +
 *                           column-tile K1 -> ncclAllReduce(SUM) of the u16 frame -> frame kernel into depth_out / bgr_out
 *                           (DEVICE pointers; both NULL: the merged frame stays in the communicator, e.g. on ranks that do not
 *                           need the result).  x / y / t: this rank's shard, 16-byte aligned, with cap_events + 8 events of
 *                           writable headroom in front and 8 behind (xm_shard_comm_info).  Asynchronous.
 *   xm_shard_comm_frame_keys  the packed-key merge (any rig, any event order, optional polarity column p): extrema ->
 *                           ncclAllReduce(MIN) -> clear + scatter with global indices (first_index = the shard's offset in the
 *                           frame) -> ncclAllReduce(MAX, uint64) -> frame kernel.  Also the redo of flagged columns frames.
 *   xm_shard_comm_failed    collective + synchronises: did ANY rank flag a columns frame since the last call?
 * RCCL is looked up at run time like for xm_create_sharded. */
#define XM_SHARD_COMM_ID_BYTES 128
typedef struct xm_shard_comm xm_shard_comm;
int xm_shard_comm_id(void* id_out);
int xm_shard_comm_create(xm_handle* h, const void* id, int rank, int world, uint64_t n_frame_events, xm_shard_comm** out);
int xm_shard_comm_info(xm_shard_comm* c, int* takes_columns, size_t* cap_events, size_t* send_bytes, size_t* frame_bytes);
int xm_shard_comm_frame(xm_shard_comm* c, uint16_t* x, uint16_t* y, int64_t* t, size_t n_own, float* depth_out, uint8_t* bgr_out);
int xm_shard_comm_frame_keys(xm_shard_comm* c, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p, size_t n_own,
                             int t_dtype, uint64_t first_index, float* depth_out, uint8_t* bgr_out);
int xm_shard_comm_failed(xm_shard_comm* c, int* failed);
void xm_shard_comm_destroy(xm_shard_comm* c);
int xm_k2_patch_cols_max(xm_handle* h, int* cols_out);
/* the stream the shard calls run on (hipStream_t as void*), so the caller can order its collective */
void* xm_stream(xm_handle* h, int slot);

/* ---- setup-time table construction ("next" row N1) --------------------------------------------------- */
/* compute_x_map_from_time_map (x_map.py:5-55): rectified projector time map f32 [height][width] (0 = undefined)
 * -> X-map int16 [height][x_map_width] (values x + x_offset, 0 = undefined) and, optionally, the matched time
 * differences f32 [height][x_map_width] (t_diffs may be NULL).  Host pointers, synchronous, no handle needed. */
int xm_build_x_map(int device, const float* time_map, int height, int width, int x_map_width, int t_px_scale,
                   int x_offset, int num_scanlines, int16_t* x_map_out, float* t_diffs_out);

/* ---- evaluation metrics ("next" row N4), python/eval/create_evaluation_table.py:14-63 --------------------------------- */
/* evaluation_stats(estimate, groundtruth) on two f32 depth maps [height][width] (host pointers, synchronous, no handle):
 * fill rate, RMSE, % of pixels off by more than 1 / 5 / 10 (the script's unit is cm), and the margin (1 % of the mean
 * ground-truth depth).  filter != 0 first applies load_and_filter(estimate, gt, min_depth, max_depth) (:57-62). */
typedef struct xm_eval_result {
  double fillrate, rmse, perc_1, perc_5, perc_10, margin;
  uint64_t n_valid;   /* pixels with gt > 0 and estimate > 0 (the RMSE's support) */
  uint64_t n_gt_zero; /* pixels without ground truth */
} xm_eval_result;
int xm_eval_stats(int device, const float* estimate, const float* groundtruth, int height, int width, int filter,
                  float min_depth, float max_depth, xm_eval_result* out);

/* ---- per-frame de-duplication filters ("next" row N3), frame_event_filter.py:19-128 --------------------------- */
#define XM_FILTER_FIRST_PER_YT 1      /* FirstEventPerYTFilter          (:68-97)  cell = (y, xp[i])          */
#define XM_FILTER_FIRST_PER_XY 2      /* FirstEventPerXYFilter          (:43-65)  cell = (y, x)              */
#define XM_FILTER_LAST_PER_XY 3       /* LastEventPerXYFilter           (:19-40)                            */
#define XM_FILTER_MEAN_FIRST_LAST_PER_XY 4 /* MeanFirstLastEventPerXYFilter (:100-128)                       */
/* EventCD records in (host, n), events with p != 1 are ignored; xp_i16[n] only for XM_FILTER_FIRST_PER_YT.  The
 * reference sizes its maps map_height = max(y)+1, map_width = max(x)+1 (or max(xp)+1): pass those.  Output: EventCD
 * records in raster order of the cells (capacity map_height*map_width), *n_out of them.  Synchronous.
 * intended_semantics == 0 reproduces what the reference computes: its "first event" maps are written through
 * reversed views (`map[y[::-1], x[::-1]] = t[::-1]`), which NumPy iterates in memory order, so every filter keeps
 * the LAST event per cell (golden vectors g5_filters).  != 0: the first event, as the class names say. */
int xm_frame_event_filter(xm_handle* h, int filter, int intended_semantics, const void* eventcd16_in, size_t n,
                          const int16_t* xp_i16, int map_height, int map_width, void* eventcd16_out, size_t* n_out);

/* ---- device-side ingest ("next" row N2): inter-event pauses of a stream ------------------------------------------ */
/* np.nonzero(np.diff(t) >= thresh_us)[0] (trigger_finder.py:153-155) for a stream of n events.  Exactly one of t
 * (int64[n]) / eventcd16 (n 16-byte EventCD records) is given; mem = XM_MEM_HOST or XM_MEM_DEVICE for that input.
 * idx_out: host buffer of capacity idx_capacity (uint32 indices, ascending); *n_out = number of pauses found (may
 * exceed idx_capacity: then only the first idx_capacity are written).  Synchronous. */
int xm_find_pauses(xm_handle* h, const int64_t* t, const void* eventcd16, size_t n, int mem, int64_t thresh_us,
                   uint32_t* idx_out, size_t idx_capacity, size_t* n_out);

/* ---- device-side ingest ("next" row N2): raw camera packets -> frames, the stream never leaves HBM ----------------------- */
/* Replaces, per packet, what depth_reprojection_pipe.py:110-119 + trigger_finder.py:128-189 do on the host: polarity filter
 * (p == 1), activity-noise filter, buffering, pause detection (diff(t) >= 40 us), frame cut (> 1/2 period, <= 1 period,
 * > 1000 events, 2 events trimmed on both sides) -- as kernels over a device-resident event RING (capacity_events rounded up to
 * a power of two; its first half is mirrored behind its end, so a frame of up to capacity / 2 events is contiguous wherever it
 * starts and nothing is ever moved).  Per packet: three ingest launches (four with the activity filter) (count, append, trigger finder -- pauses are found once,
 * when an event is appended, and kept in a ring of stream indices), the frame kernels K0 -> K1 -> K2 on the frame the DEVICE
 * described (a record in device memory; the host learns from a 16-byte verdict per packet WHETHER it cut a frame and launches
 * the frame kernels with exact grids only then), the frame's statistics, and -- on a stream of their own, beside the next frame's
 * kernels -- the DMA copies of its outputs to the result ring and the sequence number behind them.  xm_ingest_push only copies
 * the packet H2D (pinned staging ring, its own stream) and enqueues those launches; finished frames appear in a ring of pinned
 * host buffers and are picked up with xm_ingest_poll.  One frame at most is cut per push, exactly like
 * RobustTriggerFinder.process_events.  Threads: a launch thread (per packet: the copy and the launches) and an out thread (per
 * cut frame: the result copies) unless XM_INGEST_NO_LAUNCH_THREAD.  The four HIP streams come from one set per device and
 * process, lent to one ingest at a time and never destroyed (an ingest alive beside another one creates its own).
 * Activity filter (the reference builds and runs one unconditionally, depth_reprojection_pipe.py:65-67,116-117): Metavision's
 * ActivityNoiseFilterAlgorithm comes as a binary with the SDK; the rule implemented here (own definition, same in
 * oracle/ingest_oracle.py, which lists what is known of the differences): an event is kept iff an EARLIER event of the stream at
 * one of its 8 neighbouring pixels has t - t' <= activity_thresh_us; every (positive) event then joins its pixel's history.
 * Evaluated on the device for every kind of packet -- records, pinned records, EVT 3.0 / 2.0 chunks decoded there -- by one more
 * launch per packet; exact for any event order (a packet whose stamps run backwards or span more than 8 thresholds is judged
 * sequentially on the device: slow, same result).  A stricter comparison (t - t' < T) is activity_thresh_us = T - 1. */
typedef struct xm_ingest xm_ingest;
typedef struct xm_ingest_config {
  uint32_t struct_size;            /* = sizeof(xm_ingest_config) */
  int32_t projector_fps;           /* frame period = 1e6 / fps us (RuntimeParams.projector_fps) */
  int32_t use_polarity;            /* != 0: only events with p == 1 (PolarityFilterAlgorithm(1), pipe:43,114) */
  int32_t activity_filter;         /* != 0: the activity rule above */
  int64_t activity_thresh_us;      /* 0 => int(1e6 / fps) (pipe:65-68) */
  int64_t pause_thresh_us;         /* 0 => 40 (trigger_finder.py:98) */
  int32_t min_events_per_frame;    /* 0 => 1000 (trigger_finder.py:8) */
  int32_t result_ring;             /* finished frames kept in pinned host memory before they are overwritten; 0 => 8 */
  uint64_t capacity_events;        /* resident event ring, events (rounded up to a power of two; a frame may hold at most half of
                                      it, a full ring drops the incoming events and counts them in `overflow`); 0 => 2^21 */
  uint64_t max_packet_events;      /* largest packet xm_ingest_push accepts (<= capacity / 2, <= 2^21); 0 => 2^19 */
  uint64_t expected_events_per_frame; /* hint for the first frames' kernel choice (0: one thread per event until a frame
                                         has been delivered; afterwards the stream's own density decides) */
  int32_t want_depth, want_bgr;    /* which outputs the result ring holds */
  uint32_t flags;                  /* XM_INGEST_* */
  uint32_t reserved;               /* 0 */
} xm_ingest_config;
#define XM_INGEST_NO_LAUNCH_THREAD 1u /* By default xm_ingest_push* only stages the packet and posts it to a launch thread owned
                                       * by the ingest, which issues the copy and the ~10 launches (the caller pays ~2 us per pinned
                                       * packet instead of ~35).  With this flag the calling thread issues them itself. */
#define XM_INGEST_ACT_SELF 2u          /* activity filter: an earlier event at the event's OWN pixel qualifies too (a 3 x 3 window that
                                       * includes its centre).  The rule is this build's own definition (Metavision's filter is a
                                       * binary); the variants it may turn out to need are configuration: this flag, and a strict
                                       * comparison (t - t' < T) = activity_thresh_us - 1 on integer stamps. */
typedef struct xm_ingest_frame {
  uint64_t seq;                    /* frame number, from 0 */
  uint64_t n_events;               /* events of the cut frame */
  int64_t t_first, t_last;         /* its first / last time stamp */
  uint64_t n_inliers, n_index_errors;
  uint64_t live_after;             /* events left in the device buffer after the cut */
  uint32_t overflow;               /* events dropped so far because the device buffer was full */
  uint32_t lost;                   /* != 0: the ring was lapped, frames between the previous one and this were overwritten */
  const float* depth;              /* f32 [H][W] in the pinned ring (NULL if !want_depth); valid until result_ring - 1 */
  const uint8_t* bgr;              /* u8 [H][W][3]                   further frames have been produced                 */
  uint64_t push_seq;               /* number (from 1) of the xm_ingest_push* call whose packet cut the frame (API version 3) */
  float push_to_publish_us;        /* live latency measured by the library: that push call entered -> the frame's sequence number
                                      published (0 when the frame left in order on the frame stream: EVT chunks, no out thread) */
  uint32_t owned;                  /* xm_ingest_poll_owned: != 0 = depth / bgr belong to the caller now (API version 4) */
} xm_ingest_frame;
int xm_ingest_create(xm_handle* h, const xm_ingest_config* cfg, xm_ingest** out);
void xm_ingest_destroy(xm_ingest* g);
/* one packet of raw EventCD records (host memory, any polarity, time-ordered as the camera delivers them); asynchronous */
int xm_ingest_push(xm_ingest* g, const void* eventcd16, size_t n);
/* the same from PINNED host memory (xm_host_alloc / hipHostMalloc): no staging copy on the host; the packet must stay
 * untouched until 16 further packets have been pushed or xm_ingest_flush() has returned */
int xm_ingest_push_pinned(xm_ingest* g, const void* eventcd16_pinned, size_t n);
/* next finished frame, if any: returns 1 and fills *out, 0 if none is ready (never blocks), < 0 on error */
int xm_ingest_poll(xm_ingest* g, xm_ingest_frame* out);
/* The same, but the frame's buffers LEAVE the ring with it (out->owned = 1): they are the caller's -- no lifetime rule -- until
 * each has been given back with xm_frame_pool_release(*pool, buffer, kind) (kind 0: depth, 1: bgr), from any thread, also after
 * xm_ingest_destroy.  This is the reference's contract for frame_callback -- a fresh array per frame, which the window's thread
 * keeps as long as it likes (depth_reprojection_pipe.py:164-167, depth_reprojection_processor.py:62-64) -- without copying the
 * frame a second time on the host: the pinned buffer the DMA filled is handed out and the ring slot gets a spare one (released
 * buffers are reused; a pool of pinned buffers grows to what the consumer holds at once, at most 1024, "XM_INGEST_POOL_CAP").
 * out->owned = 0 (no spare buffer could be had): the frame is a view into the ring exactly as xm_ingest_poll returns it. */
typedef struct xm_frame_pool xm_frame_pool;
int xm_ingest_poll_owned(xm_ingest* g, xm_ingest_frame* out, xm_frame_pool** pool);
void xm_frame_pool_release(xm_frame_pool* pool, void* buffer, int kind);
/* buffers the pool has made so far / in consumers' hands / spare (only while a consumer still holds one of them, or the ingest
 * is alive: the pool goes with the later of the two).  Any pointer may be NULL. */
int xm_frame_pool_stats(xm_frame_pool* pool, uint64_t* allocated, uint64_t* outstanding, uint64_t* spare);
/* Upper bound of the frames the host has not polled yet: frames whose kernels have been issued and not been handed out by
 * xm_ingest_poll* + packets pushed whose verdict (did it cut a frame?) is still out.  While it stays below result_ring - 1 no
 * frame can be lost to the ring being lapped.  wait_below > 0: first wait -- for verdicts only, nothing is synchronised -- until
 * the bound is below that number or no packet is in flight any more (then the caller has frames to poll). */
int xm_ingest_backlog(xm_ingest* g, int wait_below, uint64_t* backlog);
/* 1 if the result ring still holds frame `seq` (xm_ingest_frame.seq) intact, 0 if a later frame has been or is being written
 * over it: a caller that copies a frame out of the ring asks this AFTER the copy (the ring is lapped only when the host falls
 * result_ring - 1 frames behind; never in a pipe that polls after every push with result_ring >= 3) */
int xm_ingest_frame_valid(xm_ingest* g, uint64_t seq);
int xm_ingest_flush(xm_ingest* g); /* wait for everything pushed so far */
int xm_ingest_reset(xm_ingest* g); /* RobustTriggerFinder.reset(): discard the buffered events (and the activity filter's
                                     * per-pixel history: the stream starts over) */
/* the device's counters once everything pushed so far has run (synchronises like xm_ingest_flush): frames cut, events appended
 * behind the filters, events dropped because the ring had no room (also the `overflow` of every frame), events still buffered.
 * Any pointer may be NULL. */
int xm_ingest_device_stats(xm_ingest* g, uint64_t* frames_cut, uint64_t* events_appended, uint64_t* events_dropped, uint64_t* events_live);
/* what the calling thread has paid so far: number of xm_ingest_push* calls, seconds spent inside them, how often a push had to
 * wait for a staging entry (the GPU more than 16 packets behind) and the seconds spent waiting (part of host_seconds_in_push).
 * Any pointer may be NULL. */
int xm_ingest_host_stats(xm_ingest* g, uint64_t* pushes, double* host_seconds_in_push, uint64_t* staging_waits, double* seconds_waiting);

/* ---- the activity filter alone ---------------------------------------------------------------------------------------
 * For a host that keeps the trigger finder on the CPU (the default DepthReprojectionPipe of this build): what
 * `self.act_filter.process_events(self.pos_events_buf, act_out_buf)` is in the reference (depth_reprojection_pipe.py:116-117,
 * the filter built at :65-67 as ActivityNoiseFilterAlgorithm(width, height, int(1e6 / fps))).  One packet of 16-byte EventCD
 * records in host memory, one keep flag (0 / 1) per event out; the per-pixel history stays on the device between calls.  EVERY
 * event handed in takes part (the pipe hands the filter positive events only).  Same rule, kernels and exactness as the
 * ingest's filter above.  Synchronous; thresh_us as activity_thresh_us there (but not defaulted: pass int(1e6 / fps)). */
typedef struct xm_activity xm_activity;
int xm_activity_create(xm_handle* h, int64_t thresh_us, size_t max_packet_events /* 0 => 2^19; longer packets go through in pieces */,
                       xm_activity** out);
void xm_activity_destroy(xm_activity* f);
int xm_activity_process(xm_activity* f, const void* eventcd16, size_t n, uint8_t* keep_out, size_t* n_kept /* nullable */);
int xm_activity_reset(xm_activity* f); /* forget the history */
int xm_activity_set_rule(xm_activity* f, int self_counts); /* != 0: as XM_INGEST_ACT_SELF, from the next packet on */
/* packets (pieces) so far whose stamps ran backwards or spanned more than 8 thresholds: judged sequentially on the device */
int xm_activity_stats(xm_activity* f, uint64_t* sequential_packets);
int xm_ingest_activity_stats(xm_ingest* g, uint64_t* sequential_packets); /* the same for an ingest's filter (synchronises) */
/* packets so far whose first pass of the activity filter went out inside the packet before's counting launch (they were queued
 * when that one was launched: a replay, a camera ahead of the GPU) instead of as a launch of their own (synchronises) */
int xm_ingest_fused_first_passes(xm_ingest* g, uint64_t* n);

/* ---- EVT 3.0 words -> EventCD records on the device --------------------------------------------------------------
 * The reader in front of the ingest for recordings (Prophesee RAW files, EVT 3.0: a public format; the reference reads them
 * through Metavision's closed RawReaderBase, python/bias_events_iterator.py:53-96).  The 16-bit words cross PCIe as they are
 * stored (about 2-4 bytes per event) and are decoded by three kernels (scans over the format's state machine:
 * x_maps_amd/csrc/xmaps_evt3.hpp); the decoder keeps the state (row, time, vector base, 24-bit wrap count) from chunk to chunk.
 * Same results as x_maps_amd/evt3.py's host decoder, word for word (unpinned against Metavision, like that one).
 * max_words = the largest chunk (0 = 2^20), max_events = room for xm_evt3_decode's records (0 = 2 * max_words). */
typedef struct xm_evt3 xm_evt3;
int xm_evt3_create(xm_handle* h, size_t max_words, size_t max_events, xm_evt3** out);
void xm_evt3_destroy(xm_evt3* d);
int xm_evt3_reset(xm_evt3* d); /* forget the state: the next chunk starts a stream */
/* Start-of-stream rule (either encoding; takes effect from the next chunk).  off (default): events in front of the stream's first
 * EVT_TIME_HIGH word are emitted with the time base at its initial 0 (only their low time bits are known); on: they are NOT
 * emitted -- a reader that waits for the first time base.  Which of the two Metavision's reader does is unpinned here
 * (tools/pin_thirdparty.py, case "no_first_time_high", decides). */
int xm_evt3_wait_for_time_base(xm_evt3* d, int on);
/* Synchronous.  *events_dev = the records in device memory (16-byte EventCD, valid until the next call), *n_events their number;
 * XM_ERR_TOO_MANY if the chunk has more words than max_words or decodes to more events than max_events. */
int xm_evt3_decode(xm_evt3* d, const uint16_t* words_host, size_t n_words, const void** events_dev, size_t* n_events);
/* One chunk of words as ONE packet of the ingest: decoded straight into the packet's slot on the decoder's own stream, then
 * everything xm_ingest_push does behind the copy (the activity filter included, when the ingest has it on).
 * words_pinned != 0: the words lie in pinned host memory (xm_host_alloc) and are copied from there (untouched until 16 further chunks have been
 * pushed or xm_ingest_flush() has returned).
 * n_events != NULL: the decoding is waited for and *n_events = the packet's events; a chunk that decodes to more than
 * max_packet_events returns XM_ERR_TOO_MANY with decoder and ingest unchanged (push it again in halves).
 * n_events == NULL: nothing is waited for -- the ingest's kernels read the chunk's event count from device memory; a chunk that
 * decodes to more than max_packet_events is truncated to that and the excess counted in the frames' `overflow`. */
int xm_ingest_push_evt3(xm_ingest* g, xm_evt3* d, const uint16_t* words_host, size_t n_words, int words_pinned, size_t* n_events);
/* The same for EVT 2.0 ("EVT2/EVT3 RAW reader"), the older of Prophesee's two public RAW encodings: 32-bit
 * little-endian words, [31:28] type -- 0x0 CD_OFF / 0x1 CD_ON {t[5:0], x[10:0], y[10:0]}: one event, p = type; 0x8 EVT_TIME_HIGH
 * {t[33:6]}: the time base of the words behind it; triggers and vendor words are skipped (x_maps_amd/csrc/xmaps_evt2.hpp; host
 * form x_maps_amd/evt2.py, independent checker oracle/evt2_oracle.py).  xm_evt2_create makes a decoder OBJECT OF THE SAME TYPE
 * for that encoding: xm_evt3_destroy / xm_evt3_reset serve it, xm_evt2_decode / xm_ingest_push_evt2 take its 32-bit words (the
 * EVT 3.0 entry points refuse it and vice versa).  max_events = 0: max_words (a word is at most one event). */
typedef xm_evt3 xm_raw_decoder; /* the decoder object, whichever encoding it was created for */
int xm_evt2_create(xm_handle* h, size_t max_words, size_t max_events, xm_raw_decoder** out);
int xm_evt2_decode(xm_raw_decoder* d, const uint32_t* words_host, size_t n_words, const void** events_dev, size_t* n_events);
int xm_ingest_push_evt2(xm_ingest* g, xm_raw_decoder* d, const uint32_t* words_host, size_t n_words, int words_pinned, size_t* n_events);

/* ---- pinned host memory for XM_MEM_HOST_PINNED ------------------------------------------------------------- */
int xm_host_alloc(xm_handle* h, size_t bytes, void** out);
int xm_host_free(xm_handle* h, void* p);

/* ---- small device-memory helpers so that a host without torch can stage buffers ------------------- */
int xm_dev_alloc(xm_handle* h, size_t bytes, void** out);
int xm_dev_free(xm_handle* h, void* p);
int xm_dev_upload(xm_handle* h, void* dst_dev, const void* src_host, size_t bytes);
int xm_dev_download(xm_handle* h, void* dst_host, const void* src_dev, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* XMAPS_H */
