// xm_api_ingest.hpp -- C-ABI: device-side ingest (N2): raw camera packets in, frames cut and processed on the device
// (part of libxmaps_hip.so's host side: included by ../xmaps_hip.hip, one translation unit; see that file for the order)
#pragma once

// ---- N2: device-side ingest ----------------------------------------------------------------------------------------
// Who does what (round 4):
//   caller thread   xm_ingest_push*: stages the packet (pageable memory: one memcpy into a pinned ring entry) and posts a job
//   launch thread   per packet: H2D on the copy stream, then k_ing_count / k_ing_append / k_ing_segment on the INGEST stream.
//                   k_ing_segment leaves a 16-byte verdict in pinned memory (did the packet cut a frame, of how many events);
//                   the thread reads the verdicts in packet order and, for a packet that cut a frame, launches K0 -> K1 -> K2 ->
//                   statistics on the FRAME stream with exact grids.
//   out thread      per cut frame: the DMA copies of its outputs (one of three device frames -> the pinned result ring, in 4 MB
//                   pieces) and its sequence number on the OUT stream, behind the frame's K2 -- beside the next frame's kernels.
//                   (A thread of its own because enqueuing a copy behind a running one can block the caller.)
//   xm_ingest_poll  reads the result ring's sequence numbers (pinned memory, no API call)
// The ingest stream may be `ahead` packets in front of the verdict the thread has handled last (0 on small rings: each packet's
// verdict is awaited before the next is issued); k_ing_segment's room rule keeps that many packets' worth of the ring free, and
// the ingest stream waits (on the device) for K1 of a frame before anything issued after it appends.  XM_INGEST_NO_LAUNCH_THREAD:
// the caller does the launch thread's work inside xm_ingest_push* (and waits for each packet's verdict).

// Result buffers that LEAVE the ingest with a frame (xm_ingest_poll_owned) and come back when the consumer lets go of them
// (xm_frame_pool_release): the reference hands frame_callback a fresh array per frame (depth_reprojection_pipe.py:164-167,
// SURVEY 8(b) "Ownership"); copying a 6.2 MB frame out of the result ring for that costs more host time than the GPU needs for
// the frame, so the pinned buffer the DMA filled IS the fresh array and the ring slot gets another one.  The pool outlives its
// ingest while buffers are out (the last release deletes it).
struct xm_frame_pool {
  std::mutex mu;
  int device = 0;
  size_t bytes[2] = {0, 0};            // [0] depth (f32), [1] BGR
  std::vector<void*> free_bufs[2];
  size_t outstanding = 0;              // buffers in consumers' hands
  size_t allocated = 0, cap = 0;       // buffers made so far / at most (then the caller copies, as before)
  bool closed = false;                 // the ingest is gone: a released buffer is freed
};

struct xm_ingest {
  xm_handle* h = nullptr;
  xm_ingest_config cfg{};
  hipStream_t stream = nullptr;        // ingest kernels
  hipStream_t frame_stream = nullptr;  // K0 / K1 / K2 / publish of the frames that were cut
  hipStream_t copy_stream = nullptr;   // H2D of packet k+1 runs beside the kernels of packet k
  hipStream_t out_stream = nullptr;    // DMA of a finished frame to the pinned result ring + its sequence number, beside the next frame's
                                       // kernels (two out streams taking turns were slower: two 6 MB copies at once share the link)
  u64 capacity = 0, max_packet = 0;    // capacity: a power of two (the request rounded up)
  double period = 0.0;
  long long act_thresh = 0;
  int ahead = 0;                       // packets the ingest stream may run ahead of the handled verdicts
  // device
  IngestDev dev{};                     // what every ingest kernel gets by value (ring, pause ring, state, result ring, ...)
  // Activity filter: TWO sets of per-(bucket, pixel) cells + control words, taken in turn by the packets (set = staging entry & 1):
  // the first pass of packet p (k_act_first: fills the packet's cells) then depends on nothing of packet p - 1 -- only on packet
  // p - 2 having emptied the set (k_ing_append) and reset its flags (k_ing_segment).  When packet p is already on its way to the
  // device while packet p - 1 is being launched (a replay, a camera ahead of the GPU), its first pass goes out INSIDE packet p - 1's
  // k_ing_count launch (k_ing_count_act, ingest_launch3): the stream's chain per packet is count -> append -> segment with the filter
  // on as with it off (round 6: 930-990 -> 1070-1095 Mev/s on the ESL-like stream; the first pass on the copy stream, behind the
  // packet's DMA, held up the next packet's copy and ran 705-1000, on the frame stream 600: profiles/r06_ingest.md).  A packet
  // that arrives alone (a live camera) gets its first pass as a launch of its own in front of its k_ing_count, as in round 5.
  ActDev act_base{};                   // set 0 (dev.act is pointed at the packet's set before its kernels are launched)
  static constexpr int VRING = 64;     // per-packet rings: frame descriptor, frame info (device), verdict (pinned host)
  FrameDesc* d_descs = nullptr;
  IngFrameInfo* d_infos = nullptr;
  IngVerdict* h_verdicts = nullptr;    // pinned host ...
  IngVerdict* d_verdicts = nullptr;    // ... and the address the device writes it at
  double push_t[VRING] = {};           // when the xm_ingest_push* call of packet p entered (steady clock), p % VRING
  uint64_t entry_frame[VRING] = {};    // frame number + 1 that the packet which used the ring entry last cut (0: none): the entry is
                                       // read by that frame's K2 / publishing launches, so it is reused only once the frame is out
  static constexpr int NOUT = 3;       // device-side output frames (K2 writes them, a DMA copy takes them to the pinned result ring)
  hipEvent_t k2_ev[NOUT] = {};         // frame stream: K2 has written output frame o (the out stream's DMA waits for it)
  hipEvent_t out_ev[NOUT] = {};        // out stream: output frame o has left for the result ring (the next K2 into it waits for that)
  // The out stream's work is enqueued by a thread of its own (with a launch thread; inline without): hipMemcpyAsync of a second
  // copy onto a stream whose previous copy is still running BLOCKS its caller in the HIP 7.0 runtime PyTorch bundles (seen: 160 us
  // per frame, 7 ms per 43 frames, whenever the copies ran slower than the frames came) -- it must not be the launch thread.
  struct OutJob {
    uint64_t frame_no = 0, push_no = 0;
    int slot = 0, o = 0;
    const FrameDesc* desc = nullptr;
    bool serial = false;               // on the frame stream, in order with the frames' kernels (see out_serial_now)
  };
  std::thread out_th;
  bool out_threaded = false;
  std::atomic<bool> out_stop{false}, out_sleeping{false};
  std::mutex out_mu;
  std::condition_variable out_cv;
  OutJob out_ring[8];                  // (the launch side never runs more than NOUT frames ahead of out_done)
  std::atomic<uint64_t> out_posted{0}; // frames handed to the out side
  std::atomic<uint64_t> out_done{0};   // frames whose copies + sequence number have been ENQUEUED on the out stream (out_ev[o] recorded)
  std::atomic<int> out_error{0};
  std::string out_error_text;
  bool streams_borrowed = false;       // the four streams are the process's set for the device (ingest_stream_set), else own_streams
  hipStream_t own_streams[4] = {nullptr, nullptr, nullptr, nullptr};
  bool out_on_frame_stream = false;    // "XM_INGEST_OUT_SERIAL" = 1: copies + sequence number ALWAYS on the frame stream, in order with the frames' kernels (A/B)
  // Launch side: the frames cut from now on leave on the frame stream.  Set while the packets are EVT 3.0 chunks decoded on the
  // device (typically one frame per chunk: there the in-order form measured 1000 Mev/s against 840-920 on the out stream,
  // tools/esl_evt3_probe.py), cleared for packets of records (1055-1105 on the out stream against 950).
  bool out_serial_now = false;
  size_t out_piece = 4u << 20;         // bytes per D2H copy of a result frame ("XM_INGEST_OUT_PIECE")
  // debug options are read ONCE, in xm_ingest_create (the ingest's threads must not look at the option table while another thread changes it)
  // The out thread publishes a frame's sequence number ITSELF -- a store into the pinned status ring once the frame's copies have
  // completed (it watches their event anyway) -- instead of a one-thread kernel behind them: the out stream then carries DMA copies
  // only and never occupies a compute queue.  That matters: which hardware queue a stream lands on follows the order in which the
  // process created its streams, and a queue whose head is a barrier packet waiting for a 126 us copy holds up the other queues of
  // its pipe (profiles/r05_ingest.md section 2).  "XM_INGEST_HOST_SEQ" = 0: the kernel form (A/B).
  bool host_seq = true;
  bool opt_out_no_query = false;       // "XM_INGEST_OUT_NO_QUERY"
  bool opt_evt3_out_stream = false;    // "XM_INGEST_EVT3_OUT_STREAM"
  bool opt_trace = false;              // "XM_INGEST_TRACE"
  bool opt_act_fuse = true;            // "XM_INGEST_ACT_FUSE" = 0: never ride k_act_first of the NEXT packet on this packet's k_ing_count launch (A/B)
  int act_toggle = 0;                  // launch side: the set of cells the next non-empty packet takes (ingest_act_set)
  uint64_t act_fused_push = 0;         // launch side: the packet whose k_act_first went out with its predecessor's k_ing_count (k_ing_count_act)
  uint64_t act_fused_count = 0;        // ... how many did (statistics)
  const void* next_job = nullptr;      // launch side: the job queued behind the one being run, if it is a packet that has arrived (else NULL)
  double t_out_wait_s = 0.0;           // XM_INGEST_TRACE: launch side waiting for the out side to have enqueued frame f - NOUT
  double t_out_s = 0.0;                // XM_INGEST_TRACE: host seconds the out side spent enqueuing
  float* d_out_depth[NOUT] = {};
  uint8_t* d_out_bgr[NOUT] = {};
  float** d_depth_ring = nullptr;      // the NOUT pointers above, in device memory (k_ing_segment picks one per frame)
  uint8_t** d_bgr_ring = nullptr;
  // staging (pinned host -> device), a small ring so that the copy of packet k+1 does not wait for packet k's kernels
  static constexpr int STAGE = 16;
  uint4* h_pkt[STAGE] = {};
  uint4* d_pkt[STAGE] = {};
  u32* d_pkt_n = nullptr;              // [STAGE] event counts of chunks decoded on the device (written by the decoder's prefix kernel,
                                       // read by the ingest kernels of the packet: one cell per staging entry, free when the entry is)
  hipEvent_t copied_ev[STAGE] = {};    // per staging entry: its H2D has finished (the ingest stream waits for it)
  uint64_t pkt_push[STAGE] = {};       // number of the push that used the entry last (0: never): free once that push's verdict is in
  int pkt_next = 0;
  hipEvent_t k1_ev[8] = {};            // frame stream: K1 of a frame has run (the ingest stream waits for it before appending more)
  // results (pinned host, written by the kernels)
  int ring = 0;
  IngestStatus* h_status = nullptr;
  std::vector<float*> h_depth;
  std::vector<uint8_t*> h_bgr;
  uint64_t next_seq = 0;               // frames delivered through xm_ingest_poll so far
  // owned result buffers (xm_ingest_poll_owned): which buffer a ring slot holds changes under res_mu -- the out side takes the
  // slot's pointers for frame f and notes f there in one step, the poller swaps a slot's buffers only while the slot still says
  // "frame next_seq" (so a slot the ring has lapped is never handed out while a DMA writes it)
  std::mutex res_mu;
  std::vector<uint64_t> slot_frame;    // frame number + 1 whose copies were enqueued into the slot's buffers last (0: none)
  xm_frame_pool* pool = nullptr;       // made by the first xm_ingest_poll_owned
  std::atomic<uint64_t> frames_issued_pub{0};  // = frames_issued, readable by the caller (xm_ingest_backlog)
  // launch side (the launch thread, or the caller without one)
  uint64_t issued = 0;                 // packets whose ingest kernels have been launched
  uint64_t next_verdict = 1;           // the first packet whose verdict has not been handled
  uint64_t frames_issued = 0;          // frames whose kernels have been launched
  std::atomic<uint64_t> handled{0};    // = next_verdict - 1, readable by the caller (staging flow control)
  // The slot's frame tag advances by one per cut frame: the slot is cleared (k_reset_slot: tags back to 0, key frame emptied)
  // before the tag can reach KEY_MAX_TAG -- the tag field of the packed keys is 19 bits wide
  uint64_t frames_since_clear = 0, clear_every = KEY_MAX_TAG - 16;
  // host time spent inside xm_ingest_push* (what the calling thread pays per packet), for xm_ingest_host_stats
  double push_host_s = 0.0, push_wait_s = 0.0;
  uint64_t push_calls = 0, stage_waits = 0;
  // the launch thread's queue
  struct Job {
    int kind = 0;                      // 0: records (pinned host), 1: EVT 3.0 words, 2: stop, 3: records already in d_pkt[k], 4: flush
    int k = 0;                         // staging entry
    size_t n = 0;                      // events (records) / words
    const void* host = nullptr;        // pinned source (the staging entry or the caller's pinned memory); words
    xm_evt3* dec = nullptr;
    bool pinned = true;
    bool arrived = false;              // the copy side has issued the packet's H2D copy / the chunk's decoding and recorded copied_ev[k]
    uint64_t push_no = 0;              // number of the push (from 1; the caller's count = the launch side's `issued` + 1 when its turn comes)
  };
  static constexpr unsigned QCAP = 64;
  Job queue[QCAP];
  std::atomic<unsigned long long> q_head{0}, q_tail{0}, q_done{0};
  std::atomic<int> q_error{0};
  std::string q_error_text;
  std::mutex q_mu;
  std::condition_variable q_cv;
  std::atomic<bool> q_sleeping{false};
  std::thread th;
  bool threaded = false;
  // Copy side (round 5): a second thread IN FRONT of the launch thread issues what brings a packet to the device -- the H2D copy
  // of records, or the H2D + the three decode launches of a RAW chunk, and the event behind them -- and forwards every job, in
  // order, to the launch thread, which then issues one stream-wait and the ingest kernels.  With the activity filter the launch
  // thread's 7 runtime calls per packet were what bounded the stream (profiles/r05_ingest.md); now 2-4 of them run beside the rest.
  // "XM_INGEST_NO_COPY_THREAD": the launch thread does both (A/B).
  Job cqueue[QCAP];
  std::atomic<unsigned long long> c_head{0}, c_tail{0};
  std::atomic<bool> c_sleeping{false};
  std::mutex c_mu;
  std::condition_variable c_cv;
  std::thread copy_th;
  bool copy_threaded = false;
  uint64_t posted = 0;                 // pushes accepted so far (the caller's count)
  double t_block_s = 0.0, t_frames_s = 0.0, t_jobs_s = 0.0;  // launch side: waiting for verdicts / issuing frames / inside jobs (XM_INGEST_TRACE)
};

namespace {

// The ingest's four streams (ingest / frame / copy / out) come from ONE set per device and process: created by the first ingest,
// lent to one ingest at a time (another one that is alive at the same time makes its own), never destroyed.  Which hardware
// resources a set of streams lands on decides how well the ingest's stages overlap, and it depends on what the process created
// before: the FIRST set runs a stream of records packets at 1055-1105 Mev/s, a set created after another one was destroyed at
// 680-750 (the other way round for one-frame-per-packet EVT 3.0 chunks: 840-890 against 1100-1200) -- measured, not understood
// (profiles/r04_ingest.md section 5).  Keeping the first set makes every ingest of the process behave like its first one.
struct IngestStreamSet {
  hipStream_t s[4] = {nullptr, nullptr, nullptr, nullptr};
  bool lent = false;
};
std::mutex g_ing_sets_mu;
std::map<int, IngestStreamSet> g_ing_sets;

// take = true: borrow the device's set (false when somebody has it); take = false: hand out the borrowed set's array
bool ingest_stream_set(int device, bool take, hipStream_t** out) {
  std::lock_guard<std::mutex> lk(g_ing_sets_mu);
  IngestStreamSet& e = g_ing_sets[device];
  if (take) {
    if (e.lent) return false;
    e.lent = true;
    return true;
  }
  if (out) *out = e.s;
  return true;
}

void ingest_stream_release(int device) {
  std::lock_guard<std::mutex> lk(g_ing_sets_mu);
  g_ing_sets[device].lent = false;
}

// the activity filter's device state (xmaps_ingest.hpp: ActDev), for an ingest or for the filter alone (xm_activity_*)
int act_alloc(ActDev* a, int cam_w, int cam_h, long long thresh, size_t max_packet, int n_sets = 1) {
  if (thresh < 0 || thresh >= (1ll << 31) - 2) return fail(XM_ERR_INVALID, "activity threshold must be in [0, 2^31 - 2) us");
  const size_t cam_px = (size_t)cam_w * cam_h;
  *a = ActDev{};
  a->thresh = thresh;
  a->cam_w = cam_w;
  a->cam_h = cam_h;
  HIP_TRY(hipMalloc((void**)&a->last_ts, cam_px * 8));
  HIP_TRY(hipMalloc((void**)&a->cells, cam_px * sizeof(uint2) * ACT_NB * (size_t)n_sets));
  HIP_TRY(hipMalloc((void**)&a->keep, max_packet ? max_packet : 1));
  HIP_TRY(hipMalloc((void**)&a->ctl, 4 * sizeof(u32) * (size_t)n_sets));
  std::vector<long long> init(cam_px, ING_NO_TS);
  HIP_TRY(hipMemcpy(a->last_ts, init.data(), cam_px * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(a->cells, 0, cam_px * sizeof(uint2) * ACT_NB * (size_t)n_sets));
  HIP_TRY(hipMemset(a->ctl, 0, 4 * sizeof(u32) * (size_t)n_sets));
  HIP_TRY(hipDeviceSynchronize());  // (default-stream work: non-blocking streams do not wait for it)
  return XM_OK;
}

void act_free(ActDev* a) {
  if (a->last_ts) (void)hipFree(a->last_ts);
  if (a->cells) (void)hipFree(a->cells);
  if (a->keep) (void)hipFree(a->keep);
  if (a->ctl) (void)hipFree(a->ctl);
  *a = ActDev{};
}

inline double ingest_now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// one frame's work on the out stream: wait for its K2, copy its outputs to the pinned result ring, the sequence number behind them
int ingest_out_frame(xm_ingest* g, const xm_ingest::OutJob& j) {
  xm_handle* h = g->h;
  hipStream_t os = j.serial ? g->frame_stream : g->out_stream;
  // A copy enqueued behind one that is still running can block its caller for as long as that one runs -- inside the runtime,
  // with other threads' calls waiting behind it: the previous frame's copies are seen off first (a query loop, no blocking call).
  const bool host_seq = g->host_seq && g->out_threaded && !j.serial;
  if (g->out_threaded && !j.serial && j.frame_no > 0 && !g->opt_out_no_query && !host_seq) {
    const int po = (int)((j.frame_no - 1) % xm_ingest::NOUT);
    for (int i = 0; hipEventQuery(g->out_ev[po]) == hipErrorNotReady; ++i)
      for (int k = 0; k < 64; ++k) __builtin_ia32_pause();
    (void)hipGetLastError();
  }
  const double t0 = ingest_now();
  HIP_TRY(hipStreamWaitEvent(os, g->k2_ev[j.o], 0));
  const size_t px = (size_t)h->out_w * h->out_h;
  // In pieces of 4 MB: with one frame per packet (EVT 3.0 period chunks) whole 6 MB copies gave 620-870 Mev/s in an ingest's first
  // minutes and 1170-1210 later, pieces 1130 every time (tools/esl_evt3_probe.py); 1 MB pieces overflow a queue of the runtime
  // and stall for milliseconds (tools/ubench/dma_mix.cpp), so the option does not go below 2 MB.
  const size_t piece = g->out_piece;
  const auto copy_out = [&](void* dst, const void* src, size_t bytes) -> int {
    for (size_t off = 0; off < bytes; off += piece)
      HIP_TRY(hipMemcpyAsync((char*)dst + off, (const char*)src + off, std::min(piece, bytes - off), hipMemcpyDeviceToHost, os));
    return XM_OK;
  };
  int rc;
  void *dst_bgr, *dst_depth;
  {
    std::lock_guard<std::mutex> lk(g->res_mu);
    dst_bgr = g->h_bgr[j.slot];
    dst_depth = g->h_depth[j.slot];
    g->slot_frame[j.slot] = j.frame_no + 1;
  }
  if (dst_bgr && (rc = copy_out(dst_bgr, g->d_out_bgr[j.o], px * 3))) return rc;
  if (dst_depth && (rc = copy_out(dst_depth, g->d_out_depth[j.o], px * 4))) return rc;
  if (!host_seq) {
    hipLaunchKernelGGL(k_ing_publish_seq, dim3(1), dim3(64), 0, os, g->dev.st, j.desc, g->h_status + j.slot, (u64)j.frame_no);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(g->out_ev[j.o], os));
  // (nothing else is issued on this stream until the next frame: without a query the runtime kept the last copy and the sequence
  //  number in its batch until some other call of the process flushed it -- seen in the copy trace: the second piece of a frame
  //  starting 90 us after the first, together with the next packet's H2D copy)
  (void)hipStreamQuery(os);
  g->t_out_s += ingest_now() - t0;
  if (host_seq) {
    // the copies have landed once their event has fired (a query loop: no blocking call of the runtime while the launch thread is
    // issuing): then the sequence number, the last thing the poller looks at (xm_ingest_poll reads it with acquire)
    for (hipError_t q; (q = hipEventQuery(g->out_ev[j.o])) != hipSuccess;) {
      if (q != hipErrorNotReady) HIP_TRY(q);
      for (int k = 0; k < 32; ++k) __builtin_ia32_pause();
    }
    (void)hipGetLastError();
    // (live latency as the library sees it: the push call of the packet that completed the frame entered -> now)
    const double t_push = g->push_t[j.push_no % xm_ingest::VRING];
    g->h_status[j.slot].latency_us = t_push > 0.0 ? (float)((ingest_now() - t_push) * 1e6) : 0.0f;
    __atomic_store_n(&g->h_status[j.slot].seq, (uint64_t)j.frame_no + 1, __ATOMIC_RELEASE);
  }
  return XM_OK;
}

void ingest_out_main(xm_ingest* g) {
  (void)hipSetDevice(g->h->cfg.device);
  uint64_t n = 0;  // the next frame to take
  for (;;) {
    // A frame's copy should start the moment its K2 is on the stream (with one frame per packet the copies are what bounds the
    // pipe: a sleeping thread's wake-up would go straight into the frame period): spin for about a millisecond before sleeping.
    for (int i = 0; g->out_posted.load(std::memory_order_acquire) == n; ++i) {
      if (g->out_stop.load(std::memory_order_acquire)) return;
      if (i < 40000) {
        __builtin_ia32_pause();
        continue;
      }
      std::unique_lock<std::mutex> lk(g->out_mu);
      g->out_sleeping.store(true, std::memory_order_seq_cst);
      g->out_cv.wait(lk, [&] { return g->out_stop.load(std::memory_order_acquire) || g->out_posted.load(std::memory_order_acquire) != n; });
      g->out_sleeping.store(false, std::memory_order_relaxed);
      i = 0;
    }
    {  // frames below out_done are done (the launch side took them itself while this thread slept): their entries may be gone
      const uint64_t d = g->out_done.load(std::memory_order_acquire);
      if (d > n) {
        n = d;
        continue;
      }
    }
    const xm_ingest::OutJob j = g->out_ring[n % 8];
    // (a frame the launch side took itself -- out_serial_now -- is only counted: its entry says so, or holds another frame by now)
    if (j.frame_no == n && !j.serial && !g->out_error.load(std::memory_order_relaxed)) {
      const int rc = ingest_out_frame(g, j);
      if (rc) {
        g->out_error_text = g_err;  // (thread-local text of this thread)
        g->out_error.store(rc, std::memory_order_release);
      }
    }
    n += 1;
    for (uint64_t cur = g->out_done.load(std::memory_order_acquire); cur < n;)
      if (g->out_done.compare_exchange_weak(cur, n, std::memory_order_release)) break;
  }
}

// frames below `upto` have their out work enqueued
int ingest_out_drain_upto(xm_ingest* g, uint64_t upto) {
  while (g->out_done.load(std::memory_order_acquire) < upto) {
    if (g->out_error.load(std::memory_order_acquire)) break;
    __builtin_ia32_pause();
  }
  if (g->out_error.load(std::memory_order_acquire)) return fail(g->out_error.load(), "ingest, out side: %s", g->out_error_text.c_str());
  return XM_OK;
}

// every frame issued so far has its out-stream work enqueued
int ingest_out_drain(xm_ingest* g) {
  while (g->out_done.load(std::memory_order_acquire) < g->frames_issued) {
    if (g->out_error.load(std::memory_order_acquire)) break;
    std::this_thread::yield();
  }
  if (g->out_error.load(std::memory_order_acquire)) return fail(g->out_error.load(), "ingest, out side: %s", g->out_error_text.c_str());
  return XM_OK;
}

// K0 -> K1 -> K2 -> statistics for the frame that packet `push_no` cut (n events), on the frame stream; its copies to the out side
int ingest_issue_frame(xm_ingest* g, uint64_t push_no, u64 n) {
  xm_handle* h = g->h;
  hipStream_t s = g->frame_stream;
  const int vi = (int)(push_no % xm_ingest::VRING);
  const FrameDesc* desc = g->d_descs + vi;
  if (g->frames_since_clear >= g->clear_every) {  // (stream-ordered behind every frame so far)
    hipLaunchKernelGGL(k_reset_slot, dim3(1024), dim3(BLOCK), 0, s, g->dev.slot, g->dev.key_frame, (u64)h->key_cells, (unsigned char*)nullptr);
    g->frames_since_clear = 0;
  }
  g->frames_since_clear += 1;
  const u64 nb = n < 2 ? 2 : n;
  {  // K0 (general path: the cut frame is sorted whenever the camera stream is, but nothing here relies on it)
    unsigned gx = grid_for(nb, BLOCK * 4);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL((k_minmax_batch<long long, true, false, 1>), dim3(gx, 1), dim3(BLOCK), 0, s, desc);
  }
  if (!batch_path(h, n)) {  // sparse frame: one thread per event
    const unsigned gx = grid_for(nb, BLOCK);
    if (h->cfg.view == XM_VIEW_PROJECTOR)
      hipLaunchKernelGGL((k_scatter_direct_batch<long long, true, false, 0>), dim3(gx, 1), dim3(BLOCK), 0, s, desc, h->tb, 0);
    else
      hipLaunchKernelGGL((k_scatter_direct_batch<long long, true, false, 1>), dim3(gx, 1), dim3(BLOCK), 0, s, desc, h->tb, 0);
  } else {
    const double max_ev = h->tb.xmap_w > 0 ? (h->w_ts - 1.5) * (double)n / (double)h->tb.xmap_w : 0.0;
    unsigned threads = TILE_THREADS;
    while (threads > 1024 / TILE_EPT && (double)(threads * TILE_EPT) > max_ev) threads >>= 1;
    const unsigned gx = grid_for(nb, threads * TILE_EPT);
    auto launch = [&](auto view_tag) -> int {
      constexpr int VIEW = decltype(view_tag)::value;
      auto kern = k_scatter_tiled_batch<long long, true, false, VIEW, false>;
      int rc = h->ensure_lds(reinterpret_cast<const void*>(kern), h->k1_lds);
      if (rc) return rc;
      hipLaunchKernelGGL(kern, dim3(gx, 1), dim3(threads), h->k1_lds, s, desc, h->tb, h->w_ts, h->w_x, 0);
      return XM_OK;
    };
    int rc = h->cfg.view == XM_VIEW_PROJECTOR ? launch(std::integral_constant<int, 0>{}) : launch(std::integral_constant<int, 1>{});
    if (rc) return rc;
  }
  // the ingest stream must not append over the frame's events (dead, but still in the ring) before K1 has read them: whatever
  // is issued on it from now on waits for this event; what has been issued already fits the room k_ing_segment keeps (`ahead`)
  hipEvent_t ev = g->k1_ev[g->frames_issued % 8];
  HIP_TRY(hipEventRecord(ev, s));
  HIP_TRY(hipStreamWaitEvent(g->stream, ev, 0));
  // K2 writes device output frame o = frame number % NOUT (k_ing_segment put its address into the descriptor) -- once the DMA of
  // the frame that used it last has left
  const uint64_t f = g->frames_issued;
  const int o = (int)(f % xm_ingest::NOUT);
  if (f >= (uint64_t)xm_ingest::NOUT) {
    // (the event must have been RECORDED by the out side before this stream can be told to wait for it)
    const double tw = ingest_now();
    while (g->out_done.load(std::memory_order_acquire) + xm_ingest::NOUT <= f) {
      if (g->out_error.load(std::memory_order_acquire)) return fail(g->out_error.load(), "ingest, out side: %s", g->out_error_text.c_str());
      __builtin_ia32_pause();
    }
    g->t_out_wait_s += ingest_now() - tw;
    HIP_TRY(hipStreamWaitEvent(s, g->out_ev[o], 0));
  }
  if (h->cfg.view == XM_VIEW_PROJECTOR) {
    if (h->k2_direct) return fail(XM_ERR_INVALID, "ingest needs the tiled frame kernel (XM_K2_DIRECT is set)");
    launch_k2_batch<0>(h, s, desc, 1);
  } else {
    const u64 px = (u64)h->tb.cam_w * h->tb.cam_h;
    hipLaunchKernelGGL(k_frame_direct_batch, dim3(grid_for(px, BLOCK), 1), dim3(BLOCK), 0, s, desc, px, h->tb.dlut);
  }
  // the frame's statistics into its status entry while the slot's counters and the frame's events are still the frame's ...
  hipLaunchKernelGGL(k_ing_publish, dim3(1), dim3(64), 0, s, g->dev.st, desc, (const IngFrameInfo*)(g->d_infos + vi), g->h_status, (u64)push_no);
  HIP_TRY(hipGetLastError());
  // ... device -> pinned result ring by DMA and the entry's sequence number behind it on the OUT stream: a 6 MB frame is 140 us
  // on the link, during which the frame stream already runs the next frame's kernels (on one stream the frames came out one DMA
  // + one kernel chain apart).  (frame numbers count on both sides: the verdicts arrive in packet order)
  HIP_TRY(hipEventRecord(g->k2_ev[o], s));
  xm_ingest::OutJob job;
  job.frame_no = f;
  job.push_no = push_no;
  job.slot = (int)(f % (uint64_t)g->ring);
  job.o = o;
  job.desc = desc;
  job.serial = g->out_on_frame_stream || g->out_serial_now;
  if (g->out_threaded && !job.serial) {
    g->out_ring[f % 8] = job;
    g->out_posted.store(f + 1, std::memory_order_seq_cst);
    if (g->out_sleeping.load(std::memory_order_seq_cst)) {
      std::lock_guard<std::mutex> lk(g->out_mu);
      g->out_cv.notify_one();
    }
  } else {
    int rc = g->out_threaded ? ingest_out_drain_upto(g, f) : XM_OK;  // (frames posted before the mode changed come first)
    if (!rc) rc = ingest_out_frame(g, job);
    if (rc) return rc;
    if (g->out_threaded) {  // (the out thread only counts this one)
      g->out_ring[f % 8] = job;
      g->out_posted.store(f + 1, std::memory_order_seq_cst);
    }
    g->out_done.store(f + 1, std::memory_order_release);
  }
  g->entry_frame[vi] = f + 1;
  g->frames_issued += 1;
  g->frames_issued_pub.store(g->frames_issued, std::memory_order_release);
  return XM_OK;
}

// Verdicts in packet order; for a packet that cut a frame, its kernels.  block_upto: wait for the verdicts of packets <= that
// number (0: take what is there).
int ingest_handle_verdicts(xm_ingest* g, uint64_t block_upto) {
  while (g->next_verdict <= g->issued) {
    const uint64_t v = g->next_verdict;
    const IngVerdict* e = g->h_verdicts + (v % xm_ingest::VRING);
    if (__atomic_load_n(&e->push_no, __ATOMIC_ACQUIRE) != v) {
      if (v > block_upto) return XM_OK;
      const double cb = ingest_now();
      struct Acc { double& a; double t0; ~Acc() { a += ingest_now() - t0; } } acc{g->t_block_s, cb};
      unsigned spins = 0;
      while (__atomic_load_n(&e->push_no, __ATOMIC_ACQUIRE) != v) {
        __builtin_ia32_pause();
        if ((++spins & 0x3ff) == 0) {  // make sure the runtime has handed the launches to the GPU; an idle stream without the
          hipError_t q = hipStreamQuery(g->stream);  // verdict would be a lost launch: report it instead of spinning for ever
          if (q == hipSuccess && __atomic_load_n(&e->push_no, __ATOMIC_ACQUIRE) != v)
            return fail(XM_ERR_HIP, "ingest: packet %llu left no verdict", (unsigned long long)v);
          if (q != hipSuccess && q != hipErrorNotReady) HIP_TRY(q);
        }
      }
    }
    const u64 info = __atomic_load_n(&e->info, __ATOMIC_RELAXED);
    if (info >> 63) {
      const double cf = ingest_now();
      int rc = ingest_issue_frame(g, v, info & ~(1ull << 63));
      g->t_frames_s += ingest_now() - cf;
      if (rc) return rc;
    }
    g->next_verdict = v + 1;
    g->handled.store(v, std::memory_order_release);
  }
  return XM_OK;
}

int ingest_words_to_events(const xm_evt3* d);  // (xm_api_evt3.hpp) upper bound of the events one word of the decoder's format yields

// the activity filter's state as a packet sees it: its set of cells and control words.  The packets that take part -- the
// non-empty ones -- take the two sets strictly in turns (xm_ingest::act_toggle, advanced by ingest_process; an empty push between
// two packets must not make them share a set: the second one's first pass runs beside the first one's counting launch)
ActDev ingest_act_set(const xm_ingest* g, int set) {
  ActDev a = g->act_base;
  if (a.last_ts && (set & 1)) {
    a.cells += (size_t)a.cam_w * (size_t)a.cam_h * ACT_NB;
    a.ctl += 4;
  }
  return a;
}

// the three ingest launches of one (sub-)packet.  With the activity filter on and the NEXT packet already on its way to the device
// (g->next_job: a replay, or a camera that is ahead of the GPU), that packet's first pass (k_act_first) rides on this packet's
// k_ing_count launch (k_ing_count_act: the other set of cells) instead of being a link of its own in the chain of the stream.
void ingest_launch3(xm_ingest* g, const IngestPush& pp, u32 bound) {
  const unsigned nb = (bound + ING_EPB - 1) / ING_EPB;
  if (nb) {
    const xm_ingest::Job* nx = (const xm_ingest::Job*)g->next_job;
    bool fused = false;
    if (nx && g->dev.act.last_ts && g->opt_act_fuse) {
      const bool words = nx->kind == 1;
      const size_t n2 = words ? std::min<size_t>((size_t)g->max_packet, nx->n * (size_t)ingest_words_to_events(nx->dec)) : nx->n;
      const unsigned nb2 = (unsigned)((n2 + ING_THREADS - 1) / ING_THREADS);
      if (n2 && hipStreamWaitEvent(g->stream, g->copied_ev[nx->k], 0) == hipSuccess) {
        hipLaunchKernelGGL(k_ing_count_act, dim3(nb + nb2), dim3(ING_THREADS), 0, g->stream, g->dev, pp, (u32)nb, ingest_act_set(g, g->act_toggle /* the set the next packet will take: ingest_process has advanced it for this one */),
                           (const uint4*)g->d_pkt[nx->k], words ? (const u32*)(g->d_pkt_n + nx->k) : (const u32*)nullptr, (u32)n2,
                           g->cfg.use_polarity ? 1 : 0);
        g->act_fused_push = pp.push_no + 1;
        g->act_fused_count += 1;
        fused = true;
      }
    }
    if (!fused) hipLaunchKernelGGL(k_ing_count, dim3(nb), dim3(ING_THREADS), 0, g->stream, g->dev, pp);
    hipLaunchKernelGGL(k_ing_append, dim3(nb), dim3(ING_THREADS), 0, g->stream, g->dev, pp);
  }
  hipLaunchKernelGGL(k_ing_segment, dim3(1), dim3(ING_THREADS), 0, g->stream, g->dev, pp);
}

// everything behind the packet's arrival in d_pkt[k]: filters, append, segmentation.  hp = the packet in host memory (unused: the
// activity filter is evaluated on the device for every kind of packet); NULL for a packet decoded on the device, whose event
// count then lives at n_dev (device memory) and n is the room of its slot
int ingest_process(xm_ingest* g, int k, size_t n, const uint4* hp, const u32* n_dev = nullptr) {
  xm_handle* h = g->h;
  hipStream_t s = g->stream;
  // the ingest stream stays at most `ahead` packets in front of the verdicts handled here
  int rc = XM_OK;
  while (g->issued + 1 - g->next_verdict > (uint64_t)g->ahead)
    if ((rc = ingest_handle_verdicts(g, g->next_verdict))) return rc;
  const uint64_t push_no = g->issued + 1;
  const int vi = (int)(push_no % xm_ingest::VRING);
  // The entry's previous user (packet push_no - VRING) may have cut a frame, whose K2 and publishing launches read the
  // descriptor and the frame info out of the entry -- the last of them on the out stream, behind the frame's copies.  The ingest
  // stream waits only for that frame's K1, so the entry is handed to k_ing_segment again only once the frame's sequence number
  // is out (a read of pinned memory: by now it is, except with > VRING packets between a cut and a stalled out side).
  if (const uint64_t fe = g->entry_frame[vi]) {
    const IngestStatus* stp = g->h_status + (fe - 1) % (uint64_t)g->ring;
    unsigned spins = 0;
    while (__atomic_load_n(&stp->seq, __ATOMIC_ACQUIRE) < fe) {
      if (g->out_error.load(std::memory_order_acquire)) return fail(g->out_error.load(), "ingest, out side: %s", g->out_error_text.c_str());
      __builtin_ia32_pause();
      if ((++spins & 0xfff) == 0) (void)hipStreamQuery(g->out_stream);
    }
    g->entry_frame[vi] = 0;
  }
  g->dev.desc = g->d_descs + vi;
  g->dev.info = g->d_infos + vi;
  g->dev.verdict = g->d_verdicts + vi;
  (void)hp;
  (void)h;
  IngestPush p{};
  p.flags = (g->cfg.use_polarity ? ING_F_POLARITY : 0u) | ING_F_SEGMENT;
  p.push_no = push_no;
  p.src = g->d_pkt[k];
  p.n = (u32)n;
  p.n_dev = n_dev;
  // activity filter: the packet's first pass (the per-(bucket, pixel) cells, one event per thread; xmaps_ingest.hpp) -- unless it
  // went out with the packet before (ingest_launch3) -- then the flags themselves are computed by k_ing_count as it counts.
  // Nothing is decided here: a chunk decoded on the device is treated like records.
  const int act_set = g->act_toggle;  // (the packet's set of cells: k_ing_count reads them, k_ing_append empties them, k_ing_segment resets its flags)
  if (n) g->act_toggle ^= 1;          // (an empty packet launches k_ing_segment only: it takes no turn)
  g->dev.act = ingest_act_set(g, act_set);
  if (g->dev.act.last_ts && n && g->act_fused_push != push_no)  // (fused: it went out with the packet before, ingest_launch3)
    hipLaunchKernelGGL(k_act_first, dim3((unsigned)((n + ING_THREADS - 1) / ING_THREADS)), dim3(ING_THREADS), 0, s, g->dev.act,
                       (const uint4*)g->d_pkt[k], n_dev, (u32)n, g->cfg.use_polarity ? 1 : 0);
  ingest_launch3(g, p, (u32)n);
  HIP_TRY(hipGetLastError());
  g->issued = push_no;
  // the frame (if this or an earlier packet cut one) as soon as its verdict is in: at once when nothing else is waiting
  return ingest_handle_verdicts(g, 0);
}

// one packet of records, the copy side: H2D on the copy stream (beside the previous packets' kernels) + the event behind it
int ingest_copy_records(xm_ingest* g, int k, size_t n, const uint4* hp) {
  if (n) {
    HIP_TRY(hipMemcpyAsync(g->d_pkt[k], hp, n * 16, hipMemcpyHostToDevice, g->copy_stream));
    HIP_TRY(hipEventRecord(g->copied_ev[k], g->copy_stream));
  }
  return XM_OK;
}

// ... the launch side: everything else (arrived: the copy side has done its part already)
int ingest_issue_records(xm_ingest* g, int k, size_t n, const uint4* hp, bool arrived) {
  g->out_serial_now = false;
  int rc = arrived ? XM_OK : ingest_copy_records(g, k, n, hp);
  if (rc) return rc;
  if (n) HIP_TRY(hipStreamWaitEvent(g->stream, g->copied_ev[k], 0));
  return ingest_process(g, k, n, hp, nullptr);
}

int ingest_copy_evt3(xm_ingest* g, xm_evt3* d, int k, const void* words, size_t n_words, bool pinned);  // (xm_api_evt3.hpp)
int ingest_issue_evt3(xm_ingest* g, xm_evt3* d, int k, const void* words, size_t n_words, bool pinned, bool arrived);          // (xm_api_evt3.hpp)

// every verdict in, every frame's kernels launched and run
int ingest_finish(xm_ingest* g) {
  int rc = ingest_handle_verdicts(g, g->issued);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(g->copy_stream));
  HIP_TRY(hipStreamSynchronize(g->stream));
  HIP_TRY(hipStreamSynchronize(g->frame_stream));
  if ((rc = ingest_out_drain(g))) return rc;
  HIP_TRY(hipStreamSynchronize(g->out_stream));
  return XM_OK;
}

int ingest_run_job(xm_ingest* g, const xm_ingest::Job& j) {
  switch (j.kind) {
    case 0: return ingest_issue_records(g, j.k, j.n, (const uint4*)j.host, j.arrived);
    case 1: return ingest_issue_evt3(g, j.dec, j.k, j.host, j.n, j.pinned, j.arrived);
    case 3: return ingest_process(g, j.k, j.n, nullptr);
    case 4: return ingest_finish(g);
    default: return XM_OK;
  }
}

void ingest_thread_main(xm_ingest* g) {
  (void)hipSetDevice(g->h->cfg.device);
  const auto note = [&](int rc) {
    if (rc != XM_OK && g->q_error.load(std::memory_order_relaxed) == 0) {
      g->q_error_text = g_err;  // thread-local text of this thread
      g->q_error.store(rc, std::memory_order_release);
    }
  };
  for (;;) {
    unsigned long long t = g->q_tail.load(std::memory_order_relaxed);
    if (t == g->q_head.load(std::memory_order_acquire)) {
      // Nothing to launch: verdicts first -- a frame's kernels go out the moment its packet's verdict arrives (a live camera's
      // packets are milliseconds apart: the frame must not wait for the next one) -- then spin a little, then sleep.  Never
      // asleep with a verdict outstanding (it is at most a few ten microseconds away).
      bool got = false;
      for (int i = 0; !got; ++i) {
        if (g->next_verdict <= g->issued) {
          note(ingest_handle_verdicts(g, 0));
          if (g->q_error.load(std::memory_order_relaxed)) g->next_verdict = g->issued + 1;  // (do not spin on a failed stream)
          if ((i & 0x3ff) == 0x3ff) (void)hipStreamQuery(g->stream);  // (a query makes the runtime hand over what it may still hold back)
          if (i >= 1 << 20) i = 0;
        } else if (i >= 20000) {
          break;
        }
        __builtin_ia32_pause();
        got = t != g->q_head.load(std::memory_order_acquire);
      }
      if (!got) {
        std::unique_lock<std::mutex> lk(g->q_mu);
        g->q_sleeping.store(true, std::memory_order_seq_cst);
        g->q_cv.wait(lk, [&] { return t != g->q_head.load(std::memory_order_acquire); });
        g->q_sleeping.store(false, std::memory_order_relaxed);
      }
    }
    const xm_ingest::Job j = g->queue[t % xm_ingest::QCAP];
    g->q_tail.store(t + 1, std::memory_order_release);
    if (j.kind != 2) {
      const double cj = ingest_now();
      // the job queued behind this one, if it is a packet whose copy / decoding the copy side has issued already (its slot of the
      // queue is not reused before q_tail passes it)
      g->next_job = nullptr;
      if ((j.kind == 0 || j.kind == 1 || j.kind == 3) && g->q_head.load(std::memory_order_acquire) > t + 1) {
        const xm_ingest::Job& c = g->queue[(t + 1) % xm_ingest::QCAP];
        if ((c.kind == 0 || c.kind == 1) && c.arrived && c.n) g->next_job = &c;
      }
      note(ingest_run_job(g, j));
      g->next_job = nullptr;
      g->t_jobs_s += ingest_now() - cj;
    }
    g->q_done.store(t + 1, std::memory_order_release);
    if (j.kind == 2) return;
  }
}

// into the launch thread's queue (from the caller, or -- with a copy thread -- from that one: a single producer either way)
unsigned long long ingest_post_launch(xm_ingest* g, const xm_ingest::Job& j) {
  const unsigned long long hd = g->q_head.load(std::memory_order_relaxed);
  while (hd - g->q_tail.load(std::memory_order_acquire) >= xm_ingest::QCAP) __builtin_ia32_pause();  // queue full: back-pressure
  g->queue[hd % xm_ingest::QCAP] = j;
  g->q_head.store(hd + 1, std::memory_order_seq_cst);
  if (g->q_sleeping.load(std::memory_order_seq_cst)) {
    std::lock_guard<std::mutex> lk(g->q_mu);
    g->q_cv.notify_one();
  }
  return hd + 1;
}

// The caller's door: the copy thread's queue when there is one (every job passes through it and is forwarded IN ORDER, so a job's
// number is the same in both queues and q_done counts them alike), else the launch thread's.
unsigned long long ingest_post(xm_ingest* g, const xm_ingest::Job& j) {
  if (!g->copy_threaded) return ingest_post_launch(g, j);
  const unsigned long long hd = g->c_head.load(std::memory_order_relaxed);
  while (hd - g->c_tail.load(std::memory_order_acquire) >= xm_ingest::QCAP) __builtin_ia32_pause();
  g->cqueue[hd % xm_ingest::QCAP] = j;
  g->c_head.store(hd + 1, std::memory_order_seq_cst);
  if (g->c_sleeping.load(std::memory_order_seq_cst)) {
    std::lock_guard<std::mutex> lk(g->c_mu);
    g->c_cv.notify_one();
  }
  return hd + 1;
}

// jobs handed in so far (the caller's count)
unsigned long long ingest_posted(const xm_ingest* g) {
  return g->copy_threaded ? g->c_head.load(std::memory_order_acquire) : g->q_head.load(std::memory_order_acquire);
}

void ingest_copy_thread_main(xm_ingest* g) {
  (void)hipSetDevice(g->h->cfg.device);
  for (;;) {
    const unsigned long long t = g->c_tail.load(std::memory_order_relaxed);
    for (int i = 0; t == g->c_head.load(std::memory_order_acquire); ++i) {
      if (i < 20000) {
        __builtin_ia32_pause();
        continue;
      }
      std::unique_lock<std::mutex> lk(g->c_mu);
      g->c_sleeping.store(true, std::memory_order_seq_cst);
      g->c_cv.wait(lk, [&] { return t != g->c_head.load(std::memory_order_acquire); });
      g->c_sleeping.store(false, std::memory_order_relaxed);
    }
    xm_ingest::Job j = g->cqueue[t % xm_ingest::QCAP];
    g->c_tail.store(t + 1, std::memory_order_release);
    if ((j.kind == 0 || j.kind == 1) && !g->q_error.load(std::memory_order_relaxed)) {
      const int rc = j.kind == 0 ? ingest_copy_records(g, j.k, j.n, (const uint4*)j.host) : ingest_copy_evt3(g, j.dec, j.k, j.host, j.n, j.pinned);
      if (rc != XM_OK) {
        if (g->q_error.load(std::memory_order_relaxed) == 0) {
          g->q_error_text = g_err;  // thread-local text of this thread
          g->q_error.store(rc, std::memory_order_release);
        }
        j.kind = 5;  // (nothing arrived: the launch side only counts the job)
      }
      j.arrived = true;
    }
    ingest_post_launch(g, j);
    if (j.kind == 2) return;
  }
}

int ingest_take_error(xm_ingest* g) {
  const int e = g->q_error.load(std::memory_order_acquire);
  if (!e) return XM_OK;
  g->q_error.store(0, std::memory_order_release);
  return fail(e, "%s (reported by the ingest's launch thread)", g->q_error_text.c_str());
}

// hand a job to the launch thread (or run it here); wait: until it has run
int ingest_submit(xm_ingest* g, const xm_ingest::Job& j, bool wait) {
  if (!g->threaded) return ingest_run_job(g, j);
  const unsigned long long idx = ingest_post(g, j);
  if (wait) {
    while (g->q_done.load(std::memory_order_acquire) < idx) __builtin_ia32_pause();
    return ingest_take_error(g);
  }
  return XM_OK;
}

// The staging entry's previous packet has been consumed once that packet's verdict has been handled (k_ing_segment runs behind
// the kernels that read the packet): no API call, no event.
int ingest_wait_entry(xm_ingest* g, int k) {
  const uint64_t need = g->pkt_push[k];
  if (!need || g->handled.load(std::memory_order_acquire) >= need) return XM_OK;
  if (!g->threaded) return ingest_handle_verdicts(g, need);
  const double c0 = ingest_now();
  g->stage_waits += 1;
  while (g->handled.load(std::memory_order_acquire) < need && !g->q_error.load(std::memory_order_relaxed)) __builtin_ia32_pause();
  g->push_wait_s += ingest_now() - c0;
  return ingest_take_error(g);
}

}  // namespace

extern "C" {

int xm_ingest_create(xm_handle* h, const xm_ingest_config* cfg, xm_ingest** out) {
  if (!h || !cfg || !out) return fail(XM_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(xm_ingest_config)) return fail(XM_ERR_INVALID, "xm_ingest_config.struct_size mismatch");
  if (cfg->projector_fps <= 0) return fail(XM_ERR_INVALID, "projector_fps must be positive");
  XM_ENTER(h);
  xm_ingest* g = new (std::nothrow) xm_ingest();
  if (!g) return fail(XM_ERR_NOMEM, "out of host memory");
  g->h = h;
  g->cfg = *cfg;
  const u64 want_cap = cfg->capacity_events ? cfg->capacity_events : (1u << 21);
  g->capacity = 1;
  while (g->capacity < want_cap) g->capacity <<= 1;  // the ring is indexed by (absolute stream index) & (capacity - 1)
  g->max_packet = cfg->max_packet_events ? cfg->max_packet_events : (1u << 19);
  if (g->capacity >= 0x7fffffffull || g->max_packet * 2 > g->capacity || g->max_packet > (u64)ING_MAX_BLOCKS * ING_EPB) {
    delete g;
    return fail(XM_ERR_INVALID, "capacity must be < 2^31 events and at least twice max_packet_events (itself at most %llu)",
                (unsigned long long)ING_MAX_BLOCKS * ING_EPB);
  }
  // packets the ingest stream may run ahead of the frame kernels: each costs one packet's worth of ring (the room rule)
  g->ahead = g->capacity >= 8 * g->max_packet ? (int)std::min<u64>(3, g->capacity / g->max_packet / 4) : 0;
  g->period = 1e6 / (double)cfg->projector_fps;                       // trigger_finder.py: 1e6 / self.projector_fps (float)
  g->act_thresh = cfg->activity_thresh_us > 0 ? cfg->activity_thresh_us : (long long)(1e6 / cfg->projector_fps);  // pipe:65-68
  if (g->cfg.pause_thresh_us <= 0) g->cfg.pause_thresh_us = 40;       // trigger_finder.py:98
  if (g->cfg.min_events_per_frame <= 0) g->cfg.min_events_per_frame = 1000;  // trigger_finder.py:8
  if (g->cfg.min_events_per_frame < 4) {  // the cut is evs[prev + 2 : next - 2] (trigger_finder.py:172): fewer than 4 events between
    delete g;                             // two pauses would be an empty frame, on which the reference's t.min() raises
    return fail(XM_ERR_INVALID, "min_events_per_frame must be >= 4 (the frame is evs[prev + 2 : next - 2])");
  }
  if (const char* e = dbg_opt("XM_INGEST_CLEAR_EVERY")) g->clear_every = (uint64_t)std::max(1, atoi(e));  // tests: exercise the tag clear
  g->ring = cfg->result_ring > 0 ? cfg->result_ring : 8;
  const size_t cam_px = (size_t)h->tb.cam_w * h->tb.cam_h;
  const size_t px = (size_t)h->out_w * h->out_h;
#define ING_TRY(expr)                                                                 \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) {                                                           \
      int rc_ = fail(XM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));      \
      xm_ingest_destroy(g);                                                           \
      return rc_;                                                                     \
    }                                                                                 \
  } while (0)
  int lo = 0, hi = 0;
  ING_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
  // "XM_INGEST_PRIOS": four letters h / n / l = the priority pools of the ingest, frame, copy and out stream (A/B; default below).
  // The streams come from the process's set for this device (ingest_stream_set below) when nobody else has it.
  const char* pr = dbg_opt("XM_INGEST_PRIOS");
  if (!pr || strlen(pr) != 4) pr = "hhnh";
  const auto prio_of = [&](char c) { return c == 'l' ? lo : c == 'n' ? (lo + hi) / 2 : hi; };
  g->streams_borrowed = !dbg_opt("XM_INGEST_OWN_STREAMS") && ingest_stream_set(h->cfg.device, true, nullptr);
  hipStream_t* set = g->own_streams;
  if (g->streams_borrowed) (void)ingest_stream_set(h->cfg.device, false, &set);
  for (int i = 0; i < 4; ++i)
    if (!set[i]) ING_TRY(hipStreamCreateWithPriority(&set[i], hipStreamNonBlocking, prio_of(pr[i])));
  g->stream = set[0];        // ingest kernels
  g->frame_stream = set[1];  // the cut frames' kernels
  g->copy_stream = set[2];   // H2D of a packet beside the kernels of the previous one
  g->out_stream = set[3];    // the result frames' copies + sequence numbers
  for (auto& e : g->copied_ev) ING_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : g->k1_ev) ING_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : g->k2_ev) ING_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : g->out_ev) ING_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  if (const char* e = dbg_opt("XM_INGEST_OUT_PIECE")) g->out_piece = std::max<size_t>(2u << 20, (size_t)atoll(e));
  if (const char* e = dbg_opt("XM_INGEST_OUT_SERIAL")) g->out_on_frame_stream = e[0] == '1';
  g->opt_out_no_query = dbg_opt("XM_INGEST_OUT_NO_QUERY") != nullptr;
  if (const char* e = dbg_opt("XM_INGEST_HOST_SEQ")) g->host_seq = e[0] != '0';
  g->opt_evt3_out_stream = dbg_opt("XM_INGEST_EVT3_OUT_STREAM") != nullptr;
  g->opt_trace = dbg_opt("XM_INGEST_TRACE") != nullptr;
  if (const char* e = dbg_opt("XM_INGEST_ACT_FUSE")) g->opt_act_fuse = e[0] != '0';
  IngestDev& d = g->dev;
  d.cap = g->capacity;
  d.room = g->max_packet * (u64)(1 + g->ahead);
  d.mirror = g->capacity / 2;  // frames of up to half the ring are contiguous wherever they start
  d.pcap = g->capacity * 2;    // (a pause per live event + the stale head the trigger finder has not skipped yet)
  ING_TRY(hipMalloc((void**)&d.buf, (d.cap + d.mirror) * 16));
  ING_TRY(hipMalloc((void**)&d.pring, d.pcap * 8));
  ING_TRY(hipMalloc((void**)&d.blk, sizeof(IngBlk) * ING_MAX_BLOCKS));
  if (cfg->activity_filter) {
    int rc_ = act_alloc(&d.act, h->tb.cam_w, h->tb.cam_h, g->act_thresh, (size_t)g->max_packet, 2);
    d.act.self_counts = (cfg->flags & XM_INGEST_ACT_SELF) ? 1 : 0;
    g->act_base = d.act;
    if (rc_) {
      xm_ingest_destroy(g);
      return rc_;
    }
  }
  d.cam_w = h->tb.cam_w;
  d.cam_h = h->tb.cam_h;
  d.pause_thresh = g->cfg.pause_thresh_us;
  d.period = g->period;
  d.min_events = (u32)g->cfg.min_events_per_frame;
  d.ring = (u32)g->ring;
  ING_TRY(hipMalloc((void**)&d.st, sizeof(IngestState)));
  ING_TRY(hipMemset(d.st, 0, sizeof(IngestState)));
  ING_TRY(hipMalloc((void**)&g->d_descs, sizeof(FrameDesc) * xm_ingest::VRING));
  ING_TRY(hipMemset(g->d_descs, 0, sizeof(FrameDesc) * xm_ingest::VRING));
  ING_TRY(hipMalloc((void**)&g->d_infos, sizeof(IngFrameInfo) * xm_ingest::VRING));
  ING_TRY(hipMemset(g->d_infos, 0, sizeof(IngFrameInfo) * xm_ingest::VRING));
  ING_TRY(hipHostMalloc((void**)&g->h_verdicts, sizeof(IngVerdict) * xm_ingest::VRING, hipHostMallocMapped));
  memset(g->h_verdicts, 0, sizeof(IngVerdict) * xm_ingest::VRING);
  ING_TRY(hipHostGetDevicePointer((void**)&g->d_verdicts, g->h_verdicts, 0));
  ING_TRY(hipMalloc((void**)&d.key_frame, h->key_cells * sizeof(u64)));
  ING_TRY(hipMalloc((void**)&d.slot, sizeof(SlotState)));
  ING_TRY(hipMemset(d.slot, 0, sizeof(SlotState)));
  // (a memset of device memory may return before it has run and the ingest's streams do not wait for the default stream:
  //  k_reset_slot initialises the extrema slots inside these bytes -- seen once as a first frame with a wrong time normalisation)
  ING_TRY(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_reset_slot, dim3(1024), dim3(BLOCK), 0, g->frame_stream, d.slot, d.key_frame, (u64)h->key_cells, (unsigned char*)nullptr);
  ING_TRY(hipGetLastError());
  for (int i = 0; i < xm_ingest::STAGE; ++i) {
    // (the pinned twins h_pkt[] are allocated by the first PAGEABLE push: callers that push pinned packets or RAW words never
    //  pay for 16 x max_packet x 16 bytes of page-locked memory)
    ING_TRY(hipMalloc((void**)&g->d_pkt[i], g->max_packet * 16));
    if (!g->d_pkt_n) ING_TRY(hipMalloc((void**)&g->d_pkt_n, xm_ingest::STAGE * sizeof(u32)));
  }
  ING_TRY(hipHostMalloc((void**)&g->h_status, sizeof(IngestStatus) * g->ring, hipHostMallocMapped));
  memset(g->h_status, 0, sizeof(IngestStatus) * g->ring);
  g->h_depth.assign(g->ring, nullptr);
  g->h_bgr.assign(g->ring, nullptr);
  g->slot_frame.assign(g->ring, 0);
  for (int i = 0; i < g->ring; ++i) {
    if (cfg->want_depth) ING_TRY(hipHostMalloc((void**)&g->h_depth[i], px * 4, hipHostMallocDefault));
    if (cfg->want_bgr) ING_TRY(hipHostMalloc((void**)&g->h_bgr[i], px * 3, hipHostMallocDefault));
  }
  for (int i = 0; i < xm_ingest::NOUT; ++i) {
    if (cfg->want_depth) ING_TRY(hipMalloc((void**)&g->d_out_depth[i], px * 4));
    if (cfg->want_bgr) ING_TRY(hipMalloc((void**)&g->d_out_bgr[i], px * 3));
  }
  ING_TRY(hipMalloc((void**)&g->d_depth_ring, sizeof(float*) * xm_ingest::NOUT));
  ING_TRY(hipMalloc((void**)&g->d_bgr_ring, sizeof(uint8_t*) * xm_ingest::NOUT));
  ING_TRY(hipMemcpy(g->d_depth_ring, g->d_out_depth, sizeof(float*) * xm_ingest::NOUT, hipMemcpyHostToDevice));
  ING_TRY(hipMemcpy(g->d_bgr_ring, g->d_out_bgr, sizeof(uint8_t*) * xm_ingest::NOUT, hipMemcpyHostToDevice));
  d.nout = xm_ingest::NOUT;
  // (the first DMA into a pinned buffer is several times slower than the later ones -- seen as 0.2 ms per frame for the first
  //  round through the ring: every entry takes one copy now)
  for (int i = 0; i < g->ring; ++i) {
    if (cfg->want_depth) ING_TRY(hipMemcpyAsync(g->h_depth[i], g->d_out_depth[i % xm_ingest::NOUT], px * 4, hipMemcpyDeviceToHost, g->out_stream));
    if (cfg->want_bgr) ING_TRY(hipMemcpyAsync(g->h_bgr[i], g->d_out_bgr[i % xm_ingest::NOUT], px * 3, hipMemcpyDeviceToHost, g->out_stream));
  }
  ING_TRY(hipStreamSynchronize(g->out_stream));
  d.depth_ring = g->d_depth_ring;
  d.bgr_ring = g->d_bgr_ring;
  ING_TRY(hipDeviceSynchronize());  // (the memsets above ran on the default stream, which the ingest's non-blocking streams do not wait for)
#undef ING_TRY
  if (!(cfg->flags & XM_INGEST_NO_LAUNCH_THREAD)) {
    g->threaded = true;
    g->th = std::thread(ingest_thread_main, g);
    if (!dbg_opt("XM_INGEST_NO_COPY_THREAD")) {
      g->copy_threaded = true;
      g->copy_th = std::thread(ingest_copy_thread_main, g);
    }
    if (!g->out_on_frame_stream && !dbg_opt("XM_INGEST_OUT_INLINE")) {
      g->out_threaded = true;
      g->out_th = std::thread(ingest_out_main, g);
    }
  }
  *out = g;
  return XM_OK;
}

void xm_ingest_destroy(xm_ingest* g) {
  if (!g) return;
  (void)hipSetDevice(g->h->cfg.device);
  if (g->threaded) {
    xm_ingest::Job stop;
    stop.kind = 2;
    ingest_post(g, stop);
    if (g->copy_th.joinable()) g->copy_th.join();  // (forwards the stop behind everything else, then leaves)
    if (g->th.joinable()) g->th.join();
    g->threaded = false;
    g->copy_threaded = false;
  }
  if (g->out_threaded) {  // (behind the launch thread: nobody posts any more; the queue is drained before the thread leaves)
    (void)ingest_out_drain(g);  // (every posted frame is taken before the thread is told to leave)
    {
      std::lock_guard<std::mutex> lk(g->out_mu);
      g->out_stop.store(true, std::memory_order_seq_cst);
    }
    g->out_cv.notify_all();
    if (g->out_th.joinable()) g->out_th.join();
    g->out_threaded = false;
  }
  if (g->copy_stream) (void)hipStreamSynchronize(g->copy_stream);
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  if (g->frame_stream) (void)hipStreamSynchronize(g->frame_stream);
  if (g->out_stream) (void)hipStreamSynchronize(g->out_stream);
  if (g->opt_trace)
    fprintf(stderr, "[ingest] %llu packets, %llu frames, ahead %d: launch side %.3f ms in jobs, of which %.3f ms waiting for verdicts and %.3f ms "
            "issuing frames; caller %.3f ms in push, %.3f ms of it waiting for staging entries\n", (unsigned long long)g->issued,
            (unsigned long long)g->frames_issued, g->ahead, g->t_jobs_s * 1e3, g->t_block_s * 1e3, g->t_frames_s * 1e3, g->push_host_s * 1e3,
            g->push_wait_s * 1e3);
  if (g->opt_trace)
    fprintf(stderr, "[ingest] out side (%s): %.3f ms enqueuing %llu frames' copies + sequence numbers; the launch side waited %.3f ms for it\n",
            g->cfg.flags & XM_INGEST_NO_LAUNCH_THREAD ? "inline" : "a thread of its own", g->t_out_s * 1e3, (unsigned long long)g->out_done.load(),
            g->t_out_wait_s * 1e3);
  IngestDev& d = g->dev;
  if (d.buf) (void)hipFree(d.buf);
  if (d.pring) (void)hipFree(d.pring);
  if (d.blk) (void)hipFree(d.blk);
  act_free(&g->act_base);
  d.act = ActDev{};
  if (d.st) (void)hipFree(d.st);
  if (d.key_frame) (void)hipFree(d.key_frame);
  if (d.slot) (void)hipFree(d.slot);
  if (g->d_descs) (void)hipFree(g->d_descs);
  if (g->d_infos) (void)hipFree(g->d_infos);
  if (g->h_verdicts) (void)hipHostFree(g->h_verdicts);
  if (g->d_pkt_n) (void)hipFree(g->d_pkt_n);
  if (g->d_depth_ring) (void)hipFree(g->d_depth_ring);
  if (g->d_bgr_ring) (void)hipFree(g->d_bgr_ring);
  for (int i = 0; i < xm_ingest::NOUT; ++i) {
    if (g->d_out_depth[i]) (void)hipFree(g->d_out_depth[i]);
    if (g->d_out_bgr[i]) (void)hipFree(g->d_out_bgr[i]);
  }
  for (int i = 0; i < xm_ingest::STAGE; ++i) {
    if (g->h_pkt[i]) (void)hipHostFree(g->h_pkt[i]);
    if (g->d_pkt[i]) (void)hipFree(g->d_pkt[i]);
  }
  if (g->h_status) (void)hipHostFree(g->h_status);
  for (auto p : g->h_depth) if (p) (void)hipHostFree(p);
  for (auto p : g->h_bgr) if (p) (void)hipHostFree(p);
  if (xm_frame_pool* pl = g->pool) {  // spare buffers go now, buffers in consumers' hands when they come back (the last one takes the pool along)
    bool last;
    {
      std::lock_guard<std::mutex> lk(pl->mu);
      pl->closed = true;
      for (auto& v : pl->free_bufs) {
        for (void* p : v) (void)hipHostFree(p);
        v.clear();
      }
      last = pl->outstanding == 0;
    }
    if (last) delete pl;
    g->pool = nullptr;
  }

  for (auto& e : g->copied_ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : g->k1_ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : g->k2_ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : g->out_ev) if (e) (void)hipEventDestroy(e);
  for (auto& s : g->own_streams) if (s) (void)hipStreamDestroy(s);
  if (g->streams_borrowed) ingest_stream_release(g->h->cfg.device);
  delete g;
}

static int ingest_push(xm_ingest* g, const void* eventcd16, size_t n, bool pinned);
int xm_ingest_push(xm_ingest* g, const void* eventcd16, size_t n) { return ingest_push(g, eventcd16, n, false); }
int xm_ingest_push_pinned(xm_ingest* g, const void* eventcd16_pinned, size_t n) { return ingest_push(g, eventcd16_pinned, n, true); }

static int ingest_push(xm_ingest* g, const void* eventcd16, size_t n, bool pinned) {
  if (!g || (n && !eventcd16)) return fail(XM_ERR_INVALID, "NULL argument");
  const double c0 = ingest_now();
  xm_handle* h = g->h;
  if (n > g->max_packet) return fail(XM_ERR_TOO_MANY, "packet of %zu events exceeds max_packet_events %llu", n, (unsigned long long)g->max_packet);
  int rc = ingest_take_error(g);  // an earlier packet's launches failed
  if (rc) return rc;
  if (!g->threaded) HIP_TRY(hipSetDevice(h->cfg.device));
  const int k = g->pkt_next;
  g->pkt_next = (k + 1) % xm_ingest::STAGE;
  if (!pinned && n && !g->h_pkt[k]) {
    // the pinned twins of the staging entries, ALL of them at the first pageable push (one page-locking pause of the stream's
    // first packet instead of one on each of its first 16 packets); callers that push pinned packets or RAW words never pay
    HIP_TRY(hipSetDevice(h->cfg.device));
    for (int i = 0; i < xm_ingest::STAGE; ++i)
      if (!g->h_pkt[i]) HIP_TRY(hipHostMalloc((void**)&g->h_pkt[i], g->max_packet * 16, hipHostMallocDefault));
  }
  const uint4* hp = pinned ? (const uint4*)eventcd16 : g->h_pkt[k];
  if ((rc = ingest_wait_entry(g, k))) return rc;
  if (n && !pinned) memcpy(g->h_pkt[k], eventcd16, n * 16);  // pageable memory: through the pinned staging ring
  g->posted += 1;
  g->pkt_push[k] = g->posted;
  g->push_t[g->posted % xm_ingest::VRING] = c0;
  xm_ingest::Job j;
  j.kind = 0; j.k = k; j.n = n; j.host = hp; j.push_no = g->posted;
  rc = ingest_submit(g, j, false);
  g->push_host_s += ingest_now() - c0;
  g->push_calls += 1;
  return rc;
}

static int ingest_poll(xm_ingest* g, xm_ingest_frame* out, bool owned) {
  if (!g || !out) return fail(XM_ERR_INVALID, "NULL argument");
  if (!g->threaded && g->next_verdict <= g->issued) {  // (no launch thread: a frame whose verdict has arrived meanwhile goes out now)
    int rc = ingest_handle_verdicts(g, 0);
    if (rc) return rc;
  }
  const int slot = (int)(g->next_seq % (uint64_t)g->ring);
  const IngestStatus* st = g->h_status + slot;
  const uint64_t want = g->next_seq + 1;  // the entry's seq once frame next_seq has been published
  const uint64_t seq = __atomic_load_n(&st->seq, __ATOMIC_ACQUIRE);
  if (seq < want) return 0;  // not there yet
  IngestStatus v;
  memcpy(&v, st, sizeof v);
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  const uint64_t seq2 = __atomic_load_n(&st->seq, __ATOMIC_ACQUIRE);  // did the producer rewrite the entry while it was read?
  bool lapped = seq > want || seq2 != seq;  // the ring holds a later frame here (or is being rewritten): this one is lost
  uint64_t newest_slot = 0;
  memset(out, 0, sizeof *out);
  out->seq = g->next_seq;
  if (!lapped) {
    out->depth = g->h_depth[slot];
    out->bgr = g->h_bgr[slot];
  }
  if (!lapped && owned && (out->depth || out->bgr)) {
    // The slot's buffers leave with the frame and the slot gets spare ones -- in one step with the out side's "these are the
    // buffers of frame f" (res_mu): a slot that says another frame by now has been lapped, its buffers are a DMA's target.
    if (!g->pool) {
      xm_frame_pool* pl = new (std::nothrow) xm_frame_pool();
      if (pl) {
        pl->device = g->h->cfg.device;
        const size_t px = (size_t)g->h->out_w * g->h->out_h;
        pl->bytes[0] = px * 4;
        pl->bytes[1] = px * 3;
        pl->cap = 1024;
        if (const char* e = dbg_opt("XM_INGEST_POOL_CAP")) pl->cap = (size_t)std::max(0, atoi(e));
        g->pool = pl;
      }
    }
    void* spare[2] = {nullptr, nullptr};
    bool have = g->pool != nullptr;
    if (have) {
      xm_frame_pool* pl = g->pool;
      std::lock_guard<std::mutex> lk(pl->mu);
      for (int kind = 0; kind < 2 && have; ++kind) {
        if (!(kind == 0 ? out->depth != nullptr : out->bgr != nullptr)) continue;
        if (!pl->free_bufs[kind].empty()) {
          spare[kind] = pl->free_bufs[kind].back();
          pl->free_bufs[kind].pop_back();
        } else if (pl->allocated < pl->cap && hipSetDevice(pl->device) == hipSuccess &&
                   hipHostMalloc(&spare[kind], pl->bytes[kind], hipHostMallocDefault) == hipSuccess) {
          pl->allocated += 1;
        } else {
          (void)hipGetLastError();
          spare[kind] = nullptr;
          have = false;
        }
      }
      if (!have)  // (not both: what was taken goes back; the caller copies this frame out of the ring as xm_ingest_poll's callers do)
        for (int kind = 0; kind < 2; ++kind)
          if (spare[kind]) pl->free_bufs[kind].push_back(spare[kind]);
    }
    if (have) {
      std::lock_guard<std::mutex> lk(g->res_mu);
      newest_slot = g->slot_frame[slot];
      if (newest_slot == want) {
        if (out->depth) g->h_depth[slot] = (float*)spare[0];
        if (out->bgr) g->h_bgr[slot] = (uint8_t*)spare[1];
        out->owned = 1;
      } else {
        lapped = true;
      }
    }
    if (have && out->owned) {
      std::lock_guard<std::mutex> lk(g->pool->mu);
      g->pool->outstanding += (out->depth ? 1 : 0) + (out->bgr ? 1 : 0);
    } else if (have) {
      std::lock_guard<std::mutex> lk(g->pool->mu);
      for (int kind = 0; kind < 2; ++kind)
        if (spare[kind]) g->pool->free_bufs[kind].push_back(spare[kind]);
    }
  }
  out->lost = lapped ? 1 : 0;
  if (lapped) {
    out->depth = nullptr;
    out->bgr = nullptr;
    if (g->opt_trace) fprintf(stderr, "[ingest] lapped: slot %d want %llu seq %llu seq2 %llu (frames issued %llu, pushes issued %llu)\n", slot,
                                            (unsigned long long)want, (unsigned long long)seq, (unsigned long long)seq2,
                                            (unsigned long long)g->frames_issued, (unsigned long long)g->issued);
    // Nothing of the entry can be trusted for frame next_seq (no statistics, no images: depth / bgr stay NULL).
    // Resume with the oldest frame the ring may still hold intact.
    const uint64_t newest = std::max(std::max(seq, seq2), newest_slot);  // >= want + ring - 1
    g->next_seq = std::max<uint64_t>(g->next_seq + 1, newest >= (uint64_t)g->ring ? newest - (uint64_t)g->ring : 0);
    return 1;
  }
  out->n_events = v.n_events;
  out->t_first = v.t_first;
  out->t_last = v.t_last;
  out->n_inliers = v.n_inliers;
  out->n_index_errors = v.n_index_errors;
  out->live_after = v.live_after;
  out->overflow = v.overflow;
  out->push_seq = v.push_seq;
  out->push_to_publish_us = v.latency_us;
  g->next_seq += 1;
  return 1;
}

int xm_ingest_poll(xm_ingest* g, xm_ingest_frame* out) { return ingest_poll(g, out, false); }

int xm_ingest_poll_owned(xm_ingest* g, xm_ingest_frame* out, xm_frame_pool** pool) {
  const int rc = ingest_poll(g, out, true);
  if (pool) *pool = rc == 1 && out->owned ? g->pool : nullptr;
  return rc;
}

void xm_frame_pool_release(xm_frame_pool* pl, void* buffer, int kind) {
  if (!pl || !buffer || kind < 0 || kind > 1) return;
  bool last = false, free_it = false;
  {
    std::lock_guard<std::mutex> lk(pl->mu);
    if (pl->outstanding) pl->outstanding -= 1;
    if (pl->closed) {
      free_it = true;
      last = pl->outstanding == 0;
    } else {
      pl->free_bufs[kind].push_back(buffer);
    }
  }
  if (free_it) (void)hipHostFree(buffer);
  if (last) delete pl;
}

int xm_frame_pool_stats(xm_frame_pool* pl, uint64_t* allocated, uint64_t* outstanding, uint64_t* spare) {
  if (!pl) return fail(XM_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> lk(pl->mu);
  if (allocated) *allocated = pl->allocated;
  if (outstanding) *outstanding = pl->outstanding;
  if (spare) *spare = pl->free_bufs[0].size() + pl->free_bufs[1].size();
  return XM_OK;
}

int xm_ingest_backlog(xm_ingest* g, int wait_below, uint64_t* backlog) {
  if (!g) return fail(XM_ERR_INVALID, "NULL argument");
  const auto now = [&]() -> uint64_t {
    // (handled first: a frame is counted in frames_issued_pub before its packet is in handled)
    const uint64_t hd = g->threaded ? g->handled.load(std::memory_order_acquire) : g->next_verdict - 1;
    const uint64_t fi = g->threaded ? g->frames_issued_pub.load(std::memory_order_acquire) : g->frames_issued;
    const uint64_t posted = g->threaded ? g->posted : g->issued;
    return (fi - std::min(fi, g->next_seq)) + (posted - std::min(posted, hd));
  };
  uint64_t b = now();
  if (wait_below > 0) {
    const double c0 = ingest_now();
    bool waited = false;
    for (;;) {
      const uint64_t posted = g->threaded ? g->posted : g->issued;
      const uint64_t hd = g->threaded ? g->handled.load(std::memory_order_acquire) : g->next_verdict - 1;
      if (b < (uint64_t)wait_below || hd >= posted) break;  // room, or nothing left in flight that waiting could settle
      if (!g->threaded) {
        int rc = ingest_handle_verdicts(g, g->next_verdict);
        if (rc) return rc;
      } else {
        if (g->q_error.load(std::memory_order_relaxed)) return ingest_take_error(g);
        for (int k = 0; k < 64; ++k) __builtin_ia32_pause();
      }
      waited = true;
      b = now();
    }
    if (waited) {  // (counted like a push that waited for a staging entry: the caller's time, spent waiting for the GPU)
      const double dt = ingest_now() - c0;
      g->stage_waits += 1;
      g->push_wait_s += dt;
      g->push_host_s += dt;
    }
  }
  if (backlog) *backlog = b;
  return XM_OK;
}

int xm_ingest_frame_valid(xm_ingest* g, uint64_t seq) {
  if (!g) return fail(XM_ERR_INVALID, "NULL argument");
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  // (k_ing_publish zeroes the slot's sequence number before anything of the next frame is written into the slot's buffers: the
  //  frame kernels' K2 writes device memory, the copy into this slot comes behind k_ing_publish on the frame stream's event)
  return __atomic_load_n(&g->h_status[seq % (uint64_t)g->ring].seq, __ATOMIC_ACQUIRE) == seq + 1 ? 1 : 0;
}

int xm_ingest_flush(xm_ingest* g) {
  if (!g) return fail(XM_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(g->h->cfg.device));
  xm_ingest::Job j;
  j.kind = 4;
  return ingest_submit(g, j, true);
}

int xm_ingest_reset(xm_ingest* g) {
  if (!g) return fail(XM_ERR_INVALID, "NULL argument");
  int rc = xm_ingest_flush(g);
  if (rc) return rc;
  // RobustTriggerFinder.reset(): the buffered events are discarded (trigger_finder.py:116-119).  The stream indices start over
  // (nothing live refers to the old ones); the frame counter and the sticky overflow count go on.
  IngestState z;
  HIP_TRY(hipMemcpy(&z, g->dev.st, sizeof z, hipMemcpyDeviceToHost));
  z.start_abs = z.write_abs = z.p_head = z.p_tail = 0;
  z.last_t = 0;
  z.span_ok = 0;
  HIP_TRY(hipMemcpy(g->dev.st, &z, sizeof z, hipMemcpyHostToDevice));
  // The activity filter's history goes too: a reset is "the stream starts over" (the reference resets when a recording loops,
  // depth_reprojection.py:76) and its stamps then start below everything the history holds -- against the old history every
  // event with a neighbour that ever fired would pass.  (Metavision's filter object keeps its state there; the frames behind
  // the first period of a loop are what differs.)
  if (g->act_base.last_ts) {
    std::vector<long long> init((size_t)g->act_base.cam_w * g->act_base.cam_h, ING_NO_TS);
    HIP_TRY(hipMemcpy(g->act_base.last_ts, init.data(), init.size() * 8, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipDeviceSynchronize());  // (default-stream work: the ingest's non-blocking streams do not wait for it)
  return XM_OK;
}

// the device's counters after everything pushed so far has run (synchronises like xm_ingest_flush)
int xm_ingest_device_stats(xm_ingest* g, uint64_t* frames_cut, uint64_t* events_appended, uint64_t* events_dropped, uint64_t* events_live) {
  if (!g) return fail(XM_ERR_INVALID, "NULL argument");
  int rc = xm_ingest_flush(g);
  if (rc) return rc;
  IngestState z;
  HIP_TRY(hipMemcpy(&z, g->dev.st, sizeof z, hipMemcpyDeviceToHost));
  if (frames_cut) *frames_cut = z.frames;
  if (events_appended) *events_appended = z.appended;
  if (events_dropped) *events_dropped = z.overflow;
  if (events_live) *events_live = z.write_abs - z.start_abs;
  return XM_OK;
}

int xm_ingest_host_stats(xm_ingest* g, uint64_t* pushes, double* host_seconds_in_push, uint64_t* staging_waits, double* seconds_waiting) {
  if (!g) return fail(XM_ERR_INVALID, "NULL argument");
  if (pushes) *pushes = g->push_calls;
  if (host_seconds_in_push) *host_seconds_in_push = g->push_host_s;
  if (staging_waits) *staging_waits = g->stage_waits;
  if (seconds_waiting) *seconds_waiting = g->push_wait_s;
  return XM_OK;
}

// ---- the activity filter alone ------------------------------------------------------------------------------------------
// What `act_filter.process_events(pos_events_buf, act_out_buf)` is in the reference's pipe (depth_reprojection_pipe.py:116-117)
// for a host that keeps the trigger finder on the CPU: one packet of records in, a keep flag per event out.  Same kernels and
// state as the ingest's filter (xmaps_ingest.hpp); every event handed in takes part (the pipe hands it positive events).
struct xm_activity {
  xm_handle* h = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  size_t max_packet = 0;
  ActDev act{};
  uint4* h_pkt = nullptr;  // pinned staging
  uint4* d_pkt = nullptr;
  unsigned char* h_keep = nullptr;
};

int xm_activity_create(xm_handle* h, int64_t thresh_us, size_t max_packet_events, xm_activity** out) {
  if (!h || !out) return fail(XM_ERR_INVALID, "NULL argument");
  *out = nullptr;
  XM_ENTER(h);
  xm_activity* f = new (std::nothrow) xm_activity();
  if (!f) return fail(XM_ERR_NOMEM, "out of host memory");
  f->h = h;
  f->device = h->cfg.device;
  f->max_packet = max_packet_events ? max_packet_events : ((size_t)1 << 19);
  int rc = act_alloc(&f->act, h->tb.cam_w, h->tb.cam_h, thresh_us, f->max_packet);
  const auto tr = [&](hipError_t e, const char* what) {
    if (!rc && e != hipSuccess) rc = fail(XM_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
  };
  if (!rc) tr(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking), "hipStreamCreateWithFlags");
  if (!rc) tr(hipHostMalloc((void**)&f->h_pkt, f->max_packet * 16, hipHostMallocDefault), "hipHostMalloc");
  if (!rc) tr(hipHostMalloc((void**)&f->h_keep, f->max_packet, hipHostMallocDefault), "hipHostMalloc");
  if (!rc) tr(hipMalloc((void**)&f->d_pkt, f->max_packet * 16), "hipMalloc");
  if (rc) {
    xm_activity_destroy(f);
    return rc;
  }
  *out = f;
  return XM_OK;
}

int xm_activity_set_rule(xm_activity* f, int self_counts) {
  if (!f) return fail(XM_ERR_INVALID, "NULL argument");
  f->act.self_counts = self_counts ? 1 : 0;  // (by value in every launch: takes effect with the next packet)
  return XM_OK;
}

void xm_activity_destroy(xm_activity* f) {
  if (!f) return;
  (void)hipSetDevice(f->device);
  if (f->stream) (void)hipStreamSynchronize(f->stream);
  act_free(&f->act);
  if (f->h_pkt) (void)hipHostFree(f->h_pkt);
  if (f->h_keep) (void)hipHostFree(f->h_keep);
  if (f->d_pkt) (void)hipFree(f->d_pkt);
  if (f->stream) (void)hipStreamDestroy(f->stream);
  delete f;
}

int xm_activity_process(xm_activity* f, const void* eventcd16, size_t n, uint8_t* keep_out, size_t* n_kept) {
  if (!f || (n && (!eventcd16 || !keep_out))) return fail(XM_ERR_INVALID, "NULL argument");
  if (n_kept) *n_kept = 0;
  HIP_TRY(hipSetDevice(f->device));
  size_t kept = 0;
  for (size_t a = 0; a < n; a += f->max_packet) {  // (a longer packet: piece by piece -- the rule does not depend on the cut)
    const size_t m = std::min(f->max_packet, n - a);
    memcpy(f->h_pkt, (const char*)eventcd16 + a * 16, m * 16);
    HIP_TRY(hipMemcpyAsync(f->d_pkt, f->h_pkt, m * 16, hipMemcpyHostToDevice, f->stream));
    const unsigned gx = (unsigned)((m + ING_THREADS - 1) / ING_THREADS);
    hipLaunchKernelGGL(k_act_first, dim3(gx), dim3(ING_THREADS), 0, f->stream, f->act, (const uint4*)f->d_pkt, (const u32*)nullptr, (u32)m, 0);
    hipLaunchKernelGGL(k_act_mark, dim3(gx), dim3(ING_THREADS), 0, f->stream, f->act, (const uint4*)f->d_pkt, (const u32*)nullptr, (u32)m, 0);
    hipLaunchKernelGGL(k_act_update, dim3(gx), dim3(ING_THREADS), 0, f->stream, f->act, (const uint4*)f->d_pkt, (u32)m, 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(f->h_keep, f->act.keep, m, hipMemcpyDeviceToHost, f->stream));
    HIP_TRY(hipStreamSynchronize(f->stream));
    memcpy(keep_out + a, f->h_keep, m);
    for (size_t i = 0; i < m; ++i) kept += f->h_keep[i] != 0;
  }
  if (n_kept) *n_kept = kept;
  return XM_OK;
}

int xm_activity_stats(xm_activity* f, uint64_t* sequential_packets) {
  if (!f) return fail(XM_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(f->device));
  HIP_TRY(hipStreamSynchronize(f->stream));
  u32 c[4] = {0, 0, 0, 0};
  HIP_TRY(hipMemcpy(c, f->act.ctl, sizeof c, hipMemcpyDeviceToHost));
  if (sequential_packets) *sequential_packets = c[2];
  return XM_OK;
}

int xm_ingest_activity_stats(xm_ingest* g, uint64_t* sequential_packets) {
  if (!g) return fail(XM_ERR_INVALID, "NULL argument");
  if (sequential_packets) *sequential_packets = 0;
  if (!g->act_base.last_ts) return XM_OK;
  int rc = xm_ingest_flush(g);
  if (rc) return rc;
  u32 c[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // (both sets' control words)
  HIP_TRY(hipMemcpy(c, g->act_base.ctl, sizeof c, hipMemcpyDeviceToHost));
  if (sequential_packets) *sequential_packets = (uint64_t)c[2] + c[6];
  return XM_OK;
}

int xm_ingest_fused_first_passes(xm_ingest* g, uint64_t* n) {
  if (!g || !n) return fail(XM_ERR_INVALID, "NULL argument");
  int rc = xm_ingest_flush(g);  // (the launch thread's count: read when it is idle)
  if (rc) return rc;
  *n = g->act_fused_count;
  return XM_OK;
}

int xm_activity_reset(xm_activity* f) {
  if (!f) return fail(XM_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(f->device));
  HIP_TRY(hipStreamSynchronize(f->stream));
  std::vector<long long> init((size_t)f->act.cam_w * f->act.cam_h, ING_NO_TS);
  HIP_TRY(hipMemcpyAsync(f->act.last_ts, init.data(), init.size() * 8, hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipStreamSynchronize(f->stream));
  return XM_OK;
}

}  // extern "C"
