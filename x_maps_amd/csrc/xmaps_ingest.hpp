// xmaps_ingest.hpp -- device-side ingest ("next" row N2 of SURVEY.md section 8(f)): the raw camera stream stays in HBM.
//
// What the reference does on the host per packet (python/depth_reprojection_pipe.py:110-119, python/trigger_finder.py:128-189):
//   PolarityFilterAlgorithm(1)  ->  ActivityNoiseFilterAlgorithm(w, h, int(1e6 / fps))  ->  RobustTriggerFinder:
//   buffer packets until they span one projector period; pauses = nonzero(diff(t) >= 40 us); the first pair of consecutive
//   pauses more than half a period apart decides: at most one period apart and > 1000 events -> frame = evs[prev+2 : next-2],
//   keep evs[next-2:]; otherwise drop everything up to next; no such pair -> the whole buffer is dropped.
// Here the same chain runs as THREE kernels per packet over a device-resident event ring (round 4; seventeen launches until
// then -- a kernel boundary costs ~1.5 us on this chip and a launch ~3.5 us of host time, and a live camera's quarter-period
// packets make 4-6 pushes per frame):
//   k_ing_count    per block of 512 packet events: how many pass the filters, how many pauses lie between them, the first and
//                  last kept time stamp
//   k_ing_append   the same block-local work again + the blocks' records summed in front of it (every block sums its
//                  predecessors itself: <= 4096 records, no scan kernel): kept events -> the ring in stream order, the pauses
//                  between them -> the pause ring (absolute stream indices, ascending).  Pauses are found ONCE, when an event
//                  is appended, not by re-scanning the whole buffer with every packet as the reference does
//   k_ing_segment  one block: commits the counters, then RobustTriggerFinder.find_trigger over the pause ring; the frame that is
//                  cut is described by a FrameDesc written BY THE DEVICE (pointer into the ring + count), which the multi-frame
//                  K0/K1/K2 launches read -- no index ever travels to the host
// k_ing_segment ends by telling the HOST, in pinned memory, whether this packet cut a frame and how many events it has (16 bytes,
// the only thing the host ever learns about the stream): the frame kernels K0 -> K1 -> K2 and the publishing launch are issued
// only for packets that did cut one, with exact grids, on a SECOND stream; K2 writes the frame into device memory and a DMA copy
// behind it on that stream takes it to the pinned result ring (6 MB of BGR = ~115 us per frame at PCIe speed on the reference's
// projector), which no longer holds up the next packets' ingest kernels.  (K2 storing straight into host memory, as it did until
// round 4, blocks every other kernel's stores behind its own for as long as it runs: measured, profiles/r04_ingest.md.)  The ingest stream may run `ahead` packets in front of the frame kernels; the room rule below leaves that many
// packets' worth of space, so that nothing is appended over events a frame kernel is still reading.
// The ring holds `cap` (a power of two) events at absolute stream index & (cap - 1); its first `mirror` entries are written a
// second time behind its end, so that any frame of <= mirror events is CONTIGUOUS in memory wherever it starts (the frame
// kernels take a pointer + count) and nothing is ever moved (until round 4 the live part was copied to a second buffer
// whenever the next packet might not fit).
//
// Activity-noise filter: Metavision's ActivityNoiseFilterAlgorithm ships as a binary in the SDK the reference installs
// (SURVEY.md 8(c)), so its exact rule is unpinned here.  OWN DEFINITION, implemented identically in oracle/ingest_oracle.py
// (which lists the differences from the published OpenEB header as far as they are known):
//   a (positive) event e = (x, y, t) is KEPT iff some event e' EARLIER IN THE STREAM at one of the 8 neighbouring pixels has
//   t - t' <= T (T = int(1e6 / fps)); every positive event, kept or not, then becomes its pixel's latest event.
// Parallel evaluation on the device, exact for any event order and any packet, with nothing decided on the host (round 5; until
// then the host cut a packet into sub-packets by its time stamps, which a chunk decoded on the device does not have there):
//   * events of EARLIER packets qualify through the per-pixel maximum time stamp `last_ts` (exact whatever the order);
//   * inside the packet the time axis is cut into buckets of W = T + 1 us counted from the packet's first stamp.  Of two events
//     in one bucket the earlier one always qualifies for the later one (|t - t'| <= T): a per-(bucket, pixel) MINIMUM packet
//     index answers that.  An event of the bucket before qualifies iff its stamp is within T: a per-(bucket, pixel) MAXIMUM
//     stamp answers that, PROVIDED every event of bucket b - 1 comes before every event of bucket b in the packet; buckets
//     further back never qualify (t - t' >= W + 1).  k_act_first fills both cells with one pair of 32-bit atomics per event and
//     checks the proviso (bucket numbers non-decreasing along the packet, at most ACT_NB of them);
//   * a packet that fails the check (time stamps running backwards by more than a bucket, a chunk spanning more than ACT_NB
//     thresholds) is judged sequentially in groups of 256 events against the running per-pixel maximum (by one block of
//     k_act_mark, or by k_ing_count's blocks one after the other) -- slow (~20 ms per million events) and exact.
// The keep flags go to a byte per event (the ingest: k_ing_count computes them as it counts, k_ing_append reads them and clears the
// cells the packet used; the filter alone: k_act_mark / k_act_update).
#pragma once
#include "xmaps_kernels.hpp"

namespace xm {

constexpr int ING_THREADS = 256, ING_EPT = 2, ING_EPB = ING_THREADS * ING_EPT;  // packet events per block
constexpr int ING_MAX_BLOCKS = 4096;                                            // => packets of up to 2 M events

struct IngestState {      // device
  u64 start_abs;          // absolute stream index of the first live event
  u64 write_abs;          // one past the last live event (events appended so far, minus nothing: indices never restart)
  u64 p_head, p_tail;     // live part of the pause ring (absolute counters; entry = index i with t[i+1] - t[i] >= thresh)
  u64 frames;             // frames cut so far
  u64 appended;           // events appended so far (after the filters)
  u64 published;          // frames whose kernels have run (k_ing_publish counts them)
  long long last_t;       // time stamp of event write_abs - 1 (valid when write_abs > 0)
  u32 overflow;           // events dropped because the ring was full (sticky)
  u32 span_ok;            // scratch: the live part spans at least one period
};

struct IngestStatus {     // pinned host ring entry, written by k_ing_publish when a frame has been produced
  u64 seq;                // frame number + 1; written LAST (system-scope release)
  u64 n_events;
  long long t_first, t_last;
  u64 n_inliers, n_index_errors, n_used;
  u64 live_after;         // events left in the buffer after the cut
  u64 push_seq;           // number of the xm_ingest_push call whose launches cut this frame (the host tightens its bounds with it)
  u32 overflow;
  float latency_us;       // written by the HOST (out thread, just before seq): xm_ingest_push* of that packet entered -> frame published
};

struct IngVerdict {       // pinned host ring entry (one per packet, 64 entries), written by k_ing_segment
  u64 info;               // bit 63: the packet cut a frame; low bits: its number of events
  u64 push_no;            // written LAST (system-scope release): the entry describes this packet
};

struct IngFrameInfo {     // device, one per verdict entry: what k_ing_segment knew when it cut the frame (k_ing_publish reads it)
  u64 frame_no;
  u64 live_after;         // events left in the ring after the cut
  long long t_first, t_last;  // the frame's first / last stamp (read at the cut: the ring is the frame's only up to its K1)
  u32 overflow;
  u32 pad;
};

struct IngBlk {           // what one block of k_ing_count found in its 512 packet events
  u32 kept;               // events that pass the filters
  u32 pauses;             // pauses between consecutive kept events INSIDE the block (the one in front of its first kept event
                          // depends on an earlier block: k_ing_append / k_ing_segment add it from first_t / last_t)
  long long first_t, last_t;
  u64 pad;
};
static_assert(sizeof(IngBlk) == 32, "IngBlk layout");

// ---- activity filter state (see the header comment) -------------------------------------------------------------------------
constexpr int ACT_NB = 8;          // time buckets of (threshold + 1) us a packet may span on the parallel path
constexpr int ACT_GROUP = 256;     // events per step of the sequential path
struct ActDev {
  long long* last_ts;     // [cam_px] maximum time stamp of the pixel's events of EARLIER packets (ING_NO_TS: none)
  uint2* cells;           // [ACT_NB][cam_px] per (bucket, pixel) of the current packet: .x = ~(smallest packet index), .y = largest
                          // (stamp - bucket start) + 1; 0 = no event.  All zero between packets (k_ing_append / k_act_update clear)
  unsigned char* keep;    // [max_packet] the packet's keep flags
  u32* ctl;               // [0] != 0: the packet failed k_act_first's check (judged sequentially: by block 0 of k_act_mark, or block
                          // after block inside k_ing_count); [1]: tickets taken by k_ing_count's blocks of a sequential packet (their
                          // place in the chain); [2]: packets judged sequentially so far (statistics); [3]: k_ing_count's
                          // blocks that have done their part of a sequential packet
  long long thresh;       // T
  int cam_w, cam_h;
  int self_counts;        // != 0: an earlier event at the event's OWN pixel qualifies too (a 3 x 3 window that includes its centre: one of
                          // the ways Metavision's filter may differ -- oracle/ingest_oracle.py lists them; XM_INGEST_ACT_SELF)
};

struct IngestDev {        // by value to every ingest kernel
  IngestState* st;
  uint4* buf;             // cap + mirror records
  u64 cap, mirror;        // cap: power of two
  u64 room;               // events that must stay free behind the live part: (1 + packets the ingest stream may run ahead) x the largest packet
  u64* pring;             // pause ring, pcap entries (power of two)
  u64 pcap;
  IngBlk* blk;            // ING_MAX_BLOCKS records of the current packet
  ActDev act;             // activity filter (act.last_ts == NULL: filter off)
  int cam_w, cam_h;
  long long pause_thresh;
  double period;
  u32 min_events, ring;   // ring: entries of the pinned result ring (status slot = frame number % ring)
  u32 nout, pad0;         // device-side output buffers K2 writes (frame number % nout); copied to the result ring by DMA
  FrameDesc* desc;        // this packet's entries of the descriptor / frame-info / verdict rings
  IngFrameInfo* info;
  IngVerdict* verdict;    // (pinned host memory)
  u64* key_frame;
  SlotState* slot;
  float* const* depth_ring;
  uint8_t* const* bgr_ring;
};

struct IngestPush {       // one packet
  const uint4* src;       // its records (device memory)
  const u32* n_dev;       // non-NULL: the packet's event count lives on the device (a chunk decoded there)
  u32 n;                  // ... else this many (with n_dev: the room of the packet's slot)
  u32 flags;              // ING_F_*
  u64 push_no;            // number of the packet (from 1)
};
constexpr u32 ING_F_POLARITY = 1u;  // keep p == 1 only
constexpr u32 ING_F_SEGMENT = 2u;   // k_ing_segment: run the trigger finder (else: commit the packet only)

constexpr long long ING_NO_TS = (long long)0x8000000000000000ull;

__device__ inline long long rec_t(const uint4& r) { return (long long)(((u64)r.w << 32) | r.z); }
// (a count on the device: n is the room of the packet's slot -- a chunk that decoded to more has been truncated to that)
__device__ inline u32 ing_packet_n(const IngestPush& p) {
  if (!p.n_dev) return p.n;
  const u32 m = *p.n_dev;
  return m < p.n ? m : p.n;
}

// ---- activity filter ----------------------------------------------------------------------------------------------------------
struct ActEv {            // what the rule needs of one event
  int x, y;
  long long t;
  bool part;              // takes part: passes the polarity rule and lies inside the sensor
};
__device__ inline ActEv act_event(const uint4& r, bool valid, bool use_pol, int cam_w, int cam_h) {
  ActEv e;
  e.x = (int)(r.x & 0xffff);
  e.y = (int)(r.x >> 16);
  e.t = rec_t(r);
  e.part = valid && (!use_pol || (short)(r.y & 0xffff) == 1) && e.x < cam_w && e.y < cam_h;
  return e;
}
// bucket of a stamp, counted from the packet's first stamp t0 in steps of W = T + 1: false if it lies in front of t0 or behind
// the ACT_NB-th bucket (the subtraction wraps for stamps 2^63 apart: such a packet takes the sequential path, too)
__device__ inline bool act_bucket(long long t, long long t0, long long W, int& b, u32& trel) {
  if (t < t0) return false;
  u64 d = (u64)t - (u64)t0;
  const u64 w = (u64)W;
  if (d >= w * ACT_NB) return false;
  int k = 0;
#pragma unroll
  for (int s = ACT_NB / 2; s > 0; s >>= 1)
    if (d >= w * (u64)s) {
      d -= w * (u64)s;
      k += s;
    }
  b = k;
  trel = (u32)d;
  return true;
}

// the rule against the running per-pixel maximum alone (sequential path; `last_ts` is read past the caches: the block that
// runs this has just updated it)
__device__ inline bool act_seen_recently(const ActDev& a, const ActEv& e) {
  bool act = false;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      if (dx == 0 && dy == 0 && !a.self_counts) continue;
      const int xx = e.x + dx, yy = e.y + dy;
      if (xx < 0 || xx >= a.cam_w || yy < 0 || yy >= a.cam_h) continue;
      const long long lt = __hip_atomic_load(&a.last_ts[(u32)yy * (u32)a.cam_w + (u32)xx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      act = act || (lt != ING_NO_TS && e.t - lt <= a.thresh);
    }
  return act;
}

// Events [begin, n) of the packet in stream order, 256 at a time, by ONE block: inside a group an event looks at the group's earlier
// events directly (LDS), at everything before the group through last_ts, which the group's events then join.  Leaves last_ts as
// k_ing_append / k_act_update will leave it anyway (a maximum: idempotent).
__device__ inline void act_sequential(const ActDev& a, const uint4* __restrict__ src, u32 begin, u32 n, bool use_pol) {
  __shared__ int s_x[ACT_GROUP], s_y[ACT_GROUP];
  __shared__ long long s_t[ACT_GROUP];
  __shared__ unsigned char s_part[ACT_GROUP];
  const u32 tid = threadIdx.x;
  for (u32 base = begin; base < n; base += ACT_GROUP) {
    const u32 i = base + tid;
    const bool valid = i < n;
    const ActEv e = act_event(valid ? src[i] : make_uint4(0, 0, 0, 0), valid, use_pol, a.cam_w, a.cam_h);
    s_x[tid] = e.x;
    s_y[tid] = e.y;
    s_t[tid] = e.t;
    s_part[tid] = e.part ? 1 : 0;
    __syncthreads();
    bool act = false;
    if (e.part) {
      act = act_seen_recently(a, e);
      for (u32 j = 0; j < tid && !act; ++j) {
        const int dx = s_x[j] - e.x, dy = s_y[j] - e.y;
        act = s_part[j] && dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1 && ((dx | dy) != 0 || a.self_counts) && e.t - s_t[j] <= a.thresh;
      }
    }
    if (valid) a.keep[i] = (e.part && act) ? 1 : 0;
    __syncthreads();
    if (e.part) __hip_atomic_fetch_max(&a.last_ts[(u32)e.y * (u32)a.cam_w + (u32)e.x], e.t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    __syncthreads();
  }
}

// Pass 1 of a packet: the (bucket, pixel) cells and the check that the buckets run forwards.  (A launch of its own, k_act_first, or --
// the ingest, when the packet is on the device early enough -- the tail blocks of the packet before's k_ing_count launch: k_ing_count_act.)  One event per thread: a quarter-period packet of the reference's rig
// (~42 k events) is 165 blocks -- the work is a divergent atomic or two per event, which wants the whole chip.
__device__ __forceinline__ void act_first_body(const ActDev& a, const uint4* __restrict__ src, const u32* __restrict__ n_dev, u32 n_room, int use_pol,
                                               const u32 blk) {
  u32 n = n_room;
  if (n_dev) {
    const u32 m = *n_dev;
    n = m < n_room ? m : n_room;
  }
  const u32 tid = threadIdx.x;
  const u32 i = blk * ING_THREADS + tid;
  bool bad = false;
  if (i < n) {
    const long long t0 = rec_t(src[0]), W = a.thresh + 1;
    const ActEv e = act_event(src[i], true, use_pol != 0, a.cam_w, a.cam_h);
    int b = 0, bp = 0, bl = 0;
    u32 trel = 0, trp = 0;
    if (!act_bucket(e.t, t0, W, b, trel)) {
      bad = true;
    } else {
      if (i > 0 && (!act_bucket(rec_t(src[i - 1]), t0, W, bp, trp) || bp > b)) bad = true;
      // The largest stamp of a (bucket, pixel) is read by events of the NEXT bucket only: the packet's last bucket (that of its
      // last event, when the buckets run forwards -- and when they do not, the cells are not used) has no reader.  A camera's
      // packets are a fraction of a threshold long: one bucket, and this atomic is never issued.
      const bool last_bucket = act_bucket(rec_t(src[n - 1]), t0, W, bl, trp) && bl == b;
      if (e.part) {
        uint2* c = a.cells + (size_t)b * ((u32)a.cam_w * (u32)a.cam_h) + (u32)e.y * (u32)a.cam_w + (u32)e.x;
        __hip_atomic_fetch_max(&c->x, ~i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!last_bucket) __hip_atomic_fetch_max(&c->y, trel + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  // (no fence, no hand-off between blocks here: an agent-scope release / acquire is an L2 write-back + invalidate on this chip --
  //  measured as 20 us for this kernel with a "last block finishes the job" pattern against 5 without; the kernel boundary in
  //  front of k_act_mark publishes the cells and the flag for free)
  if (bad) __hip_atomic_store(&a.ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(ING_THREADS) void k_act_first(ActDev a, const uint4* __restrict__ src, const u32* __restrict__ n_dev, u32 n_room,
                                                           int use_pol) {
  act_first_body(a, src, n_dev, n_room, use_pol, blockIdx.x);
}

// Pass 2, per event (parallel path): is event i of the packet kept?  (cells complete: k_act_first has run)
__device__ inline bool act_keep(const ActDev& a, const ActEv& e, u32 i, long long t0) {
  if (!e.part) return false;
  int b = 0;
  u32 trel = 0;
  (void)act_bucket(e.t, t0, a.thresh + 1, b, trel);  // (true for every event of a packet on this path)
  const u32 cam_px = (u32)a.cam_w * (u32)a.cam_h;
  const uint2* cb = a.cells + (size_t)b * cam_px;
  bool act = false;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      if (dx == 0 && dy == 0 && !a.self_counts) continue;
      const int xx = e.x + dx, yy = e.y + dy;
      if (xx < 0 || xx >= a.cam_w || yy < 0 || yy >= a.cam_h) continue;
      const u32 q = (u32)yy * (u32)a.cam_w + (u32)xx;
      const u32 first_inv = cb[q].x;                 // same bucket: an earlier event at q
      act = act || (first_inv != 0u && ~first_inv < i);
      if (b > 0) act = act || (cb - cam_px)[q].y >= trel + 2u;  // the bucket before: W + trel - trel' <= T  <=>  trel' > trel
      const long long lt = a.last_ts[q];             // earlier packets
      act = act || (lt != ING_NO_TS && e.t - lt <= a.thresh);
    }
  return act;
}

// Behind the flags: the event joins its pixel's history and its cells of this packet are emptied for the next one.
__device__ inline void act_retire(const ActDev& a, const ActEv& e, long long t0) {
  if (!e.part) return;
  const u32 pix = (u32)e.y * (u32)a.cam_w + (u32)e.x;
  __hip_atomic_fetch_max(&a.last_ts[pix], e.t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int b = 0;
  u32 trel = 0;
  if (act_bucket(e.t, t0, a.thresh + 1, b, trel)) a.cells[(size_t)b * ((u32)a.cam_w * (u32)a.cam_h) + pix] = make_uint2(0u, 0u);
}

// Pass 2: the flags, one event per thread (in front of k_ing_count, which reads them; or the filter alone, xm_activity_*: the
// host-side pipe's stand-in for ActivityNoiseFilterAlgorithm.process_events) ...
__global__ __launch_bounds__(ING_THREADS) void k_act_mark(ActDev a, const uint4* __restrict__ src, const u32* __restrict__ n_dev, u32 n_room,
                                                          int use_pol) {
  u32 n = n_room;
  if (n_dev) {
    const u32 m = *n_dev;
    n = m < n_room ? m : n_room;
  }
  if (a.ctl[0]) {  // the packet failed k_act_first's check: block 0 judges all of it sequentially, the others have nothing to do
    if (blockIdx.x == 0) {
      if (threadIdx.x == 0) a.ctl[2] += 1u;
      act_sequential(a, src, 0, n, use_pol != 0);
    }
    return;
  }
  const u32 i = blockIdx.x * ING_THREADS + threadIdx.x;
  if (i >= n) return;
  const ActEv e = act_event(src[i], true, use_pol != 0, a.cam_w, a.cam_h);
  a.keep[i] = act_keep(a, e, i, rec_t(src[0])) ? 1 : 0;
}
// ... then history + clean cells
__global__ __launch_bounds__(ING_THREADS) void k_act_update(ActDev a, const uint4* __restrict__ src, u32 n, int use_pol) {
  const u32 i = blockIdx.x * ING_THREADS + threadIdx.x;
  if (i >= n) return;
  if (i == 0) a.ctl[0] = 0u;  // (read by k_act_mark, in front of this kernel; the next packet's k_act_first sets it again)
  act_retire(a, act_event(src[i], true, use_pol != 0, a.cam_w, a.cam_h), rec_t(src[0]));
}

// ---- the block-local part shared by k_ing_count and k_ing_append -----------------------------------------------------------
// Thread `tid` holds events i = block * 512 + j * 256 + tid, j = 0, 1 (coalesced 16-byte loads; until round 5 a block took 2048
// events: a quarter-period packet of the reference's rig was 21 blocks on a 256-CU chip, k_ing_count 8.6 us and k_ing_append
// 16.9 us; with 83 blocks 5.7 and 8.9).  Kept events are ranked in
// stream order inside the block (ballots per slab of 256, a prefix over the (slab, wave) pairs); their time stamps go to LDS in
// rank order, where every kept event but the block's first finds its predecessor.
struct IngLocal {
  uint4 r[ING_EPT];
  u32 rank[ING_EPT];     // rank among the block's kept events (valid where the keep bit is set)
  u32 keep_bits;         // bit j: event j of this thread is kept
  u32 pause_bits;        // bit j: a pause lies between event j and the kept event before it (never set for rank 0)
  u32 n_kept;            // block totals
  u32 n_pauses;
};

struct IngShared {
  long long t[ING_EPB];
  u32 cnt[ING_EPT][ING_THREADS / 64];
  u32 off[ING_EPT][ING_THREADS / 64];
  u32 total, ptotal;
};

// MARK (k_ing_count with the activity filter on): the flags are computed here and left in act.keep, where k_ing_append
// (MARK = false) reads them.  A packet that failed k_act_first's check is judged in stream order: block b waits for block b - 1
// (`block` is a ticket taken when the block started -- ing_count_body --, so the one waited for is always running), then takes its 512 events through act_sequential
// -- a chain over the blocks, slow and exact, fences only on this path.
template <bool MARK>
__device__ inline void ing_block_local(const IngestPush& p, const ActDev& act, u32 n, u32 block, long long thresh, IngShared& s, IngLocal& L) {
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u64 lt_mask = (1ull << lane) - 1ull;
  const bool use_pol = (p.flags & ING_F_POLARITY) != 0;
  const bool act_on = act.last_ts != nullptr;
  bool compute = false;
  long long t0 = 0;
  if (MARK && act_on) {
    if (act.ctl[0]) {
      if (tid == 0) {
        while (__hip_atomic_load(&act.ctl[3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != block) __builtin_amdgcn_s_sleep(8);
        if (block == 0) act.ctl[2] += 1u;
      }
      __syncthreads();
      const u32 end = (block + 1) * ING_EPB < n ? (block + 1) * ING_EPB : n;
      act_sequential(act, p.src, block * ING_EPB, end, use_pol);
      __threadfence();
      __syncthreads();
      if (tid == 0) __hip_atomic_store(&act.ctl[3], block + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      compute = true;
      t0 = rec_t(p.src[0]);
    }
  }
  u32 before[ING_EPT];
  L.keep_bits = 0;
#pragma unroll
  for (int j = 0; j < ING_EPT; ++j) {
    const u32 i = block * ING_EPB + j * ING_THREADS + tid;
    const bool valid = i < n;
    L.r[j] = valid ? p.src[i] : make_uint4(0, 0, 0, 0);
    bool k;
    if (!act_on) {
      k = valid && (!use_pol || (short)(L.r[j].y & 0xffff) == 1);
    } else if (MARK && compute) {
      k = valid && act_keep(act, act_event(L.r[j], valid, use_pol, act.cam_w, act.cam_h), i, t0);
      if (valid) act.keep[i] = k ? 1 : 0;
    } else {
      k = valid && act.keep[i] != 0;
    }
    const u64 b = __ballot(k);
    before[j] = __popcll(b & lt_mask);
    if (lane == 0) s.cnt[j][wave] = __popcll(b);
    L.keep_bits |= k ? (1u << j) : 0u;
  }
  __syncthreads();
  if (tid < ING_EPT * (ING_THREADS / 64)) {  // 32 entries in (slab, wave) order = stream order: exclusive prefix by one wave
    const u32 v = (&s.cnt[0][0])[tid];
    u32 incl = v;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      const u32 up = __shfl_up(incl, o, 64);
      if ((int)tid >= o) incl += up;
    }
    (&s.off[0][0])[tid] = incl - v;
    if (tid == ING_EPT * (ING_THREADS / 64) - 1) s.total = incl;
  }
  __syncthreads();
  L.n_kept = s.total;
#pragma unroll
  for (int j = 0; j < ING_EPT; ++j) {
    L.rank[j] = s.off[j][wave] + before[j];
    if (L.keep_bits & (1u << j)) s.t[L.rank[j]] = rec_t(L.r[j]);
  }
  __syncthreads();
  L.pause_bits = 0;
  u32 np = 0;
#pragma unroll
  for (int j = 0; j < ING_EPT; ++j) {
    if ((L.keep_bits & (1u << j)) && L.rank[j] > 0 && rec_t(L.r[j]) - s.t[L.rank[j] - 1] >= thresh) {
      L.pause_bits |= 1u << j;
      np += 1;
    }
  }
  // block total of the pauses (their order inside the block follows from the ranks: see ing_pause_rank)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) np += __shfl_xor(np, o, 64);
  __syncthreads();  // (s.cnt is reused)
  if (lane == 0) s.cnt[0][wave] = np;
  __syncthreads();
  L.n_pauses = s.cnt[0][0] + s.cnt[0][1] + s.cnt[0][2] + s.cnt[0][3];
}

__device__ __forceinline__ void ing_count_body(const IngestDev& d, const IngestPush& p, u32 blk) {
  __shared__ IngShared s;
  const u32 n = ing_packet_n(p);
  const u32 nb = (n + ING_EPB - 1) / ING_EPB;
  if (blk >= nb) return;
  if (d.act.last_ts && d.act.ctl[0]) {
    // a packet that is judged sequentially (ing_block_local): the blocks take their place in the chain by TICKET, in the order
    // in which they start -- whatever order the hardware dispatches them in, the block a block waits for is already running
    __shared__ u32 s_ticket;
    if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(&d.act.ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    blk = s_ticket;  // (nb blocks get here: the tickets are a permutation of 0 .. nb - 1)
  }
  IngLocal L;
  ing_block_local<true>(p, d.act, n, blk, d.pause_thresh, s, L);
  if (threadIdx.x == 0) {
    IngBlk o;
    o.kept = L.n_kept;
    o.pauses = L.n_pauses;
    o.first_t = L.n_kept ? s.t[0] : 0;
    o.last_t = L.n_kept ? s.t[L.n_kept - 1] : 0;
    o.pad = 0;
    d.blk[blk] = o;
  }
}

__global__ __launch_bounds__(ING_THREADS) void k_ing_count(IngestDev d, IngestPush p) { ing_count_body(d, p, blockIdx.x); }

// k_ing_count of packet k and, in the same launch, the activity filter's first pass of packet k + 1 (already on the device: a
// replay, or a camera that is ahead of the GPU): blocks [0, nb_count) count, the blocks behind them fill packet k + 1's cells --
// the OTHER set of cells (xm_ingest: two sets, taken in turns), emptied by k_ing_append of packet k - 1, which ran in front of
// this launch.  The two halves touch nothing in common; the first pass then costs no link in the stream's chain of launches.
__global__ __launch_bounds__(ING_THREADS) void k_ing_count_act(IngestDev d, IngestPush p, u32 nb_count, ActDev a2, const uint4* __restrict__ src2,
                                                               const u32* __restrict__ n_dev2, u32 n_room2, int use_pol2) {
  if (blockIdx.x < nb_count) ing_count_body(d, p, blockIdx.x);
  else act_first_body(a2, src2, n_dev2, n_room2, use_pol2, blockIdx.x - nb_count);
}

// The packet's block records -> what lies in front of block `b` (kept events, pauses) and the packet's totals.  A pause in
// front of a block's FIRST kept event is decided here: against the last kept event of the nearest earlier block that kept
// anything, or against the stream's last event (tail_t) when there is none.  Every thread of the block takes part; nb <= 4096.
struct IngScan {
  u32 kept_before, pauses_before;  // of block b
  u32 kept_total, pauses_total;
  long long prev_t;                // time stamp in front of block b's first kept event
  bool has_prev, boundary_pause;   // ... whether there is one, and whether a pause lies between the two
};

struct IngScanShared {
  u32 kept[ING_MAX_BLOCKS];
  long long last_t[ING_MAX_BLOCKS];
  int group_last[ING_MAX_BLOCKS / 64];
  u32 red[4][ING_THREADS / 64];
  int pred_b, bp_b;
};

__device__ inline IngScan ing_scan_blocks(const IngBlk* __restrict__ blk, u32 nb, u32 b, bool has_tail, long long tail_t,
                                          long long thresh, IngScanShared& s) {
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u64 lt_mask = (1ull << lane) - 1ull;
  // pass 1: kept / last_t to LDS, the last block of every group of 64 that kept anything
  for (u32 base = 0; base < nb; base += ING_THREADS) {
    const u32 j = base + tid;
    const u32 k = j < nb ? blk[j].kept : 0u;
    if (j < nb) {
      s.kept[j] = k;
      s.last_t[j] = blk[j].last_t;
    }
    const u64 bal = __ballot(k != 0);
    if (lane == 0) s.group_last[(base >> 6) + wave] = bal ? (int)(base + wave * 64 + 63 - __builtin_clzll(bal)) : -1;
  }
  __syncthreads();
  // pass 2: per block j its predecessor, the boundary pause, the sums
  u32 kb = 0, pb = 0, kt = 0, pt = 0;
  for (u32 base = 0; base < nb; base += ING_THREADS) {
    const u32 j = base + tid;
    const u32 k = j < nb ? s.kept[j] : 0u;
    const u64 bal = __ballot(k != 0);
    if (j < nb) {
      int pred = -1;
      const u64 below = bal & lt_mask;
      if (below) pred = (int)(base + wave * 64 + 63 - __builtin_clzll(below));
      else
        for (int g = (int)(j >> 6) - 1; g >= 0 && pred < 0; --g) pred = s.group_last[g];
      bool bp = false;
      if (k) {
        const long long ft = blk[j].first_t;
        bp = pred >= 0 ? ft - s.last_t[pred] >= thresh : (has_tail && ft - tail_t >= thresh);
      }
      const u32 add_p = blk[j].pauses + (bp ? 1u : 0u);
      if (j == b) {
        s.pred_b = pred;
        s.bp_b = bp ? 1 : 0;
      }
      if (j < b) {
        kb += k;
        pb += add_p;
      }
      kt += k;
      pt += add_p;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kb += __shfl_xor(kb, o, 64);
    pb += __shfl_xor(pb, o, 64);
    kt += __shfl_xor(kt, o, 64);
    pt += __shfl_xor(pt, o, 64);
  }
  if (lane == 0) {
    s.red[0][wave] = kb;
    s.red[1][wave] = pb;
    s.red[2][wave] = kt;
    s.red[3][wave] = pt;
  }
  __syncthreads();
  IngScan o;
  o.kept_before = s.red[0][0] + s.red[0][1] + s.red[0][2] + s.red[0][3];
  o.pauses_before = s.red[1][0] + s.red[1][1] + s.red[1][2] + s.red[1][3];
  o.kept_total = s.red[2][0] + s.red[2][1] + s.red[2][2] + s.red[2][3];
  o.pauses_total = s.red[3][0] + s.red[3][1] + s.red[3][2] + s.red[3][3];
  o.has_prev = false;
  o.boundary_pause = false;
  o.prev_t = 0;
  if (b < nb) {
    const int pred = s.pred_b;
    o.boundary_pause = s.bp_b != 0;
    o.has_prev = pred >= 0 || has_tail;
    o.prev_t = pred >= 0 ? s.last_t[pred] : tail_t;
  }
  return o;
}

// kept events -> the ring, pauses -> the pause ring; every positive event joins its pixel's history (activity filter)
__global__ __launch_bounds__(ING_THREADS) void k_ing_append(IngestDev d, IngestPush p) {
  __shared__ IngShared s;
  __shared__ IngScanShared ss;
  __shared__ u32 s_prank[ING_EPT][ING_THREADS / 64];
  const u32 n = ing_packet_n(p);
  const u32 nb = (n + ING_EPB - 1) / ING_EPB;
  if (blockIdx.x >= nb) return;
  const IngestState st = *d.st;  // (not modified by this kernel: k_ing_segment commits)
  const IngScan sc = ing_scan_blocks(d.blk, nb, blockIdx.x, st.write_abs > 0, st.last_t, d.pause_thresh, ss);
  IngLocal L;
  ing_block_local<false>(p, d.act, n, blockIdx.x, d.pause_thresh, s, L);
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u64 lt_mask = (1ull << lane) - 1ull;
  // the block's first kept event: the pause in front of it was decided by the scan
  u32 pb = L.pause_bits;
#pragma unroll
  for (int j = 0; j < ING_EPT; ++j)
    if ((L.keep_bits & (1u << j)) && L.rank[j] == 0 && sc.boundary_pause) pb |= 1u << j;
  // order of the pauses inside the block = order of their events: ballots per slab again
  u32 pbefore[ING_EPT];
#pragma unroll
  for (int j = 0; j < ING_EPT; ++j) {
    const u64 b = __ballot((pb >> j) & 1u);
    pbefore[j] = __popcll(b & lt_mask);
    if (lane == 0) s_prank[j][wave] = __popcll(b);
  }
  __syncthreads();
  if (tid < ING_EPT * (ING_THREADS / 64)) {
    const u32 v = (&s_prank[0][0])[tid];
    u32 incl = v;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      const u32 up = __shfl_up(incl, o, 64);
      if ((int)tid >= o) incl += up;
    }
    (&s_prank[0][0])[tid] = incl - v;
  }
  __syncthreads();
  const u64 limit = st.start_abs + d.cap;  // the ring is full beyond: such events are dropped (k_ing_segment counts them)
  const long long act_t0 = d.act.last_ts ? rec_t(p.src[0]) : 0;
  const u64 base = st.write_abs + sc.kept_before;
  const u64 pbase = st.p_tail + sc.pauses_before;
#pragma unroll
  for (int j = 0; j < ING_EPT; ++j) {
    const u32 i = blockIdx.x * ING_EPB + j * ING_THREADS + tid;
    if (i >= n) continue;
    const uint4 r = L.r[j];
    if (d.act.last_ts) act_retire(d.act, act_event(r, true, (p.flags & ING_F_POLARITY) != 0, d.act.cam_w, d.act.cam_h), act_t0);
    if (!(L.keep_bits & (1u << j))) continue;
    const u64 abs = base + L.rank[j];
    if (abs >= limit) continue;
    uint4 o = r;
    o.y = (o.y & 0xffff0000u) | 1u;  // p = 1 (like the filters' outputs)
    const u64 pos = abs & (d.cap - 1);
    d.buf[pos] = o;
    if (pos < d.mirror) d.buf[d.cap + pos] = o;
    if ((pb >> j) & 1u) d.pring[(pbase + s_prank[j][wave] + pbefore[j]) & (d.pcap - 1)] = abs - 1;  // t[abs] - t[abs - 1] >= thresh
  }
}

// ---- segmentation -------------------------------------------------------------------------------------------------------
// The decision of find_trigger (trigger_finder.py:137-189) over the live pauses; every thread of the block takes part.
__device__ inline void ing_find_trigger(const IngestDev& d, u64& s_first) {
  IngestState* st = d.st;
  const u32 tid = threadIdx.x;
  const u64 start = st->start_abs, write = st->write_abs;
  if (write == start) return;
  const u64 mask = d.cap - 1;
  {
    const long long span = rec_t(d.buf[(write - 1) & mask]) - rec_t(d.buf[start & mask]);
    if ((double)span < d.period) return;  // fewer than one period buffered: wait for more (trigger_finder.py:137-139)
  }
  // pauses in front of the live part have left the buffer with their events (the list is ascending: skip its stale head)
  u64 ph = st->p_head;
  const u64 pt = st->p_tail;
  for (;;) {  // (block-uniform loop)
    const u64 k = ph + tid;
    const bool stale = k < pt && d.pring[k & (d.pcap - 1)] < start;
    const int cnt = __syncthreads_count(stale);
    ph += (u64)cnt;
    if (cnt < ING_THREADS) break;
  }
  // the first pair of consecutive pauses more than half a period apart
  for (u64 k0 = ph; k0 + 1 < pt; k0 += ING_THREADS) {
    const u64 k = k0 + tid;
    if (k + 1 < pt) {
      const long long gap = rec_t(d.buf[d.pring[(k + 1) & (d.pcap - 1)] & mask]) - rec_t(d.buf[d.pring[k & (d.pcap - 1)] & mask]);
      if ((double)gap > d.period / 2) atomicMin((unsigned long long*)&s_first, (unsigned long long)k);
    }
    __syncthreads();
    const u64 found = s_first;
    __syncthreads();  // (nobody moves on to the next chunk's atomicMin before everybody has read this chunk's verdict)
    if (found != ~0ull) break;
  }
  if (tid != 0) return;
  st->span_ok = 1;
  const u64 k = s_first;
  if (k == ~0ull) {  // no plausible pair: the reference has popped the whole buffer and pushes nothing back
    st->start_abs = write;
    st->p_head = pt;
    return;
  }
  const u64 prev = d.pring[k & (d.pcap - 1)], next = d.pring[(k + 1) & (d.pcap - 1)];
  const long long gap = rec_t(d.buf[next & mask]) - rec_t(d.buf[prev & mask]);
  st->p_head = k + 1;  // (entries below `next` are stale from here on; the exact head is found again next time)
  if ((double)gap <= d.period && next - prev > d.min_events) {
    const u64 first = prev + 2, last = next - 2;  // evs[prev + 2 : next - 2]
    if (last - first <= d.mirror) {
      const u64 frame_no = st->frames++;
      const u32 slot_i = (u32)(frame_no % d.ring);
      d.info->frame_no = frame_no;
      d.info->t_first = rec_t(d.buf[first & mask]);
      d.info->t_last = rec_t(d.buf[(last - 1) & mask]);
      FrameDesc* desc = d.desc;
      desc->x = nullptr; desc->y = nullptr; desc->t = nullptr; desc->p = nullptr;
      desc->aos = d.buf + (first & mask);  // contiguous: the ring's head is mirrored behind its end
      desc->n = last - first;
      desc->key_frame = d.key_frame;
      desc->st = d.slot;
      desc->depth = d.depth_ring ? d.depth_ring[frame_no % d.nout] : nullptr;
      desc->bgr = d.bgr_ring ? d.bgr_ring[frame_no % d.nout] : nullptr;
      desc->pad = slot_i;
      desc->valid = 1;
    } else {
      st->overflow += (u32)(last - first);  // a frame longer than the mirrored part cannot be handed out in one piece
    }
    st->start_abs = last;  // push(evs[next - 2 :])
  } else {
    st->start_abs = next;  // "trigger not found correctly, drop these events": push(evs[next:])
  }
}

// Commits the packet (write cursor, pause ring, overflow), then -- ING_F_SEGMENT -- the trigger finder.  One block.
__global__ __launch_bounds__(ING_THREADS) void k_ing_segment(IngestDev d, IngestPush p) {
  __shared__ IngScanShared ss;
  __shared__ u64 s_first;
  IngestState* st = d.st;
  const u32 n = ing_packet_n(p);
  const u32 nb = (n + ING_EPB - 1) / ING_EPB;
  const u32 tid = threadIdx.x;
  {
    const IngestState cur = *st;
    const IngScan sc = ing_scan_blocks(d.blk, nb, nb, cur.write_abs > 0, cur.last_t, d.pause_thresh, ss);
    __syncthreads();
    if (tid == 0) {
      const u64 limit = cur.start_abs + d.cap;
      u64 w = cur.write_abs + sc.kept_total, pt = cur.p_tail + sc.pauses_total;
      if (w > limit) {  // ring full: the packet's tail was not stored; its pauses go too
        st->overflow += (u32)(w - limit);
        w = limit;
        while (pt > cur.p_tail && d.pring[(pt - 1) & (d.pcap - 1)] + 1 >= w) --pt;
      }
      if (p.n_dev && *p.n_dev > p.n) st->overflow += *p.n_dev - p.n;  // (a chunk decoded on the device that did not fit its slot)
      st->appended += w - cur.write_abs;
      st->write_abs = w;
      st->p_tail = pt;
      if (w > 0) st->last_t = rec_t(d.buf[(w - 1) & (d.cap - 1)]);
      if (p.flags & ING_F_SEGMENT) {
        d.desc->valid = 0;
        st->span_ok = 0;
      }
      if (d.act.last_ts) {  // (the packet's flags have been consumed: the next packet on this set of cells decides afresh)
        d.act.ctl[0] = 0u;
        d.act.ctl[1] = 0u;
        d.act.ctl[3] = 0u;
      }
      s_first = ~0ull;
    }
    __syncthreads();
  }
  if (!(p.flags & ING_F_SEGMENT)) return;
  ing_find_trigger(d, s_first);
  __syncthreads();
  // No room for the packets that may follow before the next decision (d.room) behind what is still live -- a stream without a
  // usable pause: the reference's buffer would grow without bound -- : the live part is dropped and counted, as if the trigger
  // finder had given up on it.
  if (tid != 0) return;
  if (st->write_abs - st->start_abs + d.room > d.cap) {
    st->overflow += (u32)(st->write_abs - st->start_abs);
    st->start_abs = st->write_abs;
    st->p_head = st->p_tail;
  }
  d.info->live_after = st->write_abs - st->start_abs;
  d.info->overflow = st->overflow;
  // the verdict for the host: did this packet cut a frame, and of how many events (the frame kernels' grids)
  const u64 info = d.desc->valid ? ((1ull << 63) | d.desc->n) : 0ull;
  __hip_atomic_store(&d.verdict->info, info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&d.verdict->push_no, p.push_no, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// after the frame kernels (same stream, before the next frame's kernels touch the slot's counters): the frame's statistics into
// its entry of the pinned status ring -- everything but the sequence number, which k_ing_publish_seq writes once the frame's
// outputs have arrived in host memory
__global__ __launch_bounds__(64) void k_ing_publish(IngestState* st, const FrameDesc* __restrict__ desc, const IngFrameInfo* __restrict__ info,
                                                    IngestStatus* ring_status, u64 push_seq) {
  if (!desc->valid) return;
  const SlotState* s = desc->st;
  const u32 tag = s->tag_a, parity = tag & 1;
  u64 inl = 0, oob = 0, used = 0;
  for (int i = threadIdx.x; i < CNT_SLOTS; i += 64) {
    inl += s->cnt[parity][i][CNT_INLIER];
    oob += s->cnt[parity][i][CNT_OOB];
    used += s->cnt[parity][i][CNT_USED];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    inl += __shfl_xor(inl, o, 64);
    oob += __shfl_xor(oob, o, 64);
    used += __shfl_xor(used, o, 64);
  }
  if (threadIdx.x != 0) return;
  IngestStatus* out = ring_status + desc->pad;
  // (a reader that the ring has lapped may be copying this entry right now: the sequence number goes first, so that its second
  //  look at it fails -- the new one follows only when the frame's outputs have arrived, a whole copy later)
  __hip_atomic_store(&out->seq, (u64)0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  out->n_events = desc->n;
  out->t_first = info->t_first;
  out->t_last = info->t_last;
  out->n_inliers = inl;
  out->n_index_errors = oob;
  out->n_used = used;
  out->live_after = info->live_after;
  out->push_seq = push_seq;
  out->overflow = info->overflow;
  __threadfence_system();
}

// behind the DMA copies of the frame's outputs (out stream): the sequence number, written last (system scope: the host polls it)
__global__ __launch_bounds__(64) void k_ing_publish_seq(IngestState* st, const FrameDesc* __restrict__ desc, IngestStatus* out, u64 frame_no) {
  if (threadIdx.x != 0 || !desc->valid) return;
  atomicMax((unsigned long long*)&st->published, (unsigned long long)(frame_no + 1));  // (two out streams: not always in frame order)
  __threadfence_system();
  __hip_atomic_store(&out->seq, frame_no + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace xm
