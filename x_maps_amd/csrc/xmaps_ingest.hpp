// xmaps_ingest.hpp -- device-side ingest ("next" row N2 of SURVEY.md section 8(f)): the raw camera stream stays in HBM.
//
// What the reference does on the host per packet (python/depth_reprojection_pipe.py:110-119, python/trigger_finder.py:128-189):
//   PolarityFilterAlgorithm(1)  ->  ActivityNoiseFilterAlgorithm(w, h, int(1e6 / fps))  ->  RobustTriggerFinder:
//   buffer packets until they span one projector period; pauses = nonzero(diff(t) >= 40 us); the first pair of consecutive
//   pauses more than half a period apart decides: at most one period apart and > 1000 events -> frame = evs[prev+2 : next-2],
//   keep evs[next-2:]; otherwise drop everything up to next; no such pair -> the whole buffer is dropped.
// Here the same chain runs as kernels over a device-resident event buffer; the frame that is cut is described by a FrameDesc
// written BY THE DEVICE (pointer into the buffer + count), which the multi-frame K0/K1/K2 launches read -- no index ever
// travels to the host, the host only enqueues a fixed sequence of launches per packet and later finds finished frames in a
// pinned ring.
//
// Activity-noise filter: Metavision's ActivityNoiseFilterAlgorithm is closed source, so its exact rule is unpinned
// (SURVEY.md 8(c)).  OWN DEFINITION, implemented identically in oracle/ingest_oracle.py:
//   a (positive) event e = (x, y, t) is KEPT iff some event e' EARLIER IN THE STREAM at one of the 8 neighbouring pixels has
//   t - t' <= T (T = int(1e6 / fps)); every positive event, kept or not, then becomes its pixel's latest event.
// Parallel evaluation, exact for any event order: the host splits packets into sub-packets whose time span (max - min) is
// <= T; inside a sub-packet ANY earlier event at a neighbour qualifies (its t' is within the span), found through a
// first-index map (atomicMin); events of earlier sub-packets qualify through the per-pixel maximum time stamp.
#pragma once
#include "xmaps_kernels.hpp"

namespace xm {

struct IngestState {      // device
  u64 buf_start;          // first live event of the current buffer
  u64 write;              // one past the last live event
  u32 cur;                // which of the two buffers is current (compaction copies the live part to the other one)
  u32 overflow;           // events dropped because the buffer was full (sticky)
  u64 frames;             // frames cut so far
  u64 appended;           // events appended so far (after the filters)
  u32 n_pauses;           // scratch: pauses found in the live part this round
  u32 decided;            // scratch
  u32 span_ok;            // scratch: the live part spans at least one period
  u32 pad;
};

struct IngestStatus {     // pinned host ring entry, written by k_ing_publish when a frame has been produced
  u64 seq;                // frame number + 1; written LAST (system-scope release)
  u64 n_events;
  long long t_first, t_last;
  u64 n_inliers, n_index_errors, n_used;
  u64 live_after;         // events left in the buffer after the cut
  u64 push_seq;           // number of the xm_ingest_push call whose launches cut this frame (the host tightens its bounds with it)
  u32 overflow;
  u32 pad;
};

constexpr long long ING_NO_TS = (long long)0x8000000000000000ull;

__device__ inline long long rec_t(const uint4& r) { return (long long)(((u64)r.w << 32) | r.z); }

// ---- filters ----------------------------------------------------------------------------------------------------------
// first event index of the sub-packet per pixel (positive events only)
__global__ __launch_bounds__(BLOCK) void k_ing_first(const uint4* __restrict__ pkt, u32 m, int use_pol, int cam_w, int cam_h,
                                                     u32* __restrict__ first_idx) {
  const u32 i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= m) return;
  const uint4 r = pkt[i];
  if (use_pol && (short)(r.y & 0xffff) != 1) return;
  const u32 x = r.x & 0xffff, y = r.x >> 16;
  if (x >= (u32)cam_w || y >= (u32)cam_h) return;
  __hip_atomic_fetch_min(&first_idx[y * (u32)cam_w + x], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// keep flags: polarity, then the activity rule (see the header comment)
__global__ __launch_bounds__(BLOCK) void k_ing_mark(const uint4* __restrict__ pkt, u32 m, int use_pol, int activity,
                                                    long long thresh, int cam_w, int cam_h, const u32* __restrict__ first_idx,
                                                    const long long* __restrict__ last_ts, u32* __restrict__ keep) {
  const u32 i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= m) return;
  const uint4 r = pkt[i];
  bool k = !use_pol || (short)(r.y & 0xffff) == 1;
  if (k && activity) {
    const int x = (int)(r.x & 0xffff), y = (int)(r.x >> 16);
    const long long t = rec_t(r);
    bool act = false;
    if (x < cam_w && y < cam_h) {
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          if (dx == 0 && dy == 0) continue;
          const int xx = x + dx, yy = y + dy;
          if (xx < 0 || xx >= cam_w || yy < 0 || yy >= cam_h) continue;
          const u32 c = (u32)yy * (u32)cam_w + (u32)xx;
          const long long lt = last_ts[c];
          act = act || first_idx[c] < i || (lt != ING_NO_TS && t - lt <= thresh);
        }
    }
    k = act;
  }
  keep[i] = k ? 1u : 0u;
}

// compact the kept events behind the write cursor; every positive event becomes its pixel's latest event
__global__ __launch_bounds__(SCAN_BLOCK) void k_ing_append(const uint4* __restrict__ pkt, u32 m, int use_pol, int cam_w, int cam_h,
                                                          const u32* __restrict__ keep, const u32* __restrict__ pos,
                                                          const u32* __restrict__ sums, const u32* __restrict__ total,
                                                          IngestState* st, uint4* __restrict__ buf0, uint4* __restrict__ buf1,
                                                          u64 capacity, long long* __restrict__ last_ts) {
  const u32 i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  if (i >= m) return;
  const uint4 r = pkt[i];
  const bool positive = !use_pol || (short)(r.y & 0xffff) == 1;
  if (positive && last_ts) {
    const u32 x = r.x & 0xffff, y = r.x >> 16;
    if (x < (u32)cam_w && y < (u32)cam_h)
      __hip_atomic_fetch_max(&last_ts[y * (u32)cam_w + x], rec_t(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!keep[i]) return;
  const u64 w = st->write;  // not modified by this kernel (k_ing_commit advances it)
  if (w + *total > capacity) return;  // buffer full: the packet is dropped (k_ing_commit counts it)
  uint4* buf = st->cur ? buf1 : buf0;
  uint4 o = r;
  o.y = (o.y & 0xffff0000u) | 1u;  // p = 1 (like the filters' outputs)
  buf[w + sums[blockIdx.x] + pos[i]] = o;
}

__global__ void k_ing_commit(IngestState* st, const u32* __restrict__ total, u64 capacity) {
  if (threadIdx.x || blockIdx.x) return;
  const u64 n = *total;
  if (st->write + n > capacity) {
    st->overflow += (u32)n;
    return;
  }
  st->write += n;
  st->appended += n;
}

// make room: when the next packet might not fit behind the write cursor, the live part moves to the front of the OTHER buffer
__global__ __launch_bounds__(BLOCK) void k_ing_compact(IngestState* st, uint4* __restrict__ buf0, uint4* __restrict__ buf1,
                                                       u64 capacity, u64 incoming_max) {
  const u64 start = st->buf_start, write = st->write;  // read-only here (k_ing_compact_commit flips the buffers)
  if (write + incoming_max <= capacity) return;
  const uint4* src = st->cur ? buf1 : buf0;
  uint4* dst = st->cur ? buf0 : buf1;
  const u64 live = write - start, stride = (u64)gridDim.x * BLOCK;
  for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < live; i += stride) dst[i] = src[start + i];
}
__global__ void k_ing_compact_commit(IngestState* st, u64 capacity, u64 incoming_max) {
  if (threadIdx.x || blockIdx.x) return;
  if (st->write + incoming_max <= capacity) return;
  const u64 live = st->write - st->buf_start;
  st->cur ^= 1u;
  st->buf_start = 0;
  st->write = live;
  if (live + incoming_max > capacity) {  // even the live part alone leaves no room: drop it (counted)
    st->overflow += (u32)live;
    st->write = 0;
  }
}

// ---- segmentation -------------------------------------------------------------------------------------------------------
// does the live part span a period?  (trigger_finder.py:137-139)  Also clears the per-round scratch.
__global__ void k_ing_begin(IngestState* st, const uint4* __restrict__ buf0, const uint4* __restrict__ buf1, double period,
                            FrameDesc* desc) {
  if (threadIdx.x || blockIdx.x) return;
  st->n_pauses = 0;
  st->decided = 0;
  st->span_ok = 0;
  desc->valid = 0;
  if (st->write == st->buf_start) return;
  const uint4* buf = st->cur ? buf1 : buf0;
  const long long span = rec_t(buf[st->write - 1]) - rec_t(buf[st->buf_start]);
  st->span_ok = !((double)span < period);
}

// pauses of the live part: flags[i] = t[i+1] - t[i] >= thresh for live-relative i (grid covers the host's upper bound)
__global__ __launch_bounds__(SCAN_BLOCK) void k_ing_pause_flags(const IngestState* __restrict__ st, const uint4* __restrict__ buf0,
                                                               const uint4* __restrict__ buf1, long long thresh, u32 n_bound,
                                                               u32* __restrict__ flags) {
  const u32 i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  if (i >= n_bound) return;
  u32 f = 0;
  if (st->span_ok) {
    const u64 live = st->write - st->buf_start;
    if ((u64)i + 1 < live) {
      const uint4* buf = (st->cur ? buf1 : buf0) + st->buf_start;
      f = (rec_t(buf[i + 1]) - rec_t(buf[i])) >= thresh ? 1u : 0u;
    }
  }
  flags[i] = f;
}

// The decision of find_trigger (trigger_finder.py:146-189) over the compacted pause list (live-relative indices, ascending).
// One block: the first pair (k, k+1) whose pauses are more than half a period apart is found with a block-wide minimum,
// thread 0 then applies the rule and writes the frame's descriptor.
__global__ __launch_bounds__(BLOCK) void k_ing_segment(IngestState* st, const uint4* __restrict__ buf0, const uint4* __restrict__ buf1,
                                                       const u32* __restrict__ pauses, const u32* __restrict__ n_pauses_dev,
                                                       double period, u32 min_events, FrameDesc* desc, u64* key_frame,
                                                       SlotState* slot, float* const* depth_ring, uint8_t* const* bgr_ring,
                                                       u32 ring) {
  __shared__ u32 s_first;
  if (threadIdx.x == 0) s_first = 0xffffffffu;
  __syncthreads();
  if (!st->span_ok) return;  // fewer than one period buffered: wait for more (trigger_finder.py:137-139)
  const uint4* buf = (st->cur ? buf1 : buf0) + st->buf_start;
  const u32 np = *n_pauses_dev;
  for (u32 k = threadIdx.x; k + 1 < np; k += BLOCK) {
    const long long gap = rec_t(buf[pauses[k + 1]]) - rec_t(buf[pauses[k]]);
    if ((double)gap > period / 2) atomicMin(&s_first, k);
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const u32 k = s_first;
  if (k == 0xffffffffu) {  // no plausible pair: the reference has popped the whole buffer and pushes nothing back
    st->buf_start = st->write;
    return;
  }
  const u32 prev = pauses[k], next = pauses[k + 1];
  const long long gap = rec_t(buf[next]) - rec_t(buf[prev]);
  if ((double)gap <= period && next - prev > min_events) {
    const u64 first = (u64)prev + 2, last = (u64)next - 2;  // evs[prev + 2 : next - 2]
    const u32 slot_i = (u32)(st->frames % ring);
    desc->x = nullptr; desc->y = nullptr; desc->t = nullptr; desc->p = nullptr;
    desc->aos = buf + first;
    desc->n = last - first;
    desc->key_frame = key_frame;
    desc->st = slot;
    desc->depth = depth_ring ? depth_ring[slot_i] : nullptr;
    desc->bgr = bgr_ring ? bgr_ring[slot_i] : nullptr;
    desc->pad = slot_i;
    desc->valid = 1;
    st->buf_start += last;  // push(evs[next - 2 :])
    st->decided = 1;
  } else {
    st->buf_start += next;  // "trigger not found correctly, drop these events": push(evs[next:])
  }
}

// after the frame kernels: statistics + sequence number into the pinned ring (system scope: the host polls it)
__global__ __launch_bounds__(64) void k_ing_publish(IngestState* st, const FrameDesc* __restrict__ desc, IngestStatus* ring_status,
                                                    u64 push_seq) {
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
  out->n_events = desc->n;
  out->t_first = desc->n ? rec_t(desc->aos[0]) : 0;
  out->t_last = desc->n ? rec_t(desc->aos[desc->n - 1]) : 0;
  out->n_inliers = inl;
  out->n_index_errors = oob;
  out->n_used = used;
  out->live_after = st->write - st->buf_start;
  out->push_seq = push_seq;
  out->overflow = st->overflow;
  st->frames += 1;
  __threadfence_system();
  __hip_atomic_store(&out->seq, st->frames, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace xm
