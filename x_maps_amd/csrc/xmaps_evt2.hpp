// xmaps_evt2.hpp -- Prophesee EVT 2.0 words -> EventCD records on the device ("EVT2/EVT3 RAW reader"): the older
// of the two public RAW encodings, 32-bit little-endian words, one event per CD word.  (gfx950 / MI355X; included by xmaps_hip.hip
// after xmaps_evt3.hpp, whose state record and three-launch shape it shares.)  x_maps_amd/evt2.py restates the format on the host,
// oracle/evt2_oracle.py is the independent word-by-word checker.
//
//   [31:28] type   0x0 CD_OFF / 0x1 CD_ON   [27:22] t[5:0]   [21:11] x   [10:0] y          one event, p = type
//                  0x8 EVT_TIME_HIGH        [27:0]  t[33:6]                                the time base of the words behind it
//                  0xA EXT_TRIGGER, 0xE OTHERS, 0xF CONTINUED, anything else               skipped
//   t = (loops << 34) | (time_high << 6) | t[5:0]; a loop: the 28-bit field falls back by more than 2^27 (4.8 hours of stream).
//
// The state at word i is the last TIME_HIGH at or before it (+ the loops among the TIME_HIGH words so far) and the number of CD
// words before it: an inclusive scan with an associative combine, evaluated like EVT 3.0's -- block aggregates, their exclusive
// scan in one block seeded with the previous chunk's state (which also leaves the next state and the chunk's event count),
// every block re-scans its words from its prefix and writes its records in word order.
#pragma once
#include "xmaps_evt3.hpp"

namespace xm {

struct Evt2Scan {
  u32 hi_has, hi_first, hi_last, hi_wraps;  // TIME_HIGH words of the range
  u32 n_ev;
  u32 hi_word, n_pre;                       // a TIME_HIGH WORD lies in the range (the seed is not one); events in front of its first one
};
static_assert(sizeof(Evt2Scan) <= sizeof(Evt3Scan), "the decoders share the aggregates' buffer");

__device__ __forceinline__ Evt2Scan evt2_identity() {
  Evt2Scan e;
  e.hi_has = e.hi_first = e.hi_last = e.hi_wraps = 0;
  e.n_ev = 0;
  e.hi_word = e.n_pre = 0;
  return e;
}

// a = the earlier range, b = the later one
__device__ __forceinline__ Evt2Scan evt2_combine(const Evt2Scan& a, const Evt2Scan& b) {
  Evt2Scan r;
  if (!b.hi_has) {
    r.hi_has = a.hi_has; r.hi_first = a.hi_first; r.hi_last = a.hi_last; r.hi_wraps = a.hi_wraps;
  } else if (!a.hi_has) {
    r.hi_has = 1; r.hi_first = b.hi_first; r.hi_last = b.hi_last; r.hi_wraps = b.hi_wraps;
  } else {
    const bool wrap = (long long)a.hi_last - (long long)b.hi_first > (1ll << 27);
    r.hi_has = 1; r.hi_first = a.hi_first; r.hi_last = b.hi_last;
    r.hi_wraps = a.hi_wraps + b.hi_wraps + (wrap ? 1u : 0u);
  }
  r.n_ev = a.n_ev + b.n_ev;
  r.hi_word = a.hi_word | b.hi_word;
  r.n_pre = a.hi_word ? a.n_pre : a.n_ev + b.n_pre;
  return r;
}

__device__ __forceinline__ Evt2Scan evt2_element(u32 w) {
  Evt2Scan e = evt2_identity();
  const u32 typ = w >> 28;
  if (typ <= 1u) e.n_ev = e.n_pre = 1;
  else if (typ == 0x8u) { e.hi_has = 1; e.hi_first = e.hi_last = w & 0x0fffffffu; e.hi_word = 1; }
  return e;
}

__device__ __forceinline__ Evt2Scan evt2_seed(const Evt3State& s) {  // the state in front of the chunk as a range of its own
  Evt2Scan e = evt2_identity();
  e.hi_has = 1;
  e.hi_first = e.hi_last = s.t_high;
  return e;
}

// inclusive scan of one element per thread over the block; returns the thread's inclusive result, *block_total = the aggregate
__device__ __forceinline__ Evt2Scan evt2_block_scan(const Evt2Scan mine, Evt2Scan (*buf)[EVT3_THREADS], Evt2Scan* block_total) {
  const int tid = threadIdx.x;
  int cur = 0;
  buf[0][tid] = mine;
  __syncthreads();
  for (int o = 1; o < EVT3_THREADS; o <<= 1) {
    Evt2Scan v = buf[cur][tid];
    if (tid >= o) v = evt2_combine(buf[cur][tid - o], v);
    buf[cur ^ 1][tid] = v;
    cur ^= 1;
    __syncthreads();
  }
  const Evt2Scan r = buf[cur][tid];
  *block_total = buf[cur][EVT3_THREADS - 1];
  __syncthreads();
  return r;
}

// 1. the aggregate of every block of EVT3_PER_BLOCK words
__global__ __launch_bounds__(EVT3_THREADS) void k_evt2_aggregate(const u32* __restrict__ words, u32 n, Evt2Scan* __restrict__ agg) {
  __shared__ Evt2Scan buf[2][EVT3_THREADS];
  const u32 i0 = blockIdx.x * EVT3_PER_BLOCK + threadIdx.x * EVT3_IPT;
  Evt2Scan acc = evt2_identity();
#pragma unroll
  for (int k = 0; k < EVT3_IPT; ++k)
    if (i0 + k < n) acc = evt2_combine(acc, evt2_element(words[i0 + k]));
  Evt2Scan total;
  (void)evt2_block_scan(acc, buf, &total);
  if (threadIdx.x == 0) agg[blockIdx.x] = total;
}

// 2. one block: exclusive scan of the aggregates, seeded with the previous chunk's state; the chunk's event count and the state
//    for the next chunk (the EVT 3.0 record: only t_high, t_loops and n_events are used)
__global__ __launch_bounds__(EVT3_THREADS) void k_evt2_prefix(u32 n_blocks, Evt2Scan* __restrict__ agg, const Evt3State* __restrict__ st_in,
                                                             Evt3State* __restrict__ st_out, u32* __restrict__ count_out, int wait) {
  __shared__ Evt2Scan buf[2][EVT3_THREADS];
  __shared__ Evt2Scan s_incl[EVT3_THREADS];
  const Evt3State s = *st_in;
  Evt2Scan carry = evt2_seed(s);
  for (u32 b0 = 0; b0 < n_blocks; b0 += EVT3_THREADS) {
    const u32 b = b0 + threadIdx.x;
    const Evt2Scan mine = b < n_blocks ? agg[b] : evt2_identity();
    Evt2Scan total;
    const Evt2Scan incl = evt2_block_scan(mine, buf, &total);
    s_incl[threadIdx.x] = incl;
    __syncthreads();
    if (b < n_blocks) agg[b] = threadIdx.x ? evt2_combine(carry, s_incl[threadIdx.x - 1]) : carry;
    carry = evt2_combine(carry, total);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    Evt3State o;
    o.y = o.base_x = o.base_p = o.t_low = 0;
    o.have_high = s.have_high | carry.hi_word;
    o.t_high = carry.hi_last;
    o.t_loops = s.t_loops + carry.hi_wraps;
    o.n_events = carry.n_ev - evt_dropped(carry.n_pre, s.have_high, wait);
    *st_out = o;
    if (count_out) *count_out = (u32)o.n_events;  // (a cell of the consumer's: this record is rewritten two chunks from now)
  }
}

// 3. the records: every block re-scans its words from its exclusive prefix and writes its events
__global__ __launch_bounds__(EVT3_THREADS) void k_evt2_emit(const u32* __restrict__ words, u32 n, const Evt2Scan* __restrict__ prefix,
                                                           const Evt3State* __restrict__ st_in, uint4* __restrict__ out, u32 out_cap, int wait) {
  __shared__ Evt2Scan buf[2][EVT3_THREADS];
  __shared__ Evt2Scan s_incl[EVT3_THREADS];
  const unsigned long long loops0 = st_in->t_loops;
  const u32 have_high = st_in->have_high;
  const u32 i0 = blockIdx.x * EVT3_PER_BLOCK + threadIdx.x * EVT3_IPT;
  u32 w[EVT3_IPT];
  Evt2Scan acc = evt2_identity();
#pragma unroll
  for (int k = 0; k < EVT3_IPT; ++k) {
    w[k] = i0 + k < n ? words[i0 + k] : 0xE0000000u;  // (OTHERS: skipped)
    if (i0 + k < n) acc = evt2_combine(acc, evt2_element(w[k]));
  }
  Evt2Scan total;
  const Evt2Scan incl = evt2_block_scan(acc, buf, &total);
  s_incl[threadIdx.x] = incl;
  __syncthreads();
  Evt2Scan run = prefix[blockIdx.x];
  if (threadIdx.x) run = evt2_combine(run, s_incl[threadIdx.x - 1]);
#pragma unroll
  for (int k = 0; k < EVT3_IPT; ++k) {
    if (i0 + k >= n) break;
    const u32 before = run.n_ev - evt_dropped(run.n_pre, have_high, wait);
    const Evt2Scan e = evt2_element(w[k]);
    run = evt2_combine(run, e);
    if (!e.n_ev || before >= out_cap) continue;
    if (wait && !have_high && !run.hi_word) continue;  // in front of the stream's first EVT_TIME_HIGH: not emitted
    const unsigned long long t = ((loops0 + run.hi_wraps) << 34) | ((unsigned long long)run.hi_last << 6) | (unsigned long long)((w[k] >> 22) & 0x3fu);
    const u32 x = (w[k] >> 11) & 0x7ffu, y = w[k] & 0x7ffu;
    out[before] = make_uint4(x | (y << 16), w[k] >> 28, (u32)t, (u32)(t >> 32));
  }
}

}  // namespace xm
