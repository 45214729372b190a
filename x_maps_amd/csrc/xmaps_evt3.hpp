// xmaps_evt3.hpp -- Prophesee EVT 3.0 words -> EventCD records on the device: the reader in front of the ingest, so that a
// recording crosses PCIe as it is stored (about 2-4 bytes per event) instead of as 16-byte records.  (gfx950 / MI355X; included by
// xmaps_hip.hip.)  The reference reads its recordings through Metavision's closed readers (python/bias_events_iterator.py:53-96:
// RawReaderBase(...).load_delta_t(-1) yields EventCD packets); EVT 3.0 itself is a public format -- x_maps_amd/evt3.py restates
// it on the host (Evt3Decoder, the checker of this file), the word types are listed there.
//
// The format is a state machine over 16-bit words (current row, current time, vector base column).  Every piece of that state
// at word i is "the value of the last word of type T at or before i" or a sum over the words before i, i.e. an inclusive scan
// with an associative combine:
//   * last index (+1, 0 = none in this chunk) of a TIME_LOW / ADDR_Y / VECT_BASE_X word: max;
//   * TIME_HIGH: {first / last value, index of the first, 24-bit wrap-arounds inside, index of the last word that CHANGED the
//     field} -- two ranges join with one more wrap / change test between the last value of the first and the first value of
//     the second (a wrap: the 12-bit field falls back by more than 0x800; a change restarts the low field at 0);
//   * columns consumed by VECT_12 / VECT_8 words since the last VECT_BASE_X: a segmented sum;
//   * events emitted so far: a sum (ADDR_X: 1, VECT_*: the population count of its valid bits).
// Three launches per chunk: block aggregates -> their exclusive scan in one block, seeded with the state the previous chunk
// left (which also writes the next state and the chunk's event count) -> every block re-scans its words from its prefix and
// writes the records in word order, a vector word's events in ascending column order.
#pragma once
#include "xmaps_kernels.hpp"

namespace xm {

constexpr int EVT3_THREADS = 256, EVT3_IPT = 8, EVT3_PER_BLOCK = EVT3_THREADS * EVT3_IPT;

struct Evt3State {  // what a chunk hands to the next one (Evt3Decoder's fields in x_maps_amd/evt3.py)
  u32 y, base_x, base_p, t_high, t_low;
  u32 have_high;  // a TIME_HIGH word has been seen since the stream started (the "wait for the time base" option drops events before it)
  unsigned long long t_loops;
  unsigned long long n_events;  // of the chunk that wrote this state
};

struct Evt3Scan {
  u32 lo_idx, y_idx, b_idx;  // last TIME_LOW / ADDR_Y / VECT_BASE_X word: index + 1, 0 = none
  u32 hi_has, hi_first, hi_last, hi_first_idx, hi_wraps, hi_change_idx;  // TIME_HIGH (indices + 1, 0 = none)
  u32 adv_flag, adv_sum;     // columns consumed since the last VECT_BASE_X (flag: the range holds one)
  u32 n_ev;
  u32 hi_word, n_pre;        // a TIME_HIGH WORD lies in the range (the seed is not one); events in front of the range's first one
};

__device__ __forceinline__ Evt3Scan evt3_identity() {
  Evt3Scan e;
  e.lo_idx = e.y_idx = e.b_idx = 0;
  e.hi_has = e.hi_first = e.hi_last = e.hi_first_idx = e.hi_wraps = e.hi_change_idx = 0;
  e.adv_flag = e.adv_sum = 0;
  e.n_ev = 0;
  e.hi_word = e.n_pre = 0;
  return e;
}

// a = the earlier range, b = the later one
__device__ __forceinline__ Evt3Scan evt3_combine(const Evt3Scan& a, const Evt3Scan& b) {
  Evt3Scan r;
  r.lo_idx = max(a.lo_idx, b.lo_idx);
  r.y_idx = max(a.y_idx, b.y_idx);
  r.b_idx = max(a.b_idx, b.b_idx);
  if (!b.hi_has) {
    r.hi_has = a.hi_has; r.hi_first = a.hi_first; r.hi_last = a.hi_last; r.hi_first_idx = a.hi_first_idx;
    r.hi_wraps = a.hi_wraps; r.hi_change_idx = a.hi_change_idx;
  } else if (!a.hi_has) {
    r.hi_has = 1; r.hi_first = b.hi_first; r.hi_last = b.hi_last; r.hi_first_idx = b.hi_first_idx;
    r.hi_wraps = b.hi_wraps; r.hi_change_idx = b.hi_change_idx;
  } else {
    const bool wrap = (int)a.hi_last - (int)b.hi_first > 0x800, changed = b.hi_first != a.hi_last;
    r.hi_has = 1; r.hi_first = a.hi_first; r.hi_first_idx = a.hi_first_idx; r.hi_last = b.hi_last;
    r.hi_wraps = a.hi_wraps + b.hi_wraps + (wrap ? 1u : 0u);
    r.hi_change_idx = b.hi_change_idx ? b.hi_change_idx : changed ? b.hi_first_idx : a.hi_change_idx;
  }
  r.adv_flag = a.adv_flag | b.adv_flag;
  r.adv_sum = b.adv_flag ? b.adv_sum : a.adv_sum + b.adv_sum;
  r.n_ev = a.n_ev + b.n_ev;
  r.hi_word = a.hi_word | b.hi_word;
  r.n_pre = a.hi_word ? a.n_pre : a.n_ev + b.n_pre;
  return r;
}

// Start-of-stream rule (an option of the decoder, xm_evt3_wait_for_time_base): events in front of the stream's FIRST TIME_HIGH word
// carry a time of which only the low 12 bits are known.  Off (default): they are emitted with the high field at its initial 0,
// like everything else the initial state defines.  On: they are not emitted (a reader that waits for the first time base).
__device__ __forceinline__ u32 evt_dropped(const u32 n_pre, const u32 have_high, const int wait) { return wait && !have_high ? n_pre : 0u; }

__device__ __forceinline__ Evt3Scan evt3_element(u32 w, u32 i) {  // word w at index i of the chunk
  Evt3Scan e = evt3_identity();
  const u32 typ = w >> 12;
  if (typ == 0x6u) e.lo_idx = i + 1;
  else if (typ == 0x0u) e.y_idx = i + 1;
  else if (typ == 0x3u) { e.b_idx = i + 1; e.adv_flag = 1; }
  else if (typ == 0x8u) { e.hi_has = 1; e.hi_first = e.hi_last = w & 0xfffu; e.hi_first_idx = i + 1; e.hi_word = 1; }
  else if (typ == 0x2u) e.n_ev = 1;
  else if (typ == 0x4u) { e.adv_sum = 12; e.n_ev = __popc(w & 0xfffu); }
  else if (typ == 0x5u) { e.adv_sum = 8; e.n_ev = __popc(w & 0xffu); }
  e.n_pre = e.n_ev;  // (no TIME_HIGH word in a range of one event word: all of its events lie in front of "the first one")
  return e;
}

__device__ __forceinline__ Evt3Scan evt3_seed(const Evt3State& s) {  // the state in front of the chunk as a range of its own
  Evt3Scan e = evt3_identity();
  e.hi_has = 1;
  e.hi_first = e.hi_last = s.t_high;
  return e;
}

// inclusive scan of one element per thread over the block (Hillis-Steele on two LDS buffers); returns the thread's inclusive
// result, *block_total = the block's aggregate
__device__ __forceinline__ Evt3Scan evt3_block_scan(const Evt3Scan mine, Evt3Scan (*buf)[EVT3_THREADS], Evt3Scan* block_total) {
  const int tid = threadIdx.x;
  int cur = 0;
  buf[0][tid] = mine;
  __syncthreads();
  for (int o = 1; o < EVT3_THREADS; o <<= 1) {
    Evt3Scan v = buf[cur][tid];
    if (tid >= o) v = evt3_combine(buf[cur][tid - o], v);
    buf[cur ^ 1][tid] = v;
    cur ^= 1;
    __syncthreads();
  }
  const Evt3Scan r = buf[cur][tid];
  *block_total = buf[cur][EVT3_THREADS - 1];
  __syncthreads();
  return r;
}

// 1. the aggregate of every block of EVT3_PER_BLOCK words
__global__ __launch_bounds__(EVT3_THREADS) void k_evt3_aggregate(const uint16_t* __restrict__ words, u32 n, Evt3Scan* __restrict__ agg) {
  __shared__ Evt3Scan buf[2][EVT3_THREADS];
  const u32 i0 = blockIdx.x * EVT3_PER_BLOCK + threadIdx.x * EVT3_IPT;
  Evt3Scan acc = evt3_identity();
#pragma unroll
  for (int k = 0; k < EVT3_IPT; ++k)
    if (i0 + k < n) acc = evt3_combine(acc, evt3_element(words[i0 + k], i0 + k));
  Evt3Scan total;
  (void)evt3_block_scan(acc, buf, &total);
  if (threadIdx.x == 0) agg[blockIdx.x] = total;
}

// what the state machine holds once the range `r` (seed included) has been read
__device__ __forceinline__ void evt3_resolve(const Evt3Scan& r, const Evt3State& s, const uint16_t* __restrict__ words, u32& y, u32& t_high,
                                             u32& t_low, unsigned long long& loops, u32& base, u32& pol) {
  y = r.y_idx ? (u32)words[r.y_idx - 1] & 0x7ffu : s.y;
  t_high = r.hi_last;
  t_low = r.lo_idx ? (u32)words[r.lo_idx - 1] & 0xfffu : s.t_low;
  if (r.hi_change_idx > r.lo_idx) t_low = 0;  // a TIME_HIGH that changed the field restarts the low field until the next TIME_LOW
  loops = s.t_loops + r.hi_wraps;
  const u32 bw = r.b_idx ? (u32)words[r.b_idx - 1] : 0u;
  base = (r.b_idx ? bw & 0x7ffu : s.base_x) + r.adv_sum;  // (base column + what the vector words since it have consumed)
  pol = r.b_idx ? (bw >> 11) & 1u : s.base_p;
}

// 2. one block: exclusive scan of the aggregates, seeded with the previous chunk's state; the chunk's event count and the state
//    for the next chunk
__global__ __launch_bounds__(EVT3_THREADS) void k_evt3_prefix(const uint16_t* __restrict__ words, u32 n_blocks, Evt3Scan* __restrict__ agg,
                                                             const Evt3State* __restrict__ st_in, Evt3State* __restrict__ st_out,
                                                             u32* __restrict__ count_out, int wait) {
  __shared__ Evt3Scan buf[2][EVT3_THREADS];
  const Evt3State s = *st_in;
  Evt3Scan carry = evt3_seed(s);
  for (u32 b0 = 0; b0 < n_blocks; b0 += EVT3_THREADS) {
    const u32 b = b0 + threadIdx.x;
    const Evt3Scan mine = b < n_blocks ? agg[b] : evt3_identity();
    Evt3Scan total;
    const Evt3Scan incl = evt3_block_scan(mine, buf, &total);
    // exclusive prefix of block b = carry + (inclusive of b - 1): recompute from the neighbour's inclusive value
    __shared__ Evt3Scan s_incl[EVT3_THREADS];
    s_incl[threadIdx.x] = incl;
    __syncthreads();
    if (b < n_blocks) agg[b] = threadIdx.x ? evt3_combine(carry, s_incl[threadIdx.x - 1]) : carry;
    carry = evt3_combine(carry, total);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    Evt3State o;
    u32 y, th, tl, base, pol;
    unsigned long long loops;
    evt3_resolve(carry, s, words, y, th, tl, loops, base, pol);
    o.y = y; o.base_x = base; o.base_p = pol; o.t_high = th; o.t_low = tl;
    o.have_high = s.have_high | carry.hi_word;
    o.t_loops = loops;
    o.n_events = carry.n_ev - evt_dropped(carry.n_pre, s.have_high, wait);
    *st_out = o;
    if (count_out) *count_out = (u32)o.n_events;  // (a cell of the consumer's: this record is rewritten two chunks from now)
  }
}

// 3. the records: every block re-scans its words from its exclusive prefix and writes its events
__global__ __launch_bounds__(EVT3_THREADS) void k_evt3_emit(const uint16_t* __restrict__ words, u32 n, const Evt3Scan* __restrict__ prefix,
                                                           const Evt3State* __restrict__ st_in, uint4* __restrict__ out, u32 out_cap, int wait) {
  __shared__ Evt3Scan buf[2][EVT3_THREADS];
  const Evt3State s = *st_in;
  const u32 i0 = blockIdx.x * EVT3_PER_BLOCK + threadIdx.x * EVT3_IPT;
  u32 w[EVT3_IPT];
  Evt3Scan acc = evt3_identity();
#pragma unroll
  for (int k = 0; k < EVT3_IPT; ++k) {
    w[k] = i0 + k < n ? (u32)words[i0 + k] : 0xE000u;  // (OTHERS: skipped)
    if (i0 + k < n) acc = evt3_combine(acc, evt3_element(w[k], i0 + k));
  }
  Evt3Scan total;
  const Evt3Scan incl = evt3_block_scan(acc, buf, &total);
  // exclusive prefix of this thread's first word = block prefix + the threads in front of it
  __shared__ Evt3Scan s_incl[EVT3_THREADS];
  s_incl[threadIdx.x] = incl;
  __syncthreads();
  Evt3Scan run = prefix[blockIdx.x];
  if (threadIdx.x) run = evt3_combine(run, s_incl[threadIdx.x - 1]);
#pragma unroll
  for (int k = 0; k < EVT3_IPT; ++k) {
    if (i0 + k >= n) break;
    const u32 before = run.n_ev - evt_dropped(run.n_pre, s.have_high, wait);
    const Evt3Scan e = evt3_element(w[k], i0 + k);
    run = evt3_combine(run, e);
    if (!e.n_ev) continue;
    if (wait && !s.have_high && !run.hi_word) continue;  // in front of the stream's first TIME_HIGH word: not emitted
    u32 y, th, tl, base, pol;
    unsigned long long loops;
    evt3_resolve(run, s, words, y, th, tl, loops, base, pol);
    const unsigned long long t = (loops << 24) | ((unsigned long long)th << 12) | (unsigned long long)tl;
    const u32 typ = w[k] >> 12;
    if (typ == 0x2u) {
      if (before < out_cap) out[before] = make_uint4((w[k] & 0x7ffu) | (y << 16), (w[k] >> 11) & 1u, (u32)t, (u32)(t >> 32));
    } else {
      const u32 vb = base - e.adv_sum;  // the columns this word's bits stand for start where the earlier vector words stopped
      u32 bits = w[k] & (typ == 0x4u ? 0xfffu : 0xffu), o = before;
      while (bits) {
        const u32 bit = __ffs(bits) - 1;
        bits &= bits - 1;
        if (o < out_cap) out[o] = make_uint4(((vb + bit) & 0xffffu) | (y << 16), pol, (u32)t, (u32)(t >> 32));
        o += 1;
      }
    }
  }
}

}  // namespace xm
