// xm_api_evt3.hpp -- C-ABI: EVT 3.0 / EVT 2.0 words -> EventCD records on the device (xmaps_evt3.hpp, xmaps_evt2.hpp), alone or
// straight into the ingest.  One decoder object for both encodings (`format`): same buffers, same state record, same three launches.
// (part of libxmaps_hip.so's host side: included by ../xmaps_hip.hip, one translation unit; see that file for the order)
#pragma once

struct xm_evt3 {
  xm_handle* h = nullptr;
  int device = 0;  // (kept here: the decoder may be destroyed after its handle)
  hipStream_t stream = nullptr;
  size_t max_words = 0, max_events = 0;
  int format = 3;                // 3: 16-bit EVT 3.0 words; 2: 32-bit EVT 2.0 words
  size_t word_bytes = 2;
  uint16_t* h_words = nullptr;   // pinned staging (max_words * word_bytes)
  uint16_t* d_words = nullptr;
  Evt3Scan* d_agg = nullptr;
  Evt3State* d_state = nullptr;  // [2]: in / out, swapped per chunk
  Evt3State* h_state = nullptr;  // pinned: the chunk's event count comes back here
  uint4* d_out = nullptr;        // records of the last xm_evt3_decode
  int cur = 0;
  int wait_tb = 0;               // xm_evt3_wait_for_time_base: events in front of the stream's first TIME_HIGH word are not emitted
};

namespace {

// words (host) -> records at `out` (device, room for out_cap) enqueued on `stream`; the chunk's event count is left in
// d->d_state[d->cur ^ 1].n_events (device memory) -- evt3_commit() flips `cur` once the caller has decided to keep the chunk
// count_out (device, may be NULL): the chunk's event count once more, in a cell of the caller's (the state record's copy is
// overwritten two chunks later -- too soon for a consumer on another stream)
int evt3_enqueue(xm_evt3* d, const void* words_host, size_t n_words, bool pinned, uint4* out, size_t out_cap, hipStream_t stream,
                 u32* count_out = nullptr) {
  if (n_words > d->max_words) return fail(XM_ERR_TOO_MANY, "chunk of %zu words exceeds max_words %zu", n_words, d->max_words);
  if (!pinned) {  // pageable memory: through the pinned staging buffer, once its previous chunk has been copied out of it
    HIP_TRY(hipStreamSynchronize(stream));
    memcpy(d->h_words, words_host, n_words * d->word_bytes);
  }
  HIP_TRY(hipMemcpyAsync(d->d_words, pinned ? words_host : (const void*)d->h_words, n_words * d->word_bytes, hipMemcpyHostToDevice, stream));
  const u32 n = (u32)n_words, nb = (u32)grid_for(n_words, EVT3_PER_BLOCK);
  Evt3State* st_in = d->d_state + d->cur;
  Evt3State* st_out = d->d_state + (d->cur ^ 1);
  if (d->format == 2) {
    const u32* w32 = reinterpret_cast<const u32*>(d->d_words);
    Evt2Scan* agg = reinterpret_cast<Evt2Scan*>(d->d_agg);
    hipLaunchKernelGGL(k_evt2_aggregate, dim3(nb), dim3(EVT3_THREADS), 0, stream, w32, n, agg);
    hipLaunchKernelGGL(k_evt2_prefix, dim3(1), dim3(EVT3_THREADS), 0, stream, nb, agg, (const Evt3State*)st_in, st_out, count_out, d->wait_tb);
    hipLaunchKernelGGL(k_evt2_emit, dim3(nb), dim3(EVT3_THREADS), 0, stream, w32, n, (const Evt2Scan*)agg, (const Evt3State*)st_in, out,
                       (u32)std::min<size_t>(out_cap, 0xffffffffu), d->wait_tb);
    HIP_TRY(hipGetLastError());
    return XM_OK;
  }
  hipLaunchKernelGGL(k_evt3_aggregate, dim3(nb), dim3(EVT3_THREADS), 0, stream, (const uint16_t*)d->d_words, n, d->d_agg);
  hipLaunchKernelGGL(k_evt3_prefix, dim3(1), dim3(EVT3_THREADS), 0, stream, (const uint16_t*)d->d_words, nb, d->d_agg, (const Evt3State*)st_in, st_out,
                     count_out, d->wait_tb);
  hipLaunchKernelGGL(k_evt3_emit, dim3(nb), dim3(EVT3_THREADS), 0, stream, (const uint16_t*)d->d_words, n, (const Evt3Scan*)d->d_agg,
                     (const Evt3State*)st_in, out, (u32)std::min<size_t>(out_cap, 0xffffffffu), d->wait_tb);
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

// the synchronous form: *n_events once the count is back (synchronises the stream)
int evt3_run(xm_evt3* d, const void* words_host, size_t n_words, bool pinned, uint4* out, size_t out_cap, hipStream_t stream, size_t* n_events) {
  *n_events = 0;
  if (!n_words) return XM_OK;
  int rc = evt3_enqueue(d, words_host, n_words, pinned, out, out_cap, stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(d->h_state, d->d_state + (d->cur ^ 1), sizeof(Evt3State), hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  *n_events = (size_t)d->h_state->n_events;
  // Too many events for `out`: the decoder's state is NOT advanced (st_in is still current), so the caller can decode the same
  // chunk again into a larger buffer or in halves; the records written so far are a truncated prefix and must not be used.
  if (*n_events > out_cap) return fail(XM_ERR_TOO_MANY, "the chunk decodes to %zu events, room for %zu (decoder state unchanged: decode it again in smaller pieces)", *n_events, out_cap);
  d->cur ^= 1;
  return XM_OK;
}

// one chunk of words, the copy side: H2D + the three decode launches on the decoder's stream + the event behind them
int ingest_words_to_events(const xm_evt3* d) { return d && d->format == 2 ? 1 : 12; }

int ingest_copy_evt3(xm_ingest* g, xm_evt3* d, int k, const void* words, size_t n_words, bool pinned) {
  if (n_words) {
    int rc = evt3_enqueue(d, words, n_words, pinned, g->d_pkt[k], (size_t)g->max_packet, d->stream, g->d_pkt_n + k);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(g->copied_ev[k], d->stream));
    d->cur ^= 1;
  }
  return XM_OK;
}

// ... the launch side: everything behind it, nothing waited for: the ingest's kernels read the chunk's event count on the device
int ingest_issue_evt3(xm_ingest* g, xm_evt3* d, int k, const void* words, size_t n_words, bool pinned, bool arrived) {
  g->out_serial_now = !g->opt_evt3_out_stream;  // (see xm_ingest::out_serial_now)
  int rc0 = arrived ? XM_OK : ingest_copy_evt3(g, d, k, words, n_words, pinned);
  if (rc0) return rc0;
  if (n_words) HIP_TRY(hipStreamWaitEvent(g->stream, g->copied_ev[k], 0));
  // (upper bound of the chunk's events for the host's bookkeeping: an EVT 3.0 vector word yields up to 12, everything else at most one)
  const size_t bound = std::min<size_t>((size_t)g->max_packet, n_words * (d->format == 2 ? 1 : 12));
  return ingest_process(g, k, n_words ? bound : 0, nullptr, n_words ? g->d_pkt_n + k : nullptr);
}

}  // namespace

extern "C" {

static int evt_create(xm_handle* h, int format, size_t max_words, size_t max_events, xm_evt3** out) {
  if (!h || !out) return fail(XM_ERR_INVALID, "NULL argument");
  *out = nullptr;
  XM_ENTER(h);
  xm_evt3* d = new (std::nothrow) xm_evt3();
  if (!d) return fail(XM_ERR_NOMEM, "out of host memory");
  d->h = h;
  d->device = h->cfg.device;
  d->format = format;
  d->word_bytes = format == 2 ? 4 : 2;
  d->max_words = max_words ? max_words : (size_t)1 << 20;
  d->max_events = max_events ? max_events : (format == 2 ? d->max_words : 2 * d->max_words);
  if (d->max_words >= 0x7fffffffull || d->max_events >= 0x7fffffffull) {
    delete d;
    return fail(XM_ERR_INVALID, "max_words and max_events must be < 2^31");
  }
  const size_t nb = grid_for(d->max_words, EVT3_PER_BLOCK);
#define EV_TRY(expr)                                                             \
  do {                                                                           \
    hipError_t e_ = (expr);                                                      \
    if (e_ != hipSuccess) {                                                      \
      int rc_ = fail(XM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
      xm_evt3_destroy(d);                                                        \
      return rc_;                                                                \
    }                                                                            \
  } while (0)
  EV_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
  EV_TRY(hipHostMalloc((void**)&d->h_words, d->max_words * d->word_bytes, hipHostMallocDefault));
  EV_TRY(hipHostMalloc((void**)&d->h_state, sizeof(Evt3State), hipHostMallocDefault));
  EV_TRY(hipMalloc((void**)&d->d_words, d->max_words * d->word_bytes + 64));
  EV_TRY(hipMalloc((void**)&d->d_agg, (nb + 1) * sizeof(Evt3Scan)));
  EV_TRY(hipMalloc((void**)&d->d_state, 2 * sizeof(Evt3State)));
  EV_TRY(hipMemsetAsync(d->d_state, 0, 2 * sizeof(Evt3State), d->stream));  // (on the decoder's stream: it does not wait for the default one)
  EV_TRY(hipStreamSynchronize(d->stream));
  EV_TRY(hipMalloc((void**)&d->d_out, d->max_events * 16));
#undef EV_TRY
  *out = d;
  return XM_OK;
}

int xm_evt3_create(xm_handle* h, size_t max_words, size_t max_events, xm_evt3** out) { return evt_create(h, 3, max_words, max_events, out); }
int xm_evt2_create(xm_handle* h, size_t max_words, size_t max_events, xm_evt3** out) { return evt_create(h, 2, max_words, max_events, out); }

void xm_evt3_destroy(xm_evt3* d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  if (d->stream) (void)hipStreamSynchronize(d->stream);
  if (d->h_words) (void)hipHostFree(d->h_words);
  if (d->h_state) (void)hipHostFree(d->h_state);
  if (d->d_words) (void)hipFree(d->d_words);
  if (d->d_agg) (void)hipFree(d->d_agg);
  if (d->d_state) (void)hipFree(d->d_state);
  if (d->d_out) (void)hipFree(d->d_out);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  delete d;
}

int xm_evt3_wait_for_time_base(xm_evt3* d, int on) {
  if (!d) return fail(XM_ERR_INVALID, "NULL argument");
  d->wait_tb = on ? 1 : 0;
  return XM_OK;
}

int xm_evt3_reset(xm_evt3* d) {
  if (!d) return fail(XM_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(d->device));
  HIP_TRY(hipStreamSynchronize(d->stream));
  HIP_TRY(hipMemsetAsync(d->d_state, 0, 2 * sizeof(Evt3State), d->stream));  // (a memset on the default stream could be overtaken by the next chunk's kernels)
  HIP_TRY(hipStreamSynchronize(d->stream));
  d->cur = 0;
  return XM_OK;
}

static int evt_decode(xm_evt3* d, int format, const void* words_host, size_t n_words, const void** events_dev, size_t* n_events) {
  if (!d || (n_words && !words_host) || !n_events) return fail(XM_ERR_INVALID, "NULL argument");
  if (d->format != format) return fail(XM_ERR_INVALID, "this decoder was created for EVT %d.0 words", d->format);
  HIP_TRY(hipSetDevice(d->device));
  if (events_dev) *events_dev = d->d_out;
  return evt3_run(d, words_host, n_words, false, d->d_out, d->max_events, d->stream, n_events);
}

int xm_evt3_decode(xm_evt3* d, const uint16_t* words_host, size_t n_words, const void** events_dev, size_t* n_events) {
  return evt_decode(d, 3, words_host, n_words, events_dev, n_events);
}
int xm_evt2_decode(xm_evt3* d, const uint32_t* words_host, size_t n_words, const void** events_dev, size_t* n_events) {
  return evt_decode(d, 2, words_host, n_words, events_dev, n_events);
}

// The chunk is decoded on the DECODER's stream into the packet slot (free: its previous packet has been consumed) while the frame
// kernels of the packets before it keep running on the ingest's stream.  n_events != NULL: the decoding is waited for and the
// chunk's event count returned (and checked against max_packet_events: XM_ERR_TOO_MANY leaves decoder and ingest as they were).
// n_events == NULL: nothing is waited for -- the ingest's kernels read the count from device memory (round 4); a chunk that
// decodes to more than max_packet_events events is truncated to that many and the excess counted in the frames' `overflow`;
// the words are handed to the ingest's launch thread like a packet of records (pageable words: the call returns once they have been copied).
static int ingest_push_words(xm_ingest* g, xm_evt3* d, int format, const void* words_host, size_t n_words, int words_pinned, size_t* n_events) {
  if (!g || !d || (n_words && !words_host)) return fail(XM_ERR_INVALID, "NULL argument");
  if (d->format != format) return fail(XM_ERR_INVALID, "this decoder was created for EVT %d.0 words", d->format);
  if (g->h != d->h) return fail(XM_ERR_INVALID, "the decoder and the ingest belong to different handles");
  if (n_words > d->max_words) return fail(XM_ERR_TOO_MANY, "chunk of %zu words exceeds max_words %zu", n_words, d->max_words);
  const double c0 = ingest_now();
  HIP_TRY(hipSetDevice(g->h->cfg.device));
  int rc = ingest_take_error(g);
  if (rc) return rc;
  const int k = g->pkt_next;
  if ((rc = ingest_wait_entry(g, k))) return rc;  // the staging entry's previous packet has been consumed
  xm_ingest::Job j;
  j.k = k;
  if (n_events) {
    // the decoder runs here, on its own stream -- once the launch thread is done with every chunk handed to it before (those
    // use the same decoder: its state index and buffers are not to be touched from two threads) -- and the records then go to
    // the launch side like a packet that is already on the device
    if (g->threaded) {
      const unsigned long long posted = ingest_posted(g);
      while (g->q_done.load(std::memory_order_acquire) < posted && !g->q_error.load(std::memory_order_relaxed)) __builtin_ia32_pause();
      if ((rc = ingest_take_error(g))) return rc;
    }
    size_t n = 0;
    rc = evt3_run(d, words_host, n_words, words_pinned != 0, g->d_pkt[k], (size_t)g->max_packet, d->stream, &n);
    *n_events = n;
    if (rc) return rc;  // (XM_ERR_TOO_MANY: neither the decoder nor the ingest has advanced -- push the chunk again in halves)
    j.kind = 3; j.n = n;
  } else {
    j.kind = 1; j.n = n_words; j.host = words_host; j.dec = d; j.pinned = words_pinned != 0;
  }
  g->pkt_next = (k + 1) % xm_ingest::STAGE;
  g->posted += 1;
  j.push_no = g->posted;
  g->pkt_push[k] = g->posted;
  g->push_t[g->posted % xm_ingest::VRING] = c0;
  // (pageable words are copied by the launch side: wait until it has done so)
  rc = ingest_submit(g, j, j.kind == 1 && !j.pinned);
  g->push_host_s += ingest_now() - c0;
  g->push_calls += 1;
  return rc;
}

int xm_ingest_push_evt3(xm_ingest* g, xm_evt3* d, const uint16_t* words_host, size_t n_words, int words_pinned, size_t* n_events) {
  return ingest_push_words(g, d, 3, words_host, n_words, words_pinned, n_events);
}
int xm_ingest_push_evt2(xm_ingest* g, xm_evt3* d, const uint32_t* words_host, size_t n_words, int words_pinned, size_t* n_events) {
  return ingest_push_words(g, d, 2, words_host, n_words, words_pinned, n_events);
}

}  // extern "C"
