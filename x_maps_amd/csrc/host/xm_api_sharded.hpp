// xm_api_sharded.hpp -- C-ABI: one frame sharded by event index over several GPUs of ONE process (last row:
// xm_create_sharded owning the RCCL communicators; 8(e): the partitioning and the exchange)
// (part of libxmaps_hip.so's host side: included by ../xmaps_hip.hip, one translation unit; see that file for the order)
//
//   device g owns events [g N / W, (g + 1) N / W) of the frame; tables are replicated (one xm_handle per device).
//   per frame, one host thread per device, everything on that device's stream:
//     H2D of the shard -> xm_shard_minmax_device -> ncclAllReduce(MIN, {tmin, -tmax}) -> xm_shard_clear + xm_shard_scatter_device
//     (global event indices in the packed keys) -> ncclAllReduce(MAX, uint64 key frame) -> device 0: xm_shard_finish -> D2H.
//   Time-sorted int64 frames on rigs whose X-map is injective take the COLUMNS exchange instead (xm_shard_cols_*: ncclAllGather of
//   the shards' headers + last events, every time column on one device, ncclAllReduce(SUM) of the plain u16 frames: 2 bytes per
//   cell on the wire, no atomics, no extrema pass); a frame one of whose pieces objects is redone with the keys (xm_sharded_stats).
//   MAX over packed keys = the event with the largest GLOBAL index wins = NumPy's last-writer-wins across shards, bit for bit
//   (x_maps_amd/sharded.py is the same exchange for multi-process hosts on torch.distributed).
// Failure: everything that can fail on the host (allocations, launches that report at once) happens in front of an AGREEMENT
// among the device threads (a host barrier that ORs their error codes) placed before every collective: either every thread
// enters the collective or none does -- no thread is left waiting in RCCL for a peer that has returned.
// xm_debug_option("XM_SHARD_FAKE_RANKS", "W") (tests; n_dev = 1): W VIRTUAL ranks on the one device -- W handles, W threads, the
// same per-rank chain, the two collectives emulated by host barriers + a reduction kernel over the ranks' buffers -- so that the
// N > 1 orchestration (shard bounds, the columns exchange's predecessor logic, the merge, the agreement) runs through this C
// entry on a one-GPU box (RCCL refuses two ranks on one device).
// RCCL is not linked: librccl is looked up at run time (the copy a host process has loaded already -- PyTorch ships its own --
// else ROCm's), so that the library keeps loading on hosts without it; xm_create_sharded reports its absence for n_dev > 1.
#pragma once

#include <dlfcn.h>

namespace {

// the few RCCL entry points and enum values used here (rccl.h: ncclDataType_t / ncclRedOp_t)
struct RcclApi {
  void* lib = nullptr;
  struct UniqueId { char bytes[128]; };  // ncclUniqueId: 128 opaque bytes, passed by value
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
  int (*GetUniqueId)(UniqueId* id) = nullptr;
  int (*CommInitRank)(void** comm, int nranks, UniqueId id, int rank) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*AllReduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) = nullptr;
  int (*AllGather)(const void* send, void* recv, size_t sendcount, int dtype, void* comm, hipStream_t stream) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  static constexpr int Uint8 = 1, Int32 = 2, Int64 = 4, Uint64 = 5, Float64 = 8, Sum = 0, Max = 2, Min = 3;
  bool ok() const { return CommInitAll && CommDestroy && AllReduce; }
  bool ok_ranks() const { return ok() && GetUniqueId && CommInitRank && AllGather; }
  const char* err(int e) const { return GetErrorString ? GetErrorString(e) : "?"; }
};

RcclApi load_rccl() {
  RcclApi r;
  const char* loaded[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : loaded)
    if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);  // a copy the process has already (PyTorch's)
  const char* fresh[] = {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
  for (const char* n : fresh)
    if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  if (!r.lib) return r;
  r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
  return r;
}

}  // namespace

namespace xm {
// out[i] = op over r of bufs[r][i]  (virtual ranks: every buffer lives on the one device)
template <typename T, int OP>  // OP: 0 sum, 1 max, 2 min
__global__ __launch_bounds__(256) void k_fake_reduce(const T* const* __restrict__ bufs, int W, size_t n, T* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    T v = bufs[0][i];
    for (int r = 1; r < W; ++r) {
      const T u = bufs[r][i];
      v = OP == 0 ? (T)(v + u) : OP == 1 ? (u > v ? u : v) : (u < v ? u : v);
    }
    out[i] = v;
  }
}
}  // namespace xm

struct xm_sharded {
  struct Dev {
    int id = 0;
    xm_handle* h = nullptr;
    void* comm = nullptr;
    DevBuf x, y, t, p, depth, bgr;
    DevBuf send, gathered;         // columns exchange: this device's header + last events, every device's
    uint16_t* frame16 = nullptr;   // columns exchange: the plain u16 disparity frame (+ the boundary pass' scratch)
    int flagged = 0;               // columns exchange: this device's piece could not be handled
    uint64_t* key = nullptr;
    void* mm = nullptr;            // {tmin, -tmax} of the shard, then of the frame (16 bytes, int64 or float64)
    hipEvent_t ev[4] = {};         // device 0: around the two all-reduces
    bool peer_only = false;        // this frame: the device itself was fine, it stopped because a peer had failed
    long long mm_back[2] = {0, 0}; // device 0: the frame's {tmin, -tmax} copied back (the copy outlives an early return: not on the stack)
    DevBuf fake_tmp;               // virtual ranks: the reduction's result before it replaces the rank's own buffer
    const void** fake_ptrs = nullptr;  // virtual ranks: device array of the W ranks' buffers
    const void* fake_cur = nullptr;    // virtual ranks: the buffer this rank brings to the collective in flight
    std::thread th;
    int rc = XM_OK;
    std::string err;
  };
  std::vector<std::unique_ptr<Dev>> devs;
  RcclApi rccl;
  bool use_rccl = false;
  bool fake = false;               // XM_SHARD_FAKE_RANKS: virtual ranks on one device, collectives emulated
  int fail_dev = -1, fail_point = 0;  // XM_SHARD_FAIL_AT (tests)
  // agreement / barrier among the device threads (sharded_agree)
  std::mutex bmu;
  std::condition_variable bcv;
  int b_arrived = 0, b_rc = 0, b_rc_out = 0;
  unsigned long long b_gen = 0;
  int b_poison = 0;                // != 0: a device thread has LEFT the frame with this error outside an agreement point -- nobody waits for it any more
  // the frame in flight (set by xm_sharded_process_frame, read by the device threads)
  const uint16_t *x = nullptr, *y = nullptr;
  const void* t = nullptr;
  const int16_t* p = nullptr;
  size_t n = 0;
  int t_dtype = XM_T_INT64;
  float* depth_out = nullptr;
  uint8_t* bgr_out = nullptr;
  u32 tag = 0;
  // the exchange of the frame in flight: every time column on one device + SUM of u16 frames (xm_shard_cols_*), or packed keys
  bool cols_now = false;
  size_t cols_cap = 0, cols_send_bytes = 0, cols_reduce_u32 = 0, cols_frame_bytes = 0;
  unsigned long long frames_columns = 0, frames_keys = 0, frames_redone = 0;
  double mm_host[2] = {0, 0};
  float coll_ms[2] = {0, 0};
  // start / done hand-shake
  std::mutex mu;
  std::condition_variable cv;
  unsigned long long gen = 0;
  int done = 0;
  bool stop = false;
};

namespace {

// fault injection for the tests: xm_debug_option("XM_SHARD_FAIL_AT", "<device index>:<point>") makes that device thread fail in
// front of collective <point> (1: the first of the frame, 2: the second; 3: BEHIND the first agreement, i.e. outside any agreement
// point -- its peers are then on their way into the collective and must be woken by the poisoned barrier) -- read when the handle is created
bool dbg_fail_at(const xm_sharded* s, int g, int point) { return s->fail_dev == g && s->fail_point == point; }

// Host barrier among the device threads; returns the first non-zero rc any of them brought (0: everybody is fine).
// A thread that leaves the frame with an error anywhere else -- hipSetDevice at the top, a collective that returned an error
// right behind an agreement, the copy at the end of a virtual all-reduce -- will never arrive at the next agreement: it POISONS
// the barrier on its way out (sharded_leave), which wakes everybody waiting here and makes every later arrival of the frame
// return at once with that error.  xm_sharded_process_frame clears the barrier before it starts the next frame.
int sharded_agree(xm_sharded* s, int rc) {
  const int W = (int)s->devs.size();
  if (W == 1) return rc;
  std::unique_lock<std::mutex> lk(s->bmu);
  if (s->b_poison) return rc ? rc : s->b_poison;
  if (rc && !s->b_rc) s->b_rc = rc;
  const unsigned long long gen = s->b_gen;
  if (++s->b_arrived == W) {
    s->b_rc_out = s->b_rc;
    s->b_rc = 0;
    s->b_arrived = 0;
    s->b_gen += 1;
    s->bcv.notify_all();
  } else {
    s->bcv.wait(lk, [&] { return s->b_gen != gen || s->b_poison != 0; });
    if (s->b_gen == gen) return rc ? rc : s->b_poison;  // (poisoned while waiting: the round never completes)
  }
  return s->b_rc_out;
}
// a device thread leaves the frame with rc (called once per thread and frame, wherever it returned from)
void sharded_leave(xm_sharded* s, int rc) {
  if (!rc || s->devs.size() == 1) return;
  {
    std::lock_guard<std::mutex> lk(s->bmu);
    if (!s->b_poison) s->b_poison = rc;
  }
  s->bcv.notify_all();
}
// (a peer failed: this thread has nothing to report itself -- xm_sharded_process_frame reports the peer's error, not this one)
int sharded_peer_failed(xm_sharded::Dev& d, int rc_agreed, int rc_own) {
  if (rc_own) return rc_own;
  d.peer_only = true;
  return fail(rc_agreed, "another device of the sharded handle failed in front of a collective");
}

// virtual ranks: ncclAllGather / ncclAllReduce over buffers that all live on the one device.  Every rank: stream-sync (its
// contribution is complete), barrier, read everybody's buffer on its own stream, barrier (nobody overwrites a buffer a peer is
// still reading), [all-reduce: result -> own buffer].
int fake_all_gather(xm_sharded* s, int g, const void* send, void* recv, size_t bytes, hipStream_t st) {
  xm_sharded::Dev& d = *s->devs[g];
  const int W = (int)s->devs.size();
  d.fake_cur = send;
  int rc = hipStreamSynchronize(st) == hipSuccess ? XM_OK : fail(XM_ERR_HIP, "hipStreamSynchronize failed");
  if (const int agreed = sharded_agree(s, rc)) return sharded_peer_failed(d, agreed, rc);
  for (int r = 0; r < W && !rc; ++r)
    if (hipMemcpyAsync((char*)recv + (size_t)r * bytes, s->devs[r]->fake_cur, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
      rc = fail(XM_ERR_HIP, "hipMemcpyAsync (virtual all-gather) failed");
  if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = fail(XM_ERR_HIP, "hipStreamSynchronize failed");
  if (const int agreed = sharded_agree(s, rc)) return sharded_peer_failed(d, agreed, rc);
  return XM_OK;
}

template <typename T, int OP>
int fake_all_reduce(xm_sharded* s, int g, T* buf, size_t count, hipStream_t st) {
  xm_sharded::Dev& d = *s->devs[g];
  const int W = (int)s->devs.size();
  d.fake_cur = buf;
  int rc = d.fake_tmp.reserve(count * sizeof(T));
  if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = fail(XM_ERR_HIP, "hipStreamSynchronize failed");
  if (const int agreed = sharded_agree(s, rc)) return sharded_peer_failed(d, agreed, rc);
  std::vector<const void*> ptrs(W);
  for (int r = 0; r < W; ++r) ptrs[r] = s->devs[r]->fake_cur;
  if (hipMemcpyAsync(d.fake_ptrs, ptrs.data(), sizeof(void*) * W, hipMemcpyHostToDevice, st) != hipSuccess) rc = fail(XM_ERR_HIP, "hipMemcpyAsync failed");
  if (!rc) {
    const unsigned gx = (unsigned)std::min<size_t>(4096, (count + 255) / 256);
    hipLaunchKernelGGL((k_fake_reduce<T, OP>), dim3(gx ? gx : 1), dim3(256), 0, st, (const T* const*)d.fake_ptrs, W, count, (T*)d.fake_tmp.p);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = fail(XM_ERR_HIP, "virtual all-reduce kernel failed");
  }
  if (const int agreed = sharded_agree(s, rc)) return sharded_peer_failed(d, agreed, rc);  // (everybody has read everybody's buffer)
  if (hipMemcpyAsync(buf, d.fake_tmp.p, count * sizeof(T), hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(XM_ERR_HIP, "hipMemcpyAsync failed");
  return XM_OK;
}

void sharded_frame_on(xm_sharded* s, int g) {
  xm_sharded::Dev& d = *s->devs[g];
  const int W = (int)s->devs.size();
  d.rc = XM_OK;
  d.peer_only = false;
  const auto run = [&]() -> int {
    HIP_TRY(hipSetDevice(d.id));
    hipStream_t st = (hipStream_t)xm_stream(d.h, 0);
    const size_t a = (size_t)(((unsigned __int128)g * s->n) / (unsigned)W), b = (size_t)(((unsigned __int128)(g + 1) * s->n) / (unsigned)W);
    const size_t m = b - a, tsz = t_size(s->t_dtype);
    int rc;
    d.flagged = 0;
    if (s->cols_now) {
      // every time column on one device, the plain u16 frames merged by SUM (xm_api_shard.hpp: xm_shard_cols_*): the shard goes
      // into its buffers behind cap + 8 events of headroom (the predecessor's last column is copied there on the device)
      const size_t hr = s->cols_cap + 8;
      uint16_t *dx = nullptr, *dy = nullptr;
      int64_t* dt = nullptr;
      // everything that can fail before the all-gather, then the agreement: every device enters the collective or none does
      const auto before_gather = [&]() -> int {
        int rc;
        if (dbg_fail_at(s, g, 1)) return fail(XM_ERR_HIP, "injected failure (XM_SHARD_FAIL_AT) on device index %d in front of the all-gather", g);
        if ((rc = d.x.reserve((hr + m + 8) * 2)) || (rc = d.y.reserve((hr + m + 8) * 2)) || (rc = d.t.reserve((hr + m + 8) * 8))) return rc;
        if ((rc = d.send.reserve(s->cols_send_bytes)) || (rc = d.gathered.reserve(s->cols_send_bytes * (size_t)W))) return rc;
        if (g == 0 && s->depth_out && (rc = d.depth.reserve((size_t)d.h->out_w * d.h->out_h * 4))) return rc;
        if (g == 0 && s->bgr_out && (rc = d.bgr.reserve((size_t)d.h->out_w * d.h->out_h * 3))) return rc;
        if (!d.frame16) {
          HIP_TRY(hipMalloc((void**)&d.frame16, s->cols_frame_bytes));
          HIP_TRY(hipMemsetAsync(d.frame16, 0, s->cols_frame_bytes, st));
        }
        dx = (uint16_t*)d.x.p + hr;
        dy = (uint16_t*)d.y.p + hr;
        dt = (int64_t*)d.t.p + hr;
        HIP_TRY(hipMemcpyAsync(dx, s->x + a, m * 2, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(dy, s->y + a, m * 2, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(dt, (const int64_t*)s->t + a, m * 8, hipMemcpyHostToDevice, st));
        if ((rc = xm_shard_cols_pack(d.h, dx, dy, dt, m, d.send.p, s->cols_cap))) return rc;
        if (g == 0) HIP_TRY(hipEventRecord(d.ev[0], st));
        return XM_OK;
      };
      rc = before_gather();
      if (const int agreed = sharded_agree(s, rc)) return sharded_peer_failed(d, agreed, rc);
      if (dbg_fail_at(s, g, 3)) return fail(XM_ERR_HIP, "injected failure (XM_SHARD_FAIL_AT) on device index %d BEHIND the agreement, outside any agreement point", g);
      const void* gathered = d.send.p;  // (one device without RCCL: its own send buffer is the gathered buffer)
      if (s->fake) {
        if ((rc = fake_all_gather(s, g, d.send.p, d.gathered.p, s->cols_send_bytes, st))) return rc;
        gathered = d.gathered.p;
      } else if (s->use_rccl) {
        const int e = s->rccl.AllGather(d.send.p, d.gathered.p, s->cols_send_bytes, RcclApi::Uint8, d.comm, st);
        if (e) return fail(XM_ERR_HIP, "ncclAllGather(headers + last events) failed: %s", s->rccl.err(e));
        gathered = d.gathered.p;
      }
      const auto before_reduce = [&]() -> int {
        int rc;
        if (dbg_fail_at(s, g, 2)) return fail(XM_ERR_HIP, "injected failure (XM_SHARD_FAIL_AT) on device index %d in front of the all-reduce", g);
        if (g == 0) HIP_TRY(hipEventRecord(d.ev[1], st));
        if ((rc = xm_shard_cols_scatter(d.h, dx, dy, dt, m, s->n, gathered, s->cols_send_bytes, g, W, s->cols_cap, d.frame16))) return rc;
        if (g == 0) HIP_TRY(hipEventRecord(d.ev[2], st));
        return XM_OK;
      };
      rc = before_reduce();
      if (const int agreed = sharded_agree(s, rc)) return sharded_peer_failed(d, agreed, rc);
      if (s->fake) {
        if ((rc = fake_all_reduce<u32, 0>(s, g, (u32*)d.frame16, s->cols_reduce_u32, st))) return rc;
      } else if (s->use_rccl) {
        const int e = s->rccl.AllReduce(d.frame16, d.frame16, s->cols_reduce_u32, RcclApi::Int32, RcclApi::Sum, d.comm, st);
        if (e) return fail(XM_ERR_HIP, "ncclAllReduce(SUM, u16 frame) failed: %s", s->rccl.err(e));
      }
      if (g == 0) {
        HIP_TRY(hipEventRecord(d.ev[3], st));
        const size_t px = (size_t)d.h->out_w * d.h->out_h;
        float* dd = nullptr;
        uint8_t* db = nullptr;
        if (s->depth_out) {
          if ((rc = d.depth.reserve(px * 4))) return rc;
          dd = (float*)d.depth.p;
        }
        if (s->bgr_out) {
          if ((rc = d.bgr.reserve(px * 3))) return rc;
          db = (uint8_t*)d.bgr.p;
        }
        if ((dd || db) && (rc = xm_shard_finish_u16(d.h, d.frame16, dd, db))) return rc;
        if (dd) HIP_TRY(hipMemcpyAsync(s->depth_out, dd, px * 4, hipMemcpyDeviceToHost, st));
        if (db) HIP_TRY(hipMemcpyAsync(s->bgr_out, db, px * 3, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(d.mm_back, d.h->d_shard_n, 16, hipMemcpyDeviceToHost, st));  // {tmin, -tmax} of the frame (prepare left them there)
        if ((rc = xm_shard_cols_failed(d.h, &d.flagged))) {                                 // (synchronises the stream)
          (void)hipStreamSynchronize(st);
          return rc;
        }
        s->mm_host[0] = (double)d.mm_back[0];
        s->mm_host[1] = -(double)d.mm_back[1];
        HIP_TRY(hipEventElapsedTime(&s->coll_ms[0], d.ev[0], d.ev[1]));
        HIP_TRY(hipEventElapsedTime(&s->coll_ms[1], d.ev[2], d.ev[3]));
      } else if ((rc = xm_shard_cols_failed(d.h, &d.flagged))) {
        return rc;
      }
      return XM_OK;
    }
    const int16_t* dp = nullptr;
    const auto before_min = [&]() -> int {
      int rc;
      if (dbg_fail_at(s, g, 1)) return fail(XM_ERR_HIP, "injected failure (XM_SHARD_FAIL_AT) on device index %d in front of the extrema all-reduce", g);
      if ((rc = stage_in(d.x, s->x + a, m * 2, st))) return rc;
      if ((rc = stage_in(d.y, s->y + a, m * 2, st))) return rc;
      if ((rc = stage_in(d.t, (const char*)s->t + a * tsz, m * tsz, st))) return rc;
      if (s->p && (rc = stage_in(d.p, s->p + a, m * 2, st))) return rc;
      if (g == 0 && s->depth_out && (rc = d.depth.reserve((size_t)d.h->out_w * d.h->out_h * 4))) return rc;
      if (g == 0 && s->bgr_out && (rc = d.bgr.reserve((size_t)d.h->out_w * d.h->out_h * 3))) return rc;
      dp = s->p ? (const int16_t*)d.p.p : nullptr;
      if ((rc = xm_shard_minmax_device(d.h, d.t.p, dp, m, s->t_dtype, d.mm))) return rc;
      if (g == 0) HIP_TRY(hipEventRecord(d.ev[0], st));
      return XM_OK;
    };
    rc = before_min();
    if (const int agreed = sharded_agree(s, rc)) return sharded_peer_failed(d, agreed, rc);
    if (dbg_fail_at(s, g, 3)) return fail(XM_ERR_HIP, "injected failure (XM_SHARD_FAIL_AT) on device index %d BEHIND the agreement, outside any agreement point", g);
    if (s->fake) {
      rc = s->t_dtype == XM_T_INT64 ? fake_all_reduce<long long, 2>(s, g, (long long*)d.mm, 2, st) : fake_all_reduce<double, 2>(s, g, (double*)d.mm, 2, st);
      if (rc) return rc;
    } else if (s->use_rccl) {
      const int e = s->rccl.AllReduce(d.mm, d.mm, 2, s->t_dtype == XM_T_INT64 ? RcclApi::Int64 : RcclApi::Float64, RcclApi::Min, d.comm, st);
      if (e) return fail(XM_ERR_HIP, "ncclAllReduce(MIN, extrema) failed: %s", s->rccl.GetErrorString ? s->rccl.GetErrorString(e) : "?");
    }
    const auto before_max = [&]() -> int {
      int rc;
      if (dbg_fail_at(s, g, 2)) return fail(XM_ERR_HIP, "injected failure (XM_SHARD_FAIL_AT) on device index %d in front of the key all-reduce", g);
      if (g == 0) HIP_TRY(hipEventRecord(d.ev[1], st));
      if ((rc = xm_shard_clear(d.h, d.key))) return rc;
      if ((rc = xm_shard_scatter_device(d.h, (const uint16_t*)d.x.p, (const uint16_t*)d.y.p, d.t.p, dp, m, s->t_dtype, (uint64_t)a, d.mm,
                                        s->tag, d.key)))
        return rc;
      if (g == 0) HIP_TRY(hipEventRecord(d.ev[2], st));
      return XM_OK;
    };
    rc = before_max();
    if (const int agreed = sharded_agree(s, rc)) return sharded_peer_failed(d, agreed, rc);
    if (s->fake) {
      if ((rc = fake_all_reduce<unsigned long long, 1>(s, g, (unsigned long long*)d.key, d.h->key_cells, st))) return rc;
    } else if (s->use_rccl) {
      const int e = s->rccl.AllReduce(d.key, d.key, d.h->key_cells, RcclApi::Uint64, RcclApi::Max, d.comm, st);
      if (e) return fail(XM_ERR_HIP, "ncclAllReduce(MAX, key frame) failed: %s", s->rccl.GetErrorString ? s->rccl.GetErrorString(e) : "?");
    }
    if (g == 0) {
      HIP_TRY(hipEventRecord(d.ev[3], st));
      const size_t px = (size_t)d.h->out_w * d.h->out_h;
      float* dd = s->depth_out ? (float*)d.depth.p : nullptr;
      uint8_t* db = s->bgr_out ? (uint8_t*)d.bgr.p : nullptr;
      if ((rc = xm_shard_finish(d.h, d.key, s->tag, dd, db))) return rc;
      if (dd) HIP_TRY(hipMemcpyAsync(s->depth_out, dd, px * 4, hipMemcpyDeviceToHost, st));
      if (db) HIP_TRY(hipMemcpyAsync(s->bgr_out, db, px * 3, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(d.mm_back, d.mm, 16, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (s->t_dtype == XM_T_INT64) {
        s->mm_host[0] = (double)d.mm_back[0];
        s->mm_host[1] = -(double)d.mm_back[1];
      } else {
        double v[2];
        memcpy(v, d.mm_back, 16);
        s->mm_host[0] = v[0];
        s->mm_host[1] = -v[1];
      }
      HIP_TRY(hipEventElapsedTime(&s->coll_ms[0], d.ev[0], d.ev[1]));
      HIP_TRY(hipEventElapsedTime(&s->coll_ms[1], d.ev[2], d.ev[3]));
    } else {
      HIP_TRY(hipStreamSynchronize(st));
    }
    return XM_OK;
  };
  d.rc = run();
  if (d.rc) d.err = g_err;
  sharded_leave(s, d.rc);  // (an error return outside an agreement point must not leave the peers waiting at the next one)
}

void sharded_thread_main(xm_sharded* s, int g) {
  unsigned long long seen = 0;
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(s->mu);
      s->cv.wait(lk, [&] { return s->stop || s->gen != seen; });
      if (s->stop) return;
      seen = s->gen;
    }
    sharded_frame_on(s, g);
    {
      std::lock_guard<std::mutex> lk(s->mu);
      s->done += 1;
    }
    s->cv.notify_all();
  }
}

}  // namespace

extern "C" {

void xm_sharded_destroy(xm_sharded* s) {
  if (!s) return;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    s->stop = true;
  }
  s->cv.notify_all();
  for (auto& d : s->devs)
    if (d->th.joinable()) d->th.join();
  for (auto& d : s->devs) {
    (void)hipSetDevice(d->id);
    if (d->h) (void)xm_sync(d->h);
    if (d->comm && s->rccl.CommDestroy) (void)s->rccl.CommDestroy(d->comm);
    d->x.release(); d->y.release(); d->t.release(); d->p.release(); d->depth.release(); d->bgr.release();
    d->send.release(); d->gathered.release(); d->fake_tmp.release();
    if (d->fake_ptrs) (void)hipFree(d->fake_ptrs);
    if (d->frame16) (void)hipFree(d->frame16);
    if (d->key) (void)hipFree(d->key);
    if (d->mm) (void)hipFree(d->mm);
    for (auto& e : d->ev) if (e) (void)hipEventDestroy(e);
    if (d->h) xm_destroy(d->h);
  }
  delete s;
}

int xm_create_sharded(const int* dev_ids, int n_dev, const xm_config* cfg, xm_sharded** out) {
  if (!dev_ids || n_dev <= 0 || !cfg || !out) return fail(XM_ERR_INVALID, "bad argument");
  *out = nullptr;
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return fail(XM_ERR_HIP, "no AMD GPU visible");
  for (int i = 0; i < n_dev; ++i) {
    if (dev_ids[i] < 0 || dev_ids[i] >= have) return fail(XM_ERR_INVALID, "device %d is not one of the %d visible", dev_ids[i], have);
    for (int j = 0; j < i; ++j)
      if (dev_ids[j] == dev_ids[i]) return fail(XM_ERR_INVALID, "device %d is listed twice", dev_ids[i]);
  }
  xm_sharded* s = new (std::nothrow) xm_sharded();
  if (!s) return fail(XM_ERR_NOMEM, "out of host memory");
  std::vector<int> ids(dev_ids, dev_ids + n_dev);
  if (const char* fr = dbg_opt("XM_SHARD_FAKE_RANKS")) {  // tests: W virtual ranks on the one device (see the header comment)
    const int W = atoi(fr);
    if (n_dev != 1 || W < 1 || W > 64) {
      delete s;
      return fail(XM_ERR_INVALID, "XM_SHARD_FAKE_RANKS = %s needs n_dev == 1 and 1 <= W <= 64", fr);
    }
    s->fake = W > 1;
    ids.assign(W, dev_ids[0]);
    n_dev = W;
  }
  if (const char* fa = dbg_opt("XM_SHARD_FAIL_AT")) {
    if (sscanf(fa, "%d:%d", &s->fail_dev, &s->fail_point) != 2) s->fail_dev = -1;
  }
  dev_ids = ids.data();
  s->rccl = load_rccl();
  s->use_rccl = !s->fake && s->rccl.ok();
  if (n_dev > 1 && !s->use_rccl && !s->fake) {
    delete s;
    return fail(XM_ERR_INVALID, "librccl was not found: a sharded handle over %d devices needs it", n_dev);
  }
  int rc = XM_OK;
  for (int i = 0; i < n_dev && !rc; ++i) {
    s->devs.emplace_back(new xm_sharded::Dev());
    xm_sharded::Dev& d = *s->devs.back();
    d.id = dev_ids[i];
    xm_config c = *cfg;
    c.device = d.id;
    if ((rc = xm_create(&c, &d.h))) break;
    hipError_t e = hipSetDevice(d.id);
    if (e == hipSuccess) e = hipMalloc((void**)&d.key, d.h->key_cells * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc(&d.mm, 16);
    for (auto& ev : d.ev)
      if (e == hipSuccess) e = hipEventCreate(&ev);
    if (e == hipSuccess && s->fake) e = hipMalloc((void**)&d.fake_ptrs, sizeof(void*) * 64);
    if (e != hipSuccess) rc = fail(XM_ERR_HIP, "device %d: %s", d.id, hipGetErrorString(e));
  }
  if (!rc && s->use_rccl) {  // one communicator per device, all in this process
    std::vector<void*> comms(n_dev, nullptr);
    const int e = s->rccl.CommInitAll(comms.data(), n_dev, dev_ids);
    if (e) rc = fail(XM_ERR_HIP, "ncclCommInitAll failed: %s", s->rccl.GetErrorString ? s->rccl.GetErrorString(e) : "?");
    else
      for (int i = 0; i < n_dev; ++i) s->devs[i]->comm = comms[i];
  }
  if (rc) {
    const std::string keep = g_err;
    xm_sharded_destroy(s);
    return fail(rc, "%s", keep.c_str());
  }
  for (int i = 0; i < n_dev; ++i) s->devs[i]->th = std::thread(sharded_thread_main, s, i);
  *out = s;
  return XM_OK;
}

int xm_sharded_process_frame(xm_sharded* s, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p, size_t n, int t_dtype,
                             float* depth_out, uint8_t* bgr_out, xm_frame_stats* stats) {
  if (!s || (n && (!x || !y || !t))) return fail(XM_ERR_INVALID, "NULL argument");
  if (t_dtype != XM_T_INT64 && t_dtype != XM_T_FLOAT32 && t_dtype != XM_T_FLOAT64) return fail(XM_ERR_INVALID, "unknown t_dtype");
  if (n >= XM_KEY_MAX_EVENTS) return fail(XM_ERR_TOO_MANY, "a frame of %zu events exceeds the packed keys' 2^%d", n, XM_KEY_IDX_BITS);
  s->x = x; s->y = y; s->t = t; s->p = p; s->n = n; s->t_dtype = t_dtype;
  s->depth_out = depth_out;
  s->bgr_out = bgr_out;
  s->tag = s->tag >= 1000 ? 1 : s->tag + 1;  // (the key frames are cleared every frame: any tag in [1, 2^19) would do)
  const auto run_frame = [&]() -> int {
    {  // (the device threads are idle: the barrier starts the frame clean, whatever the last frame left in it)
      std::lock_guard<std::mutex> lk(s->bmu);
      s->b_arrived = 0;
      s->b_rc = 0;
      s->b_poison = 0;
    }
    {
      std::lock_guard<std::mutex> lk(s->mu);
      s->done = 0;
      s->gen += 1;
    }
    s->cv.notify_all();
    {
      std::unique_lock<std::mutex> lk(s->mu);
      s->cv.wait(lk, [&] { return s->done == (int)s->devs.size(); });
    }
    for (int pass = 0; pass < 2; ++pass)  // (the device that failed first, not the ones that stopped because of it)
      for (size_t i = 0; i < s->devs.size(); ++i) {
        auto& d = s->devs[i];
        if (d->rc && (pass == 1 || !d->peer_only)) return fail(d->rc, "device %d (index %zu): %s", d->id, i, d->err.c_str());
      }
    return XM_OK;
  };
  // Time-sorted int64 frames on rigs whose X-map is injective take the columns exchange (2-byte cells on the wire, no atomics,
  // no extrema pass); a frame one of whose pieces objects (a shard inside one column, events out of order ...) is redone with
  // the packed keys -- as is everything else.
  const char* force = dbg_opt("XM_SHARDED_KEYS");
  s->cols_now = t_dtype == XM_T_INT64 && !p && n >= 64 * s->devs.size() && !(force && force[0] == '1') &&
                (!s->use_rccl || s->rccl.AllGather) &&
                xm_shard_cols_info(s->devs[0]->h, n, &s->cols_frame_bytes, &s->cols_reduce_u32, &s->cols_send_bytes, &s->cols_cap) == XM_OK;
  int rc = run_frame();
  if (rc) return rc;
  if (s->cols_now) {
    bool flagged = false;
    for (auto& d : s->devs) flagged = flagged || d->flagged;
    s->frames_columns += 1;
    if (flagged) {
      s->frames_redone += 1;
      s->cols_now = false;
      if ((rc = run_frame())) return rc;
    }
  } else {
    s->frames_keys += 1;
  }
  if (stats) {
    memset(stats, 0, sizeof *stats);
    stats->n_events = n;
    stats->n_used = n;
    stats->t_min = n ? s->mm_host[0] : 0.0;
    stats->t_max = n ? s->mm_host[1] : 0.0;
    stats->gpu_ms[0] = s->coll_ms[0];  // the two all-reduces on device 0's stream (HIP events around them)
    stats->gpu_ms[1] = s->coll_ms[1];
  }
  return XM_OK;
}

int xm_sharded_stats(xm_sharded* s, uint64_t* frames_columns, uint64_t* frames_keys, uint64_t* frames_redone) {
  if (!s) return fail(XM_ERR_INVALID, "NULL argument");
  if (frames_columns) *frames_columns = s->frames_columns;
  if (frames_keys) *frames_keys = s->frames_keys;
  if (frames_redone) *frames_redone = s->frames_redone;
  return XM_OK;
}

int xm_sharded_info(xm_sharded* s, int* n_dev, int* uses_rccl, uint64_t* key_frame_bytes) {
  if (!s) return fail(XM_ERR_INVALID, "NULL argument");
  if (n_dev) *n_dev = (int)s->devs.size();
  if (uses_rccl) *uses_rccl = s->use_rccl ? 1 : 0;
  if (key_frame_bytes) *key_frame_bytes = (uint64_t)s->devs[0]->h->key_cells * 8;
  return XM_OK;
}

}  // extern "C"
