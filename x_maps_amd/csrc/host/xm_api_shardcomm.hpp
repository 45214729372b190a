// xm_api_shardcomm.hpp -- C-ABI: one rank of a frame sharded over several PROCESSES (one per GPU), the library driving RCCL itself
// (the partitioning and the exchange; the multi-process sibling of xm_create_sharded)
// (part of libxmaps_hip.so's host side: included by ../xmaps_hip.hip, one translation unit; see that file for the order)
//
//   rank 0:      xm_shard_comm_id(id)                 -- 128 opaque bytes; the host hands them to every rank by whatever it has
//                                                         (MPI_Bcast, a file, torch.distributed: x_maps_amd.sharded.ShardComm)
//   every rank:  xm_shard_comm_create(h, id, rank, world, n_frame_events, &c)     -- collective (ncclCommInitRank on h's device)
//   per frame:   xm_shard_comm_frame(c, x, y, t, n_own, depth, bgr)               -- ONE call, everything on h's stream, asynchronous:
//                  pack -> ncclAllGather(headers + last events) -> prepare + boundary pass + column-tile K1
//                       -> ncclAllReduce(SUM, the u16 frame as int32 pairs) -> frame kernel              (the "columns" merge)
//                xm_shard_comm_frame_keys(c, x, y, t, p, n_own, first_index, depth, bgr)                  -- the packed keys:
//                  extrema -> ncclAllReduce(MIN) -> clear + scatter with global indices -> ncclAllReduce(MAX, uint64) -> frame kernel
//                (any rig, any event order, polarity column; also the redo of frames the columns merge flagged)
//   now and then: xm_shard_comm_failed(c, &failed)    -- collective + synchronises: did ANY rank flag a columns frame?
// What a host thread pays per frame is the enqueue of ~8 launches (~15 us) instead of five Python calls and two torch.distributed
// collectives (~64 us): the frame loop is GPU-bound again, and a C / C++ host (one process per GPU) needs no Python at all.
#pragma once

struct xm_shard_comm {
  xm_handle* h = nullptr;
  RcclApi rccl;
  void* comm = nullptr;
  int rank = 0, world = 1;
  uint64_t n_frame = 0;
  bool cols = false;               // the rig / density takes the column tiles
  size_t cap = 0, send_bytes = 0, reduce_u32 = 0, frame_bytes = 0;
  unsigned char *send = nullptr, *gathered = nullptr;
  uint16_t* frame16 = nullptr;
  uint64_t* key = nullptr;         // packed keys (allocated on first use)
  void* mm = nullptr;              // {tmin, -tmax}
  int* flag = nullptr;             // the ranks' verdicts (device)
  u32 tag = 0;
};

extern "C" {

int xm_shard_comm_id(void* id_out) {
  if (!id_out) return fail(XM_ERR_INVALID, "NULL argument");
  RcclApi r = load_rccl();
  if (!r.ok_ranks()) return fail(XM_ERR_INVALID, "librccl (ncclGetUniqueId / ncclCommInitRank / ncclAllGather) was not found");
  RcclApi::UniqueId id;
  const int e = r.GetUniqueId(&id);
  if (e) return fail(XM_ERR_HIP, "ncclGetUniqueId failed: %s", r.err(e));
  memcpy(id_out, id.bytes, sizeof id.bytes);
  return XM_OK;
}

void xm_shard_comm_destroy(xm_shard_comm* c) {
  if (!c) return;
  if (c->h) {
    (void)hipSetDevice(c->h->cfg.device);
    (void)xm_sync(c->h);
  }
  if (c->comm && c->rccl.CommDestroy) (void)c->rccl.CommDestroy(c->comm);
  if (c->send) (void)hipFree(c->send);
  if (c->gathered) (void)hipFree(c->gathered);
  if (c->frame16) (void)hipFree(c->frame16);
  if (c->key) (void)hipFree(c->key);
  if (c->mm) (void)hipFree(c->mm);
  if (c->flag) (void)hipFree(c->flag);
  delete c;
}

int xm_shard_comm_create(xm_handle* h, const void* id, int rank, int world, uint64_t n_frame_events, xm_shard_comm** out) {
  if (!h || !id || !out) return fail(XM_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return fail(XM_ERR_INVALID, "bad rank / world");
  XM_ENTER(h);
  xm_shard_comm* c = new (std::nothrow) xm_shard_comm();
  if (!c) return fail(XM_ERR_NOMEM, "out of host memory");
  c->h = h;
  c->rank = rank;
  c->world = world;
  c->n_frame = n_frame_events;
  c->rccl = load_rccl();
  const auto bail = [&](int rc) {
    const std::string keep = g_err;
    xm_shard_comm_destroy(c);
    return fail(rc, "%s", keep.c_str());
  };
  if (!c->rccl.ok_ranks()) {
    (void)fail(XM_ERR_INVALID, "librccl (ncclGetUniqueId / ncclCommInitRank / ncclAllGather) was not found");
    return bail(XM_ERR_INVALID);
  }
  c->cols = xm_shard_cols_info(h, n_frame_events, &c->frame_bytes, &c->reduce_u32, &c->send_bytes, &c->cap) == XM_OK;
  hipError_t e = hipSuccess;
  if (c->cols) {
    e = hipMalloc((void**)&c->send, c->send_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&c->gathered, c->send_bytes * (size_t)world);
    if (e == hipSuccess) e = hipMalloc((void**)&c->frame16, c->frame_bytes);
    if (e == hipSuccess) e = hipMemset(c->send, 0, c->send_bytes);
    if (e == hipSuccess) e = hipMemset(c->frame16, 0, c->frame_bytes);
  }
  if (e == hipSuccess) e = hipMalloc(&c->mm, 16);
  if (e == hipSuccess) e = hipMalloc((void**)&c->flag, sizeof(int));
  if (e != hipSuccess) {
    (void)fail(XM_ERR_HIP, "device buffers of the shard communicator: %s", hipGetErrorString(e));
    return bail(XM_ERR_HIP);
  }
  RcclApi::UniqueId uid;
  memcpy(uid.bytes, id, sizeof uid.bytes);
  const int ne = c->rccl.CommInitRank(&c->comm, world, uid, rank);  // (collective: every rank of the id is in here now)
  if (ne) {
    c->comm = nullptr;
    (void)fail(XM_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, c->rccl.err(ne));
    return bail(XM_ERR_HIP);
  }
  *out = c;
  return XM_OK;
}

int xm_shard_comm_info(xm_shard_comm* c, int* takes_columns, size_t* cap_events, size_t* send_bytes, size_t* frame_bytes) {
  if (!c) return fail(XM_ERR_INVALID, "NULL argument");
  if (takes_columns) *takes_columns = c->cols ? 1 : 0;
  if (cap_events) *cap_events = c->cap;
  if (send_bytes) *send_bytes = c->send_bytes;
  if (frame_bytes) *frame_bytes = c->cols ? c->reduce_u32 * 4 : (size_t)c->h->key_cells * 8;
  return XM_OK;
}

int xm_shard_comm_frame(xm_shard_comm* c, uint16_t* x, uint16_t* y, int64_t* t, size_t n_own, float* depth_out, uint8_t* bgr_out) {
  if (!c) return fail(XM_ERR_INVALID, "NULL argument");
  if (!c->cols) return fail(XM_ERR_INVALID, "this rig / frame density does not take the column tiles: xm_shard_comm_frame_keys");
  xm_handle* h = c->h;
  hipStream_t st = h->slots[0].stream;
  int rc, e;
  if ((rc = xm_shard_cols_pack(h, x, y, t, n_own, c->send, c->cap))) return rc;
  if ((e = c->rccl.AllGather(c->send, c->gathered, c->send_bytes, RcclApi::Uint8, c->comm, st)))
    return fail(XM_ERR_HIP, "ncclAllGather(headers + last events) failed: %s", c->rccl.err(e));
  if ((rc = xm_shard_cols_scatter(h, x, y, t, n_own, c->n_frame, c->gathered, c->send_bytes, c->rank, c->world, c->cap, c->frame16))) return rc;
  if ((e = c->rccl.AllReduce(c->frame16, c->frame16, c->reduce_u32, RcclApi::Int32, RcclApi::Sum, c->comm, st)))
    return fail(XM_ERR_HIP, "ncclAllReduce(SUM, u16 frame) failed: %s", c->rccl.err(e));
  if (depth_out || bgr_out)
    if ((rc = xm_shard_finish_u16(h, c->frame16, depth_out, bgr_out))) return rc;
  return XM_OK;
}

int xm_shard_comm_frame_keys(xm_shard_comm* c, const uint16_t* x, const uint16_t* y, const void* t, const int16_t* p, size_t n_own,
                             int t_dtype, uint64_t first_index, float* depth_out, uint8_t* bgr_out) {
  if (!c) return fail(XM_ERR_INVALID, "NULL argument");
  xm_handle* h = c->h;
  XM_ENTER(h);
  hipStream_t st = h->slots[0].stream;
  if (!c->key) HIP_TRY(hipMalloc((void**)&c->key, (size_t)h->key_cells * sizeof(uint64_t)));
  c->tag = c->tag >= 1000 ? 1 : c->tag + 1;  // (the key frame is cleared every frame: any tag in [1, 2^19) would do)
  int rc, e;
  if ((rc = xm_shard_minmax_device(h, t, p, n_own, t_dtype, c->mm))) return rc;
  if ((e = c->rccl.AllReduce(c->mm, c->mm, 2, t_dtype == XM_T_INT64 ? RcclApi::Int64 : RcclApi::Float64, RcclApi::Min, c->comm, st)))
    return fail(XM_ERR_HIP, "ncclAllReduce(MIN, extrema) failed: %s", c->rccl.err(e));
  if ((rc = xm_shard_clear(h, c->key))) return rc;
  if ((rc = xm_shard_scatter_device(h, x, y, t, p, n_own, t_dtype, first_index, c->mm, c->tag, c->key))) return rc;
  if ((e = c->rccl.AllReduce(c->key, c->key, h->key_cells, RcclApi::Uint64, RcclApi::Max, c->comm, st)))
    return fail(XM_ERR_HIP, "ncclAllReduce(MAX, key frame) failed: %s", c->rccl.err(e));
  if (depth_out || bgr_out)
    if ((rc = xm_shard_finish(h, c->key, c->tag, depth_out, bgr_out))) return rc;
  return XM_OK;
}

// did ANY rank flag a columns frame since the last call?  Collective (every rank calls it at the same point of its frame
// sequence); synchronises the handle's stream.
int xm_shard_comm_failed(xm_shard_comm* c, int* failed) {
  if (!c || !failed) return fail(XM_ERR_INVALID, "NULL argument");
  xm_handle* h = c->h;
  int mine = 0, rc;
  if ((rc = xm_shard_cols_failed(h, &mine))) return rc;
  hipStream_t st = h->slots[0].stream;
  HIP_TRY(hipMemcpyAsync(c->flag, &mine, sizeof mine, hipMemcpyHostToDevice, st));
  const int e = c->rccl.AllReduce(c->flag, c->flag, 1, RcclApi::Int32, RcclApi::Max, c->comm, st);
  if (e) return fail(XM_ERR_HIP, "ncclAllReduce(MAX, verdicts) failed: %s", c->rccl.err(e));
  int any = 0;
  HIP_TRY(hipMemcpyAsync(&any, c->flag, sizeof any, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  *failed = any;
  return XM_OK;
}

}  // extern "C"
