"""One frame's event buffer sharded by index over the GPUs of a node.

    rank g owns events [g*N/W, (g+1)*N/W); tables are replicated; every rank scatters its shard into a
    private full-size packed-key frame; ONE all-reduce (MAX on int64) of that frame over RCCL/xGMI merges
    the shards; the frame kernel (dilate o remap -> depth -> BGR) then runs on the reduced keys.

Why MAX of packed keys and not "min of the depth frame": the reference's collision rule is NumPy's
last-writer-wins (python/cam_proj_calibration.py:299-303).  key = tag<<44 | GLOBAL event index<<16 | disp,
so the element-wise maximum over shards is exactly the event with the largest global index -- bit-exact
with the single-GPU frame, which min-of-depth (nearest-surface-wins) is not.  Bit 63 of a key is always 0,
so a signed int64 MAX (what torch/RCCL offer) orders keys like the unsigned compare on the device.

The only other coupling is (tmin, tmax) of the whole frame (python/x_maps_disparity.py:12-13): a 16-byte
MIN all-reduce of (tmin, -tmax).

The collective is torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).  Nothing on the frame's
critical path visits the host: the shard's extrema are left in a 16-byte device tensor as {tmin, -tmax}
(xm_shard_minmax_device), that tensor is MIN-all-reduced in place, the scatter kernel reads it from device memory
(xm_shard_scatter_device), the key frame is MAX-all-reduced in place and the frame kernel runs on it -- five enqueues on one
stream, zero synchronisations.  Compute is behind a small provider protocol so that the sharding logic itself is testable
without a GPU:
  provider.new_minmax_buffer(shard)          -> 2-element tensor (int64 for int64 t, float64 for float t)
  provider.minmax_into(shard, mm)            mm <- {tmin, -tmax} of the shard ({+max, +max} when empty)
  provider.scatter(shard, idx_offset, mm, tag, key_frame)   mm = the FRAME's {tmin, -tmax} after the MIN-reduce (in place)
  provider.finish(key_frame, tag)            -> (depth, bgr)
`GpuShardProvider` is the product implementation (C-ABI xm_shard_* on device tensors).
"""
from __future__ import annotations

import numpy as np

KEY_MAX_TAG = (1 << 19) - 1


def shard_bounds(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Index range of rank's shard: [rank*N/W, (rank+1)*N/W) with integer arithmetic."""
    return (rank * n_total) // world, ((rank + 1) * n_total) // world


def resident_with_headroom(torch, device, shard, cap_events):
    """The shard's columns (x, y, t device tensors; no polarity column, int64 stamps) once more with `cap_events` + 8 events of
    headroom in front -- the predecessor's last column is copied there on the device -- and 8 of slack behind for the last 16-byte
    load: the layout a host that keeps its shards resident for the columns exchange allocates in the first place."""
    x, y, t, p = shard
    assert p is None and t.dtype == torch.int64
    out = []
    for a in (x, y, t):
        buf = torch.zeros(cap_events + 8 + len(a) + 8, dtype=a.dtype, device=device)
        buf[cap_events + 8:cap_events + 8 + len(a)].copy_(a)
        out.append(buf)
    torch.cuda.current_stream(device).synchronize()
    return tuple(out)


def first_own_event(buf, cap_events):
    """address of the first own event in a resident_with_headroom() buffer (an empty slice has no data_ptr)"""
    return buf.data_ptr() + (cap_events + 8) * buf.element_size()


class GpuShardProvider:
    """Shard compute on one MI355X through the C-ABI.  Event columns are torch CUDA tensors."""

    def __init__(self, engine, device):
        import torch
        self.torch = torch
        self.eng = engine
        self.device = device
        self.stream = torch.cuda.ExternalStream(engine.stream(0), device=device)

    def new_key_frame(self, pad_to: int = 1):
        """int64 key frame, flat, padded with zero cells to a multiple of `pad_to` (the reduce-scatter wants equal chunks); the
        engine only ever touches the first key_shape[0] * key_shape[1] cells."""
        cells = self.eng.key_shape[0] * self.eng.key_shape[1]
        padded = (cells + pad_to - 1) // pad_to * pad_to
        kf = self.torch.zeros(padded, dtype=self.torch.int64, device=self.device)
        self.torch.cuda.current_stream(self.device).synchronize()
        return kf.view(self.eng.key_shape) if padded == cells and pad_to == 1 else kf

    def decode_u16(self, key_chunk, tag, out_chunk):
        self.eng.shard_decode_u16(key_chunk.data_ptr(), key_chunk.numel(), tag, out_chunk.data_ptr())

    def _zeros(self, n, dtype):
        """torch fills on ITS current stream; the engine's streams are non-blocking ones, so the fill is waited for here --
        otherwise it can land after a kernel of the engine has already written the buffer (seen once: a shard's extrema
        overwritten with zeros)."""
        z = self.torch.zeros(n, dtype=dtype, device=self.device)
        self.torch.cuda.current_stream(self.device).synchronize()
        return z

    def new_u16(self, n):
        return self._zeros(n, self.torch.int16)

    def finish_u16(self, disp_frame, want_bgr=True):
        torch = self.torch
        depth = torch.empty((self.eng.out_h, self.eng.out_w), dtype=torch.float32, device=self.device)
        bgr = torch.empty((self.eng.out_h, self.eng.out_w, 3), dtype=torch.uint8, device=self.device) if want_bgr else None
        self.eng.shard_finish_u16(disp_frame.data_ptr(), depth.data_ptr(), None if bgr is None else bgr.data_ptr())
        return depth, bgr

    def frame_lines(self):
        """(lines, cells per line) of the key frame along its slow axis: the device frame is column-major, a line = a frame column"""
        return self.eng.rect_w, self.eng.rect_h

    def halo_lines(self):
        """Frame columns a band-sharded rank needs from either neighbour: the widest tile patch + 1 (-1: no band can be cut)."""
        c = self.eng.k2_patch_cols_max()
        return -1 if c < 0 or self.eng.camera_perspective else c + 1

    def finish_u16_band(self, disp_frame, col_lo, col_hi, want_bgr=True):
        """Partial projector frame: the tiles centred on frame columns [col_lo, col_hi); zeros elsewhere."""
        torch = self.torch
        depth = self._zeros(self.eng.out_h * self.eng.out_w, torch.float32).view(self.eng.out_h, self.eng.out_w)
        bgr = self._zeros(self.eng.out_h * self.eng.out_w * 3, torch.uint8).view(self.eng.out_h, self.eng.out_w, 3) if want_bgr else None
        self.eng.shard_finish_u16_band(disp_frame.data_ptr(), col_lo, col_hi, depth.data_ptr(), None if bgr is None else bgr.data_ptr())
        return depth, bgr

    def clear_key_frame(self, kf):
        self.eng.shard_clear(kf.data_ptr())

    def new_minmax_buffer(self, shard):
        t = shard[2]
        dt = self.torch.int64 if t.dtype == self.torch.int64 else self.torch.float64
        return self._zeros(2, dt)

    @staticmethod
    def _t_dtype(t):
        import torch
        return {torch.int64: 0, torch.float32: 1, torch.float64: 2}[t.dtype]

    def minmax_into(self, shard, mm):
        x, y, t, p = shard
        self.eng.shard_minmax_device(t.data_ptr() if len(t) else None, None if p is None else p.data_ptr(), len(t),
                                     mm.data_ptr(), t_dtype=self._t_dtype(t))

    def scatter(self, shard, idx_offset, mm, tag, key_frame):
        x, y, t, p = shard
        if len(t):
            self.eng.shard_scatter_device(x.data_ptr(), y.data_ptr(), t.data_ptr(), None if p is None else p.data_ptr(),
                                          len(t), idx_offset, mm.data_ptr(), tag, key_frame.data_ptr(),
                                          t_dtype=self._t_dtype(t))

    def finish(self, key_frame, tag, want_bgr=True):
        torch = self.torch
        depth = torch.empty((self.eng.out_h, self.eng.out_w), dtype=torch.float32, device=self.device)
        bgr = torch.empty((self.eng.out_h, self.eng.out_w, 3), dtype=torch.uint8, device=self.device) if want_bgr else None
        self.eng.shard_finish(key_frame.data_ptr(), tag, depth.data_ptr(), None if bgr is None else bgr.data_ptr())
        return depth, bgr

    # ---- merge = "columns": every time column on one rank, plain u16 frames merged by SUM (xm_shard_cols_*) ----
    def cols_setup(self, n_frame_events):
        """buffers for frames of that many events, or None when this rig / density does not take the column tiles"""
        info = self.eng.shard_cols_info(n_frame_events)
        if info is None:
            return None
        torch = self.torch
        info["frame"] = self._zeros(info["frame_bytes"], torch.uint8)
        info["send"] = self._zeros(info["send_bytes"], torch.uint8)
        return info

    def cols_resident(self, shard, cap_events):
        return resident_with_headroom(self.torch, self.device, shard, cap_events)

    _own = staticmethod(lambda buf, cap: first_own_event(buf, cap))

    def cols_pack(self, res, n, cap, send):
        self.eng.shard_cols_pack(self._own(res[0], cap), self._own(res[1], cap), self._own(res[2], cap), n, send.data_ptr(), cap)

    def cols_scatter(self, res, n, cap, n_frame, gathered, send_bytes, rank, world, frame):
        self.eng.shard_cols_scatter(self._own(res[0], cap), self._own(res[1], cap), self._own(res[2], cap), n, n_frame, gathered.data_ptr(),
                                    send_bytes, rank, world, cap, frame.data_ptr())

    def cols_failed(self):
        return self.eng.shard_cols_failed()

    def cols_finish(self, frame, want_bgr=True):
        return self.finish_u16(frame, want_bgr)

    def as_tensor(self, a):
        return a

    def collective_stream(self):
        return self.torch.cuda.stream(self.stream)  # collectives are ordered on the engine's own stream


class ShardedFrameProcessor:
    """Drives one rank of the sharded frame.  `dist` = torch.distributed (already initialised).
    always_reduce: issue the two all-reduces even when world_size == 1 (exercises the RCCL path on a single-GPU box;
    a one-rank all-reduce leaves the data unchanged)."""

    def __init__(self, provider, dist, group=None, always_reduce=False, merge="all_reduce"):
        """merge: "all_reduce" -- MAX all-reduce of the whole 8-byte key frame (2 (W-1)/W x 8 bytes per cell and rank);
        "reduce_scatter" -- MAX reduce-scatter of the key frame, the own chunk decoded to u16 disparities, all-gather of
        the u16 chunks, frame kernel on the plain disparity frame ((W-1)/W x (8 + 2) bytes per cell): 37 % less traffic,
        and the frame kernel reads 2 instead of 8 bytes per cell;
        "bands" -- the same reduce-scatter, but the u16 frame is never gathered: every rank keeps its band of frame columns, gets a
        halo of a few columns from either neighbour (point to point), finishes the projector tiles centred on its band, and the
        partial projector frames are MAX-all-reduced (K2 sharded as well; what crosses the links after the
        reduce-scatter is 2 halos + the projector frame instead of the whole disparity frame).  Projector view; falls back to
        "reduce_scatter" when a band is narrower than the halo."""
        assert merge in ("all_reduce", "reduce_scatter", "bands", "columns")
        self.p = provider
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.merge = merge
        self.cols = None  # merge == "columns": set up by process_shard_columns for the frame size at hand
        if merge == "columns":
            merge = self.merge = "columns"
            self.key_frame = None
            self.always_reduce = always_reduce
            self.mm = None
            self._mm_for = None
            self.tag = 0
            self.collectives_issued = 0
            self.collective_bytes_per_frame = None
            return
        self.key_frame = provider.new_key_frame(self.world) if merge != "all_reduce" else provider.new_key_frame()
        if merge != "all_reduce":
            chunk = self.key_frame.numel() // self.world
            self.kf_chunk = self.key_frame.new_zeros(chunk)
            self.u16_chunk = provider.new_u16(chunk)
            self.u16_full = provider.new_u16(chunk * self.world)
        self.always_reduce = always_reduce
        self.mm = None
        self._mm_for = None
        self.tag = 0
        self.collectives_issued = 0

    def _next_tag(self):
        if self.tag >= KEY_MAX_TAG:
            self.p.clear_key_frame(self.key_frame)
            self.tag = 0
        self.tag += 1
        return self.tag

    def _all_reduce(self, tensor, op):
        if self.world > 1 or self.always_reduce:
            self.dist.all_reduce(self.p.as_tensor(tensor), op=op, group=self.group)
            self.collectives_issued += 1

    def process_shard(self, shard, idx_offset: int, want_bgr=True, finish_on_all_ranks=True):
        """shard = (x, y, t, p|None) of THIS rank's events; idx_offset = global index of its first event.
        Returns (depth, bgr) of the whole frame (on every rank, or only rank 0).  Asynchronous on the provider's stream."""
        import contextlib
        tag = self._next_tag()
        if self.mm is None or self._mm_for != shard[2].dtype:
            self.mm = self.p.new_minmax_buffer(shard)
            self._mm_for = shard[2].dtype
        ctx = self.p.collective_stream() if hasattr(self.p, "collective_stream") else contextlib.nullcontext()
        with ctx:
            # 1. frame extrema: MIN-reduce {tmin, -tmax}, on the device
            self.p.minmax_into(shard, self.mm)
            self._all_reduce(self.mm, self.dist.ReduceOp.MIN)
            # 2. private scatter (reads the reduced extrema from device memory), 3. merge
            self.p.scatter(shard, idx_offset, self.mm, tag, self.key_frame)
            if self.merge == "bands" and self._bands_ok():
                self._reduce_scatter_max(self.key_frame, self.kf_chunk)
                self.p.decode_u16(self.kf_chunk, tag, self.u16_chunk)
                return self._finish_bands(want_bgr)
            if self.merge != "all_reduce":
                self._reduce_scatter_max(self.key_frame, self.kf_chunk)
                self.p.decode_u16(self.kf_chunk, tag, self.u16_chunk)
                self._all_gather(self.u16_full, self.u16_chunk)
                if finish_on_all_ranks or self.rank == 0:
                    return self.p.finish_u16(self.u16_full, want_bgr)
                return None, None
            self._all_reduce(self.key_frame, self.dist.ReduceOp.MAX)
            # 4. frame kernel on the merged keys
            if finish_on_all_ranks or self.rank == 0:
                return self.p.finish(self.key_frame, tag, want_bgr)
        return None, None

    # ---- merge = "columns" ----------------------------------------------------------------------------------------------
    def columns_resident(self, shard, n_frame_events):
        """Prepare THIS rank's shard for merge="columns": (resident, n_own) to hand to process_shard_columns, or None when the rig /
        the frame density does not take the column tiles (use another merge).  The resident copy has headroom in front of the
        events for the predecessor's last column -- the layout a host that keeps its shards in HBM allocates in the first place."""
        if self.cols is None or self.cols.get("n_frame") != n_frame_events:
            info = self.p.cols_setup(n_frame_events)
            if info is None:
                return None
            info["n_frame"] = n_frame_events
            info["gathered"] = info["send"].new_zeros(info["send_bytes"] * self.world)
            self.cols = info
            # what crosses the links per frame and rank: the gathered headers + last events, the u16 frame
            self.collective_bytes_per_frame = {"last_events_all_gather": info["send_bytes"] * self.world,
                                               "u16_frame_sum_all_reduce": info["reduce_u32"] * 4}
        return self.p.cols_resident(shard, self.cols["cap_events"]), len(shard[2])

    def process_shard_columns(self, resident, n_own, want_bgr=True, finish_on_all_ranks=True):
        """One frame, this rank's shard as columns_resident() returned it.  Every time column ends up on one rank (each rank but
        the last hands its last column's events to its successor), the plain u16 frames are disjoint and merge by SUM: 2 bytes
        per cell on the wire, no packed keys, no atomics, no extrema pass.  Asynchronous; columns_failed() (a synchronisation +
        a one-word all-reduce) tells whether any frame since the last check has to be redone with the packed keys."""
        import contextlib
        import torch
        c = self.cols
        ctx = self.p.collective_stream() if hasattr(self.p, "collective_stream") else contextlib.nullcontext()
        with ctx:
            self.p.cols_pack(resident, n_own, c["cap_events"], c["send"])
            self._all_gather(c["gathered"], c["send"])  # the frame's extrema and every predecessor's last events in ONE collective
            self.p.cols_scatter(resident, n_own, c["cap_events"], c["n_frame"], c["gathered"], c["send_bytes"], self.rank, self.world, c["frame"])
            if "red" not in c:  # (the view is built once: per frame it is a few microseconds of host time, and this loop is host-bound)
                c["red"] = self.p.as_tensor(c["frame"])[:c["reduce_u32"] * 4].view(torch.int32)
            self._all_reduce(c["red"], self.dist.ReduceOp.SUM)  # (disjoint cells: SUM of the packed pairs = the union)
            if finish_on_all_ranks or self.rank == 0:
                return self.p.cols_finish(c["frame"], want_bgr)
        return None, None

    def columns_failed(self):
        """did any rank object to any frame since the last call?  (synchronises; the frames in question are redone with another merge)"""
        import torch
        f = torch.tensor([1 if self.p.cols_failed() else 0], dtype=torch.int32)
        dev = getattr(self.p, "device", None)
        if dev is not None:
            f = f.to(dev)
        if self.world > 1 or self.always_reduce:
            self.dist.all_reduce(f, op=self.dist.ReduceOp.MAX, group=self.group)
        return bool(int(f.item()))

    # ---- merge = "bands" ------------------------------------------------------------------------------------------------
    def _frame_geometry(self):
        n_lines, line_len = self.p.frame_lines()  # the key frame's slow axis (device: frame columns of rect_h cells)
        return n_lines, line_len, self.kf_chunk.numel()

    def _bands_ok(self):
        hl = self.p.halo_lines() if hasattr(self.p, "halo_lines") else -1
        if hl < 0:
            return False
        n_lines, line_len, chunk = self._frame_geometry()
        return hl * line_len <= chunk  # a neighbour's band must hold the whole halo

    def _finish_bands(self, want_bgr):
        import torch
        n_lines, line_len, C = self._frame_geometry()
        r, W = self.rank, self.world
        Hc = self.p.halo_lines() * line_len
        full = self.u16_full  # [W * C] cells; only [r C - Hc, (r + 1) C + Hc) will be valid
        full[r * C:(r + 1) * C].copy_(self.u16_chunk)
        ops, keep = [], []
        as_u8 = lambda a: self.p.as_tensor(a).view(torch.uint8)  # (bytes on the wire: gloo has no int16)
        if W > 1:
            if r > 0:       # my first cells are the left neighbour's right halo; its last cells are my left halo
                send_l = self.u16_chunk[:Hc].contiguous()
                keep.append(send_l)
                ops.append(self.dist.P2POp(self.dist.isend, as_u8(send_l), self._peer(r - 1), group=self.group))
                ops.append(self.dist.P2POp(self.dist.irecv, as_u8(full[r * C - Hc:r * C]), self._peer(r - 1), group=self.group))
            if r < W - 1:
                send_r = self.u16_chunk[C - Hc:].contiguous()
                keep.append(send_r)
                ops.append(self.dist.P2POp(self.dist.isend, as_u8(send_r), self._peer(r + 1), group=self.group))
                ops.append(self.dist.P2POp(self.dist.irecv, as_u8(full[(r + 1) * C:(r + 1) * C + Hc]), self._peer(r + 1), group=self.group))
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
            self.collectives_issued += 1
        lo = -(-(r * C) // line_len)  # the lines that START inside this rank's chunk
        hi = min(-(-((r + 1) * C) // line_len), n_lines) if r < W - 1 else n_lines
        depth, bgr = self.p.finish_u16_band(full, lo, max(hi, lo + 1), want_bgr)
        # non-owners hold zeros: MAX assembles the frame (depth >= 0 orders like its int32 bits)
        self._all_reduce(self.p.as_tensor(depth).view(torch.int32), self.dist.ReduceOp.MAX)
        if bgr is not None:
            self._all_reduce(bgr, self.dist.ReduceOp.MAX)
        return depth, bgr

    def _peer(self, group_rank):
        return self.dist.get_global_rank(self.group, group_rank) if self.group is not None else group_rank

    def _reduce_scatter_max(self, full, chunk):
        """chunk <- MAX over ranks of full[rank * len(chunk) : (rank + 1) * len(chunk)]."""
        n = chunk.numel()
        if self.world > 1 or self.always_reduce:
            try:
                self.dist.reduce_scatter_tensor(self.p.as_tensor(chunk), self.p.as_tensor(full), op=self.dist.ReduceOp.MAX,
                                                group=self.group)
            except (RuntimeError, NotImplementedError):  # backends without reduce-scatter (gloo): all-reduce, keep the own chunk
                self.dist.all_reduce(self.p.as_tensor(full), op=self.dist.ReduceOp.MAX, group=self.group)
                chunk.copy_(full[self.rank * n:(self.rank + 1) * n])
            self.collectives_issued += 1
        else:
            chunk.copy_(full[:n])

    def _all_gather(self, full, chunk):
        if self.world > 1 or self.always_reduce:
            import torch  # bytes on the wire: every backend gathers uint8 (gloo has no int16)
            self.dist.all_gather_into_tensor(self.p.as_tensor(full).view(torch.uint8), self.p.as_tensor(chunk).view(torch.uint8),
                                             group=self.group)
            self.collectives_issued += 1
        else:
            full[:chunk.numel()].copy_(chunk)


class ShardedDevices:
    """One frame sharded over several GPUs of THIS process through the C-ABI's own entry (xm_create_sharded: one host thread and
    one RCCL communicator per device, owned by the handle) -- what a host that is not Python / torch.distributed binds.
    Host event columns in, depth / BGR out; synchronous.

        with ShardedDevices(tables, devices=[0, 1, 2, 3]) as sh:
            depth, bgr, stats = sh.process_frame(x, y, t)
    """

    def __init__(self, tables: dict, devices=(0,), camera_perspective: bool = False):
        import ctypes as C

        from . import _native as N
        from .engine import make_config
        self._C, self._N = C, N
        self._lib = N.load_library()
        cfg, keep = make_config(tables, camera_perspective, 0, 1, 0)
        ids = (C.c_int * len(devices))(*[int(d) for d in devices])
        self._s = C.c_void_p(None)
        N.check(self._lib.xm_create_sharded(ids, len(devices), C.byref(cfg), C.byref(self._s)))
        del keep
        mapx = np.asarray(tables["cam_mapx_i16"])
        pm = tables.get("disp_proj_mapxy_i16")
        self.out_h, self.out_w = mapx.shape if camera_perspective else np.asarray(pm).shape[:2]
        nd, rc, kb = C.c_int(0), C.c_int(0), C.c_uint64(0)
        N.check(self._lib.xm_sharded_info(self._s, C.byref(nd), C.byref(rc), C.byref(kb)))
        self.n_dev, self.uses_rccl, self.key_frame_bytes = int(nd.value), bool(rc.value), int(kb.value)

    def process_frame(self, x, y, t, p=None, want_depth=True, want_bgr=True):
        C, N = self._C, self._N
        x = np.ascontiguousarray(x, dtype=np.uint16)
        y = np.ascontiguousarray(y, dtype=np.uint16)
        t = np.ascontiguousarray(t)
        td = {np.dtype(np.int64): N.XM_T_INT64, np.dtype(np.float32): N.XM_T_FLOAT32, np.dtype(np.float64): N.XM_T_FLOAT64}[t.dtype]
        p = None if p is None else np.ascontiguousarray(p, dtype=np.int16)
        depth = np.empty((self.out_h, self.out_w), np.float32) if want_depth else None
        bgr = np.empty((self.out_h, self.out_w, 3), np.uint8) if want_bgr else None
        st = N.xm_frame_stats()
        ptr = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
        N.check(self._lib.xm_sharded_process_frame(self._s, ptr(x), ptr(y), ptr(t), ptr(p), len(x), td, ptr(depth), ptr(bgr), C.byref(st)))
        return depth, bgr, {"n_events": int(st.n_events), "t_min": float(st.t_min), "t_max": float(st.t_max),
                            "extrema_all_reduce_ms": float(st.gpu_ms[0]), "key_frame_all_reduce_ms": float(st.gpu_ms[1])}

    def stats(self) -> dict:
        """which exchange the frames so far took (columns / packed keys / columns redone with the keys)"""
        C = self._C
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._N.check(self._lib.xm_sharded_stats(self._s, C.byref(a), C.byref(b), C.byref(c)))
        return {"frames_columns": int(a.value), "frames_keys": int(b.value), "frames_redone": int(c.value)}

    def close(self):
        if getattr(self, "_s", None) is not None and self._s.value:
            self._lib.xm_sharded_destroy(self._s)
            self._s = self._C.c_void_p(None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardComm:
    """One rank of a frame sharded over several PROCESSES (one per GPU) with the LIBRARY driving RCCL (xm_shard_comm_*): one native
    call per frame enqueues the kernels and both collectives on the engine's stream -- ~15 us of host time where
    ShardedFrameProcessor pays ~64 us for five calls and two torch.distributed collectives.  torch.distributed (or anything else)
    is only needed once, to hand rank 0's 128-byte id to the other ranks.

        comm = ShardComm.over_torch_dist(engine, dist, n_frame_events, device)      # collective
        res, n_own = comm.resident(shard)                                           # the shard with headroom in front
        depth, bgr = comm.frame(res, n_own)                                         # asynchronous; outputs are device tensors
        ... comm.failed()                                                           # collective: redo with frame_keys() when True

    Outputs rotate through `n_out` buffers: a frame's tensors stay valid until n_out more frames have been enqueued."""

    ID_BYTES = 128

    def __init__(self, engine, comm_id: bytes, rank: int, world: int, n_frame_events: int, device, n_out: int = 2):
        import ctypes as C

        import torch

        from . import _native as N
        assert len(comm_id) == self.ID_BYTES
        self._C, self._N, self.torch = C, N, torch
        self._lib = N.load_library()
        self.eng, self.device, self.rank, self.world, self.n_frame = engine, device, int(rank), int(world), int(n_frame_events)
        self._c = C.c_void_p(None)
        idb = (C.c_char * self.ID_BYTES).from_buffer_copy(comm_id)
        N.check(self._lib.xm_shard_comm_create(engine._h, idb, self.rank, self.world, self.n_frame, C.byref(self._c)))
        tc, cap, sb, fb = C.c_int(0), C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        N.check(self._lib.xm_shard_comm_info(self._c, C.byref(tc), C.byref(cap), C.byref(sb), C.byref(fb)))
        self.takes_columns, self.cap_events, self.send_bytes, self.frame_bytes = bool(tc.value), int(cap.value), int(sb.value), int(fb.value)
        self.collective_bytes_per_frame = ({"last_events_all_gather": self.send_bytes * self.world, "u16_frame_sum_all_reduce": self.frame_bytes}
                                           if self.takes_columns else {"extrema_min_all_reduce": 16, "key_frame_max_all_reduce": self.frame_bytes})
        self._outs = [(torch.empty((engine.out_h, engine.out_w), dtype=torch.float32, device=device),
                       torch.empty((engine.out_h, engine.out_w, 3), dtype=torch.uint8, device=device)) for _ in range(max(1, n_out))]
        torch.cuda.current_stream(device).synchronize()
        self._i = 0

    @classmethod
    def new_id(cls) -> bytes:
        import ctypes as C

        from . import _native as N
        buf = (C.c_char * cls.ID_BYTES)()
        N.check(N.load_library().xm_shard_comm_id(buf))
        return bytes(buf.raw)

    @classmethod
    def over_torch_dist(cls, engine, dist, n_frame_events, device, group=None, n_out=2):
        """rank 0 draws the id, torch.distributed carries it to the others (the only use of it); collective"""
        import torch
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        on = "cpu" if dist.get_backend(group) == "gloo" else device
        idt = torch.zeros(cls.ID_BYTES, dtype=torch.uint8, device=on)
        if rank == 0:
            idt = torch.frombuffer(bytearray(cls.new_id()), dtype=torch.uint8).to(on)
        src = 0 if group is None else dist.get_global_rank(group, 0)
        dist.broadcast(idt, src=src, group=group)
        return cls(engine, bytes(idt.cpu().numpy().tobytes()), rank, world, n_frame_events, device, n_out=n_out)

    def resident(self, shard):
        """(x, y, t) of the shard once more with cap_events + 8 events of headroom in front and 8 behind, and its length"""
        return resident_with_headroom(self.torch, self.device, shard, self.cap_events), len(shard[2])

    def _next_out(self, want_depth, want_bgr):
        d, b = self._outs[self._i % len(self._outs)]
        self._i += 1
        return (d if want_depth else None), (b if want_bgr else None)

    def frame(self, resident, n_own, want_depth=True, want_bgr=True):
        """the columns merge; asynchronous on the engine's stream"""
        d, b = self._next_out(want_depth, want_bgr)
        own = lambda a: first_own_event(a, self.cap_events)
        self._N.check(self._lib.xm_shard_comm_frame(self._c, own(resident[0]), own(resident[1]), own(resident[2]), int(n_own),
                                                    None if d is None else d.data_ptr(), None if b is None else b.data_ptr()))
        return d, b

    def frame_keys(self, shard, first_index, want_depth=True, want_bgr=True):
        """the packed-key merge of the same frame (any rig / order / polarity column); asynchronous"""
        x, y, t, p = shard
        d, b = self._next_out(want_depth, want_bgr)
        n = len(t)
        ptr = lambda a: None if (a is None or n == 0) else a.data_ptr()
        self._N.check(self._lib.xm_shard_comm_frame_keys(self._c, ptr(x), ptr(y), ptr(t), ptr(p), n, GpuShardProvider._t_dtype(t), int(first_index),
                                                         None if d is None else d.data_ptr(), None if b is None else b.data_ptr()))
        return d, b

    def failed(self) -> bool:
        v = self._C.c_int(0)
        self._N.check(self._lib.xm_shard_comm_failed(self._c, self._C.byref(v)))
        return bool(v.value)

    def close(self):
        if getattr(self, "_c", None) is not None and self._c.value:
            self._lib.xm_shard_comm_destroy(self._c)
            self._c = self._C.c_void_p(None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
