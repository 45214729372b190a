"""CPU, world_size 2, gloo: the sharding protocol (index partition, MIN all-reduce of the {tmin, -tmax} buffer, packed-key
MAX all-reduce, finish) reproduces the single-process frame exactly.  Compute provider = the oracle (tests only)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleShardProvider:
    """CPU stand-in for GpuShardProvider, built on oracle/xmaps_oracle.py (checker only)."""

    def __init__(self, tables, camera):
        import xmaps_oracle as O
        self.O, self.tb, self.camera = O, tables, camera
        self.shape = (tables["cam_h"], tables["cam_w"]) if camera else (tables["rect_h"], tables["rect_w"])

    def new_key_frame(self, pad_to=1):
        cells = self.shape[0] * self.shape[1]
        return torch.zeros((cells + pad_to - 1) // pad_to * pad_to, dtype=torch.int64)

    def _frame(self, kf):
        return kf[:self.shape[0] * self.shape[1]].view(self.shape)

    def new_u16(self, n):
        return torch.zeros(n, dtype=torch.int16)

    def decode_u16(self, key_chunk, tag, out_chunk):
        k = key_chunk.numpy().astype(np.uint64)
        d = np.where((k >> np.uint64(44)) == np.uint64(tag), k & np.uint64(0xffff), np.uint64(0)).astype(np.uint16)
        out_chunk.copy_(torch.from_numpy(d.view(np.int16)))

    def finish_u16(self, disp_frame, want_bgr=True):
        O = self.O
        d = self._frame(disp_frame).numpy().view(np.uint16).astype(np.float32)
        if not self.camera:
            d = O.remap_rectified_disp_map_to_proj(d, self.tb["disp_proj_mapxy_i16"])
        depth = O.disparity_to_depth_rectified(d, self.tb["p03"])
        bgr = O.generate_color_map(O.clip_normalize_uint8_depth_frame(depth, self.tb["z_near"], self.tb["z_far"]))
        return depth, bgr

    # ---- merge = "bands": this stand-in's frame is row-major, so its bands are bands of frame ROWS; a pixel belongs to the band
    #      that holds its map target's row (pixels mapped outside the frame go with row 0), and needs 3 rows of halo (7 x 7 dilate)
    def frame_lines(self):
        return self.shape

    def halo_lines(self):
        return -1 if self.camera else 4

    def finish_u16_band(self, disp_frame, lo, hi, want_bgr=True):
        O = self.O
        depth, bgr = self.finish_u16(disp_frame, want_bgr)
        m = self.tb["disp_proj_mapxy_i16"]
        my, mx = m[..., 1].astype(np.int64), m[..., 0].astype(np.int64)
        inside = (mx >= 0) & (mx < self.tb["rect_w"]) & (my >= 0) & (my < self.tb["rect_h"])
        line = np.where(inside, my, 0)
        own = (line >= lo) & (line < hi)
        depth = torch.from_numpy(np.where(own, depth, np.float32(0.0)).astype(np.float32))
        bgr = torch.from_numpy(np.where(own[..., None], bgr, np.uint8(0)).astype(np.uint8))
        return depth, bgr

    def clear_key_frame(self, kf):
        kf.zero_()

    def new_minmax_buffer(self, shard):
        return torch.zeros(2, dtype=torch.int64 if np.issubdtype(shard[2].dtype, np.integer) else torch.float64)

    def minmax_into(self, shard, mm):
        t = shard[2]
        if len(t) == 0:  # neutral element of MIN for both entries, like xm_shard_minmax_device
            big = np.iinfo(np.int64).max if mm.dtype == torch.int64 else np.inf
            mm[0], mm[1] = big, big
        else:
            mm[0], mm[1] = t.min().item(), -t.max().item()

    def scatter(self, shard, idx_offset, mm, tag, key_frame):
        x, y, t, _ = shard
        if len(t) == 0:
            return
        lo, hi = t.dtype.type(mm[0].item()), t.dtype.type(-mm[1].item())
        kf = self.O.key_frame(self.tb, x.astype(np.int64), y.astype(np.int64), t, lo, hi, idx_offset=idx_offset,
                              tag=tag, camera_perspective=self.camera)
        view = self._frame(key_frame)
        torch.maximum(view, torch.from_numpy(kf.astype(np.int64)), out=view)

    def finish(self, key_frame, tag, want_bgr=True):
        O = self.O
        d = O.decode_key_frame(self._frame(key_frame).numpy().astype(np.uint64), tag)
        if not self.camera:
            d = O.remap_rectified_disp_map_to_proj(d, self.tb["disp_proj_mapxy_i16"])
        depth = O.disparity_to_depth_rectified(d, self.tb["p03"])
        bgr = O.generate_color_map(O.clip_normalize_uint8_depth_frame(depth, self.tb["z_near"], self.tb["z_far"]))
        return depth, bgr

    # ---- merge = "columns" (CPU stand-in for xm_shard_cols_*: the same protocol, the frame row-major like this provider's others) ----
    def _columns(self, t, mm):
        tmin, tmax = int(mm[0]), -int(mm[1])
        if tmax == tmin:
            return np.zeros(len(t), np.int64)
        return np.rint(((t - tmin) / (tmax - tmin)) * self.tb["t_px_scale"]).astype(np.int16).astype(np.int64)

    def cols_setup(self, n_frame_events):
        if self.camera:
            return None
        cap = 64 + 4 * n_frame_events // max(int(self.tb["t_px_scale"]), 1)
        cells = self.shape[0] * self.shape[1]
        self._fail = False
        return {"cap_events": cap, "send_bytes": 32 + 12 * cap, "reduce_u32": (cells + 1) // 2, "frame_bytes": 4 * ((cells + 1) // 2),
                "frame": torch.zeros(4 * ((cells + 1) // 2), dtype=torch.uint8), "send": torch.zeros(32 + 12 * cap, dtype=torch.uint8)}

    def cols_resident(self, shard, cap):
        x, y, t, p = shard
        assert p is None and t.dtype == np.int64
        return tuple(np.concatenate((np.zeros(cap, a.dtype), a)) for a in (x, y, t))

    def cols_pack(self, res, n, cap, send):
        x, y, t = (a[cap:cap + n] for a in res)
        buf = send.numpy()
        cnt = min(n, cap)
        buf[:32].view(np.int64)[:] = (int(t[0]) if n else 0, int(t[-1]) if n else 0, n, cnt)
        buf[32:32 + 2 * cap].view(np.uint16)[:cnt] = x[n - cnt:]
        buf[32 + 2 * cap:32 + 4 * cap].view(np.uint16)[:cnt] = y[n - cnt:]
        buf[32 + 4 * cap:32 + 12 * cap].view(np.int64)[:cnt] = t[n - cnt:]

    def cols_scatter(self, res, n, cap, n_frame, gathered, send_bytes, rank, world, frame):
        g = gathered.numpy()
        hdrs = [g[r * send_bytes:r * send_bytes + 32].view(np.int64) for r in range(world)]
        cells = self.shape[0] * self.shape[1]
        out = frame.numpy()[:2 * cells].view(np.uint16)
        out[:] = 0
        if any(h[2] <= 0 for h in hdrs):
            self._fail = True
            return
        tmin, tmax = min(int(h[0]) for h in hdrs), max(int(h[1]) for h in hdrs)
        mm = (tmin, -tmax)
        x, y, t = (a[cap:cap + n] for a in res)
        own_n, bad = n, False
        if rank < world - 1:
            col = self._columns(t, mm)
            own_n = int(np.searchsorted(col, col[-1], side="left")) if np.all(np.diff(col) >= 0) else 0
            bad = own_n == 0 or n - own_n > cap
        px = py = pt = np.zeros(0, np.int64)
        if rank > 0 and not bad:
            pb = g[(rank - 1) * send_bytes:rank * send_bytes]
            pc = int(hdrs[rank - 1][3])
            px = pb[32:32 + 2 * cap].view(np.uint16)[:pc]
            py = pb[32 + 2 * cap:32 + 4 * cap].view(np.uint16)[:pc]
            pt = pb[32 + 4 * cap:32 + 12 * cap].view(np.int64)[:pc]
            col = self._columns(pt, mm)
            j0 = int(np.searchsorted(col, self._columns(np.array([hdrs[rank - 1][1]]), mm)[0], side="left")) if np.all(np.diff(col) >= 0) else 0
            bad = bad or j0 == 0
            px, py, pt = px[j0:], py[j0:], pt[j0:]
        if bad:
            self._fail = True
            return
        x, y, t = np.concatenate((px, x[:own_n])), np.concatenate((py, y[:own_n])), np.concatenate((pt, t[:own_n]))
        if len(t):
            if bool(np.any(np.diff(t) < 0)) or bool(t.min() < tmin) or bool(t.max() > tmax):
                self._fail = True  # (the device's tiles object and the frame is discarded: nothing sensible to compute)
                return
            kf = self.O.key_frame(self.tb, x.astype(np.int64), y.astype(np.int64), t, np.int64(tmin), np.int64(tmax), idx_offset=0, tag=1)
            out[:] = self.O.decode_key_frame(kf.astype(np.uint64), 1).astype(np.uint16).reshape(-1)

    def cols_failed(self):
        f, self._fail = self._fail, False
        return f

    def cols_finish(self, frame, want_bgr=True):
        cells = self.shape[0] * self.shape[1]
        return self.finish_u16(frame[:2 * cells].view(torch.int16), want_bgr)

    def as_tensor(self, a):
        return a


def _cuts(n, world, uneven):
    """shard boundaries: the library's even split, or a lopsided one (10 % | nothing | 55 % | 35 %) for world_size 4"""
    if not uneven:
        from x_maps_amd.sharded import shard_bounds
        return [shard_bounds(n, r, world) for r in range(world)]
    assert world == 4
    e = [0, n // 10, n // 10, n // 10 + (n * 55) // 100, n]
    return [(e[r], e[r + 1]) for r in range(4)]


def _worker(rank, world, port, camera, out_dir, merge, uneven=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from x_maps_amd import synthetic as S
    from x_maps_amd.sharded import ShardedFrameProcessor, shard_bounds
    tb = S.make_tables(S.C_TINY)
    proc = ShardedFrameProcessor(OracleShardProvider(tb, camera), dist, merge=merge)
    assert merge != "bands" or camera or proc._bands_ok()  # (the camera view has no band finish: it falls back to the all-gather)
    for frame, n in ((0, 4000), (1, 2501), (2, 1)):  # incl. an odd split and a frame with an EMPTY shard
        evs = S.make_events(S.C_TINY, frame=frame, n=n, shuffled=(frame == 1))
        x, y, t, _ = S.to_soa(evs)
        a, b = _cuts(n, world, uneven)[rank]
        depth, bgr = proc.process_shard((x[a:b], y[a:b], t[a:b], None), a)
        np.savez(os.path.join(out_dir, f"r{rank}_f{frame}.npz"), depth=np.asarray(depth), bgr=np.asarray(bgr))
    dist.destroy_process_group()


@pytest.mark.parametrize("merge", ["all_reduce", "reduce_scatter", "bands"])
@pytest.mark.parametrize("camera", [False, True])
def test_two_rank_shards_equal_single_process(tmp_path, camera, merge):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, camera, str(tmp_path), merge), nprocs=2, join=True)
    import xmaps_oracle as O
    from x_maps_amd import synthetic as S
    tb = S.make_tables(S.C_TINY)
    for frame, n in ((0, 4000), (1, 2501), (2, 1)):
        evs = S.make_events(S.C_TINY, frame=frame, n=n, shuffled=(frame == 1))
        x, y, t, _ = S.to_soa(evs)
        ref = O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t, camera_perspective=camera)
        for r in range(2):
            got = np.load(os.path.join(str(tmp_path), f"r{r}_f{frame}.npz"))
            assert np.array_equal(got["depth"], ref["depth"]), (frame, r)
            assert np.array_equal(got["bgr"], ref["bgr"]), (frame, r)


@pytest.mark.parametrize("merge", ["all_reduce", "reduce_scatter", "bands"])
def test_four_ranks_with_uneven_shards_equal_single_process(tmp_path, merge):
    """world_size 4, shards of 10 % / nothing / 55 % / 35 % of the frame (whoever cuts the stream need not cut it evenly; a rank
    may get no events at all): the extrema MIN-reduce and the packed-key MAX-merge reproduce the single-process frame on every rank."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(4, port, False, str(tmp_path), merge, True), nprocs=4, join=True)
    import xmaps_oracle as O
    from x_maps_amd import synthetic as S
    tb = S.make_tables(S.C_TINY)
    for frame, n in ((0, 4000), (1, 2501), (2, 1)):
        evs = S.make_events(S.C_TINY, frame=frame, n=n, shuffled=(frame == 1))
        x, y, t, _ = S.to_soa(evs)
        ref = O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t)
        for r in range(4):
            got = np.load(os.path.join(str(tmp_path), f"r{r}_f{frame}.npz"))
            assert np.array_equal(got["depth"], ref["depth"]) and np.array_equal(got["bgr"], ref["bgr"]), (frame, r)


def _cols_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from x_maps_amd import synthetic as S
    from x_maps_amd.sharded import ShardedFrameProcessor, shard_bounds
    tb = S.make_tables(S.C_TINY)
    proc = ShardedFrameProcessor(OracleShardProvider(tb, False), dist, merge="columns")
    flags = []
    for frame, n, kind in ((0, 6000, "even"), (1, 6000, "lopsided"), (2, 6000, "empty_shard"), (3, 6000, "unsorted")):
        evs = S.make_events(S.C_TINY, frame=frame, n=n, shuffled=(kind == "unsorted"))
        x, y, t, _ = S.to_soa(evs)
        if kind == "even":
            a, b = shard_bounds(n, rank, world)
        else:
            e = {2: [0, n // 7, n], 4: [0, n // 10, n // 10 + (0 if kind == "empty_shard" else 700), (n * 65) // 100, n]}[world]
            a, b = e[rank], e[rank + 1]
        res, n_own = proc.columns_resident((x[a:b], y[a:b], t[a:b], None), n)
        depth, bgr = proc.process_shard_columns(res, n_own)
        failed = proc.columns_failed()
        flags.append(failed)
        if failed:  # what a caller does then: the packed keys for this frame
            fb = ShardedFrameProcessor(OracleShardProvider(tb, False), dist, merge="all_reduce")
            depth, bgr = fb.process_shard((x[a:b], y[a:b], t[a:b], None), a)
        np.savez(os.path.join(out_dir, f"r{rank}_f{frame}.npz"), depth=np.asarray(depth), bgr=np.asarray(bgr), failed=failed)
    assert proc.collective_bytes_per_frame["u16_frame_sum_all_reduce"] < 8 * tb["rect_w"] * tb["rect_h"] / 3
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_column_shards_equal_single_process(tmp_path, world):
    """merge="columns": every time column on one rank (each rank but the last hands its last column's events on), the u16 frames
    merged by SUM.  Even and lopsided shards reproduce the single-process frame; a shard without events (world 4) and a stream
    that is not sorted raise the flag ON EVERY RANK, and the fallback (packed keys) gives the frame."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_cols_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    import xmaps_oracle as O
    from x_maps_amd import synthetic as S
    tb = S.make_tables(S.C_TINY)
    for frame, kind in ((0, "even"), (1, "lopsided"), (2, "empty_shard"), (3, "unsorted")):
        evs = S.make_events(S.C_TINY, frame=frame, n=6000, shuffled=(kind == "unsorted"))
        x, y, t, _ = S.to_soa(evs)
        ref = O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t)
        for r in range(world):
            got = np.load(os.path.join(str(tmp_path), f"r{r}_f{frame}.npz"))
            assert np.array_equal(got["depth"], ref["depth"]) and np.array_equal(got["bgr"], ref["bgr"]), (kind, r)
            want_fail = kind == "unsorted" or (kind == "empty_shard" and world == 4)
            assert bool(got["failed"]) == want_fail, (kind, r)
