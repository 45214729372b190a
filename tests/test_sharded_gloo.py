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
