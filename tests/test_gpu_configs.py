"""-m gpu: the BASELINE.json configurations that are more than one frame on one handle, each checked against the oracle.

  config 3  ESL static seq1 full replay            -> stand-in: >= 30 consecutive ESL-like frames (real calibration geometry,
                                                      microsecond stamps, 3.6 ms dark gaps, negative-polarity + gap noise) fed
                                                      as packets through DepthReprojectionProcessor -> trigger finder -> GPU
                                                      (reference caller: python/depth_reprojection.py:10-29)
  config 4  10 M events/frame, sharded by index    -> C-10M split into 2 / 4 / 8 index shards on ONE GPU (private key frames,
                                                      device-side extrema, element-wise max merge) == unsharded == C oracle;
                                                      the same through ShardedFrameProcessor over a real RCCL group (one rank
                                                      here, both all-reduces issued; every rank when > 1 GPU is visible)
  config 5  60 frames x 1 M events, hipGraph       -> seeds 20230..20289 replayed from one captured graph,
                                                      all 60 depth + BGR frames == C oracle; also xm_process_batch
"""
import os
import socket
import sys

import numpy as np
import pytest

import xmaps_oracle as O
from x_maps_amd import XMapsEngine
from x_maps_amd import synthetic as S

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _soa_dev(torch, evs, dev):
    x, y, t, _ = S.to_soa(evs)
    return tuple(torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), t))


# ---------------------------------------------------------------------------------------------------------------------
# config 5
# ---------------------------------------------------------------------------------------------------------------------
def _c1m_batch(torch, n_frames, first_seed=0):
    """n_frames C-1M frames (rng seeds 20230 + first_seed ...) laid out back to back in HBM + their C-oracle frames."""
    from c_oracle import COracle
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    co = COracle(tb, False, omp=True)
    dev = torch.device("cuda", 0)
    X = torch.empty(n_frames * cfg.n_events, dtype=torch.int16, device=dev)
    Y = torch.empty_like(X)
    T = torch.empty(n_frames * cfg.n_events, dtype=torch.int64, device=dev)
    refs = []
    for f in range(n_frames):
        evs = S.make_events(cfg, frame=first_seed + f)
        x, y, t, _ = S.to_soa(evs)
        a = f * cfg.n_events
        X[a:a + cfg.n_events] = torch.from_numpy(x.view(np.int16))
        Y[a:a + cfg.n_events] = torch.from_numpy(y.view(np.int16))
        T[a:a + cfg.n_events] = torch.from_numpy(t)
        r = co.process_ev_frame(x, y, t)
        refs.append((r["depth"].copy(), r["bgr"].copy(), int(r["n_inliers"])))
    torch.cuda.synchronize()
    return cfg, tb, (X, Y, T), refs


@pytest.mark.parametrize("n_slots,sorted_decl", [(8, False), (60, False), (8, True)])
def test_config5_graph_60_frames_of_1m_events(n_slots, sorted_decl):
    """60 x C-1M from one hipGraph: n_slots = 8 -> 15 groups of 4 frames on two graph branches; n_slots = 60 -> the whole
    batch is three kernel nodes.  Replayed twice (device-side tags advance), every frame == C oracle."""
    torch = pytest.importorskip("torch")
    F = 60
    cfg, tb, (X, Y, T), refs = _c1m_batch(torch, F)
    dev = X.device
    depth = torch.zeros((F, cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
    bgr = torch.zeros((F, cfg.proj_h, cfg.proj_w, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    offs = np.arange(F + 1, dtype=np.uint64) * cfg.n_events
    with XMapsEngine(tb, n_slots=n_slots, assume_time_sorted=sorted_decl, default_priority_streams=True) as eng:
        g = eng.graph_create(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, offs, depth.data_ptr(), bgr.data_ptr())
        for rep in range(2):
            g.launch()
            eng.sync()
            d, b = depth.cpu().numpy(), bgr.cpu().numpy()
            for f in range(F):
                assert np.array_equal(d[f], refs[f][0]), (rep, f)
                assert np.array_equal(b[f], refs[f][1]), (rep, f)
            depth.zero_()
            bgr.zero_()
            torch.cuda.synchronize()
        g.close()
        # the handle still serves eager frames afterwards
        eng.process_frame_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, cfg.n_events, depth[0].data_ptr(), None)
        eng.sync()
        assert np.array_equal(depth[0].cpu().numpy(), refs[0][0])
        assert eng.last_frame_stats().n_inliers == refs[0][2]


@pytest.mark.parametrize("camera", [False, True])
@pytest.mark.parametrize("try_sorted", [False, True])
def test_process_batch_groups_of_frames(camera, try_sorted):
    """xm_process_batch: groups of 4 frames per set of launches over 8 slots, interleaved with single-frame calls; one
    frame of every second group is shuffled (try-sorted mode must redo exactly those)."""
    torch = pytest.importorskip("torch")
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    F, n = 24, 30_000
    evs = []
    for f in range(F):
        e = S.make_events(cfg, frame=100 + f, n=n - 97 * (f % 5))
        if f % 8 == 3:
            e = e[rng.permutation(len(e))]
        evs.append(e)
    refs = []
    for e in evs:
        x, y, t, _ = S.to_soa(e)
        refs.append(O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t, camera_perspective=camera))
    offs = np.concatenate(([0], np.cumsum([len(e) for e in evs]))).astype(np.uint64)
    cat = np.concatenate(evs)
    X, Y, T = _soa_dev(torch, cat, dev)
    H, W = (cfg.cam_h, cfg.cam_w) if camera else (cfg.proj_h, cfg.proj_w)
    depth = torch.zeros((F, H, W), dtype=torch.float32, device=dev)
    bgr = torch.zeros((F, H, W, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    with XMapsEngine(tb, camera_perspective=camera, n_slots=8, try_sorted=try_sorted) as eng:
        for rep in range(2):
            f = 0
            while f < F:
                if f == 12:  # a single-frame call between two groups
                    a = int(offs[f])
                    eng.process_frame_device(X[a:].data_ptr(), Y[a:].data_ptr(), T[a:].data_ptr(), None, len(evs[f]),
                                             depth[f].data_ptr(), bgr[f].data_ptr())
                    f += 1
                    continue
                k = min(4 if f != 13 else 3, F - f)
                a = int(offs[f])
                eng.process_batch_device(X[a:].data_ptr(), Y[a:].data_ptr(), T[a:].data_ptr(), None, offs[f:f + k + 1] - offs[f],
                                         depth[f].data_ptr(), bgr[f].data_ptr())
                f += k
            eng.sync()
            d, b = depth.cpu().numpy(), bgr.cpu().numpy()
            for i in range(F):
                assert np.array_equal(d[i], refs[i]["depth"]), (rep, i)
                assert np.array_equal(b[i], refs[i]["bgr"]), (rep, i)
            depth.zero_()
            bgr.zero_()
            torch.cuda.synchronize()
        if try_sorted:
            assert eng.sorted_fallbacks() == 2 * 3
        with pytest.raises(ValueError):  # more frames than slots
            eng.process_batch_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, offs[:10], depth.data_ptr(), None)


def test_process_batch_c1m_and_empty_frame():
    """Full-size frames through the multi-frame launches, with an EMPTY frame in the middle of the group."""
    torch = pytest.importorskip("torch")
    cfg, tb, (X, Y, T), refs = _c1m_batch(torch, 3, first_seed=70)
    dev = X.device
    n = cfg.n_events
    offs = np.array([0, n, n, 2 * n, 3 * n], dtype=np.uint64)  # frame 1 is empty
    depth = torch.ones((4, cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
    bgr = torch.zeros((4, cfg.proj_h, cfg.proj_w, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for kw in ({}, {"try_sorted": True}):
        with XMapsEngine(tb, n_slots=4, **kw) as eng:
            for rep in range(2):
                eng.process_batch_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, offs, depth.data_ptr(), bgr.data_ptr())
                eng.sync()
                d, b = depth.cpu().numpy(), bgr.cpu().numpy()
                for i, r in ((0, 0), (2, 1), (3, 2)):
                    assert np.array_equal(d[i], refs[r][0]) and np.array_equal(b[i], refs[r][1]), (kw, rep, i)
                assert not d[1].any() and (b[1] == 255).all()
                depth.fill_(1.0)
                torch.cuda.synchronize()


def test_process_batch_verdicts_are_read_from_the_groups_stream():
    """Two groups of 8 full-size frames back to back on the SAME 8 slots, one frame of each group shuffled: when the second
    call settles the first group's verdicts the group's launches are still running (K1 of 8 frames, then K2) -- on the group's
    stream, not on the slots' own streams.  The shuffled frames must be found and redone (regression: the wait looked at
    the slot's own, idle stream and took 'nothing pending' for 'shortcut held')."""
    torch = pytest.importorskip("torch")
    from c_oracle import COracle
    cfg, tb, (X, Y, T), refs = _c1m_batch(torch, 8, first_seed=90)
    dev = X.device
    n = cfg.n_events
    rng = np.random.default_rng(2)
    perm = torch.from_numpy(rng.permutation(n)).to(dev)
    a = 3 * n
    X[a:a + n], Y[a:a + n], T[a:a + n] = X[a:a + n][perm].clone(), Y[a:a + n][perm].clone(), T[a:a + n][perm].clone()
    torch.cuda.synchronize()
    r3 = COracle(tb, False, omp=True).process_ev_frame(X[a:a + n].cpu().numpy().view(np.uint16), Y[a:a + n].cpu().numpy().view(np.uint16),
                                                       T[a:a + n].cpu().numpy())
    refs[3] = (r3["depth"].copy(), r3["bgr"].copy(), int(r3["n_inliers"]))
    offs = np.arange(9, dtype=np.uint64) * n
    d1 = torch.zeros((8, cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
    d2 = torch.zeros_like(d1)
    torch.cuda.synchronize()
    with XMapsEngine(tb, n_slots=8) as eng:
        eng.process_batch_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, offs, d1.data_ptr(), None)
        eng.process_batch_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, offs, d2.data_ptr(), None)
        eng.sync()
        assert eng.sorted_fallbacks() == 2
        for d in (d1, d2):
            h = d.cpu().numpy()
            for f in range(8):
                assert np.array_equal(h[f], refs[f][0]), f


# ---------------------------------------------------------------------------------------------------------------------
# config 4
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c10m():
    torch = pytest.importorskip("torch")
    from c_oracle import COracle
    cfg = S.C_10M
    tb = S.make_tables(cfg)
    evs = S.make_events(cfg)
    x, y, t, _ = S.to_soa(evs)
    ref = COracle(tb, False, omp=True).process_ev_frame(x, y, t)
    ref = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in ref.items()}
    return cfg, tb, (x, y, t), ref


@pytest.mark.parametrize("world", [2, 4, 8])
def test_config4_c10m_index_shards_on_one_gpu(c10m, world):
    """The sharded protocol with all `world` ranks played by one GPU: per-shard extrema left in device memory, MIN over the
    shards (what the all-reduce computes), private key frames, element-wise MAX merge, frame kernel -- no host round trip
    anywhere.  Result == the unsharded frame == the C oracle."""
    import torch
    from x_maps_amd.sharded import shard_bounds
    cfg, tb, (x, y, t), ref = c10m
    dev = torch.device("cuda", 0)
    X, Y, T = (torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), t))
    n = len(t)
    with XMapsEngine(tb) as eng:
        stream = torch.cuda.ExternalStream(eng.stream(0), device=dev)
        depth = torch.zeros((cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
        bgr = torch.zeros((cfg.proj_h, cfg.proj_w, 3), dtype=torch.uint8, device=dev)
        kfs = [torch.zeros(eng.key_shape, dtype=torch.int64, device=dev) for _ in range(world)]
        mms = torch.zeros((world, 2), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        for tag in (1, 2):  # two consecutive frames on the same key frames: tags separate them
            with torch.cuda.stream(stream):
                for r in range(world):
                    a, b = shard_bounds(n, r, world)
                    eng.shard_minmax_device(T[a:].data_ptr(), None, b - a, mms[r].data_ptr())
                mm = mms.min(dim=0).values.contiguous()  # = all_reduce(MIN) of the ranks' 16-byte buffers
                for r in range(world):
                    a, b = shard_bounds(n, r, world)
                    eng.shard_scatter_device(X[a:].data_ptr(), Y[a:].data_ptr(), T[a:].data_ptr(), None, b - a, a,
                                             mm.data_ptr(), tag, kfs[r].data_ptr())
                merged = kfs[0].clone()
                for r in range(1, world):
                    torch.maximum(merged, kfs[r], out=merged)  # = all_reduce(MAX) of the key frames
                eng.shard_finish(merged.data_ptr(), tag, depth.data_ptr(), bgr.data_ptr())
            eng.sync()
            torch.cuda.synchronize()
            assert mm[0].item() == t.min() and -mm[1].item() == t.max()
            assert np.array_equal(depth.cpu().numpy(), ref["depth"]), (world, tag)
            assert np.array_equal(bgr.cpu().numpy(), ref["bgr"]), (world, tag)
            depth.zero_()
            torch.cuda.synchronize()


@pytest.mark.parametrize("world", [1, 3, 8])
def test_config4_reduce_scatter_merge_on_one_gpu(c10m, world):
    """The cheaper exchange (reduce-scatter of the key frame, decode of the own chunk to u16, all-gather of the u16 chunks,
    frame kernel on the plain disparity frame) with every rank played by one GPU: chunks of the padded flat key frame are
    max-merged and decoded one by one (what each rank does with its chunk), concatenated (the all-gather), finished."""
    import torch
    from x_maps_amd.sharded import shard_bounds
    cfg, tb, (x, y, t), ref = c10m
    dev = torch.device("cuda", 0)
    X, Y, T = (torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), t))
    n = len(t)
    with XMapsEngine(tb) as eng:
        stream = torch.cuda.ExternalStream(eng.stream(0), device=dev)
        cells = eng.key_shape[0] * eng.key_shape[1]
        padded = (cells + world - 1) // world * world
        chunk = padded // world
        depth = torch.zeros((cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
        bgr = torch.zeros((cfg.proj_h, cfg.proj_w, 3), dtype=torch.uint8, device=dev)
        kfs = [torch.zeros(padded, dtype=torch.int64, device=dev) for _ in range(world)]
        mms = torch.zeros((world, 2), dtype=torch.int64, device=dev)
        u16 = torch.zeros(padded, dtype=torch.int16, device=dev)
        torch.cuda.synchronize()
        tag = 5
        with torch.cuda.stream(stream):
            for r in range(world):
                a, b = shard_bounds(n, r, world)
                eng.shard_minmax_device(T[a:].data_ptr(), None, b - a, mms[r].data_ptr())
            mm = mms.min(dim=0).values.contiguous()
            for r in range(world):
                a, b = shard_bounds(n, r, world)
                eng.shard_scatter_device(X[a:].data_ptr(), Y[a:].data_ptr(), T[a:].data_ptr(), None, b - a, a, mm.data_ptr(), tag,
                                         kfs[r].data_ptr())
            for r in range(world):  # rank r's part of the reduce-scatter + its decode
                red = kfs[0][r * chunk:(r + 1) * chunk].clone()
                for q in range(1, world):
                    torch.maximum(red, kfs[q][r * chunk:(r + 1) * chunk], out=red)
                eng.shard_decode_u16(red.data_ptr(), chunk, tag, u16[r * chunk:].data_ptr())
            eng.shard_finish_u16(u16.data_ptr(), depth.data_ptr(), bgr.data_ptr())
        eng.sync()
        torch.cuda.synchronize()
        assert np.array_equal(depth.cpu().numpy(), ref["depth"]) and np.array_equal(bgr.cpu().numpy(), ref["bgr"])


@pytest.mark.parametrize("world", [2, 5, 8])
def test_config4_band_sharded_finish_on_one_gpu(c10m, world):
    """merge = "bands" with every rank played by one GPU: rank r holds the merged u16 disparities of its chunk of frame columns
    plus a halo of k2_patch_cols_max + 1 columns from either neighbour (everything else of its frame buffer is POISONED here),
    finishes the projector tiles centred on its band into a zeroed output, and the element-wise maximum of the ranks' partial
    frames must be the oracle's frame: every tile has exactly one owner, and an owner sees every cell its tiles read."""
    import torch
    cfg, tb, (x, y, t), ref = c10m
    dev = torch.device("cuda", 0)
    X, Y, T = (torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), t))
    n = len(t)
    with XMapsEngine(tb) as eng:
        stream = torch.cuda.ExternalStream(eng.stream(0), device=dev)
        rect_w, rect_h = eng.key_shape
        cells = rect_w * rect_h
        padded = (cells + world - 1) // world * world
        C = padded // world
        halo = eng.k2_patch_cols_max() + 1
        assert 0 < halo and halo * rect_h <= C
        Hc = halo * rect_h
        kf = torch.zeros(padded, dtype=torch.int64, device=dev)
        mm = torch.zeros(2, dtype=torch.int64, device=dev)
        u16 = torch.zeros(padded, dtype=torch.int16, device=dev)
        acc_d = torch.zeros((cfg.proj_h, cfg.proj_w), dtype=torch.int32, device=dev)
        acc_b = torch.zeros((cfg.proj_h, cfg.proj_w, 3), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        tag = 3
        with torch.cuda.stream(stream):
            eng.shard_minmax_device(T.data_ptr(), None, n, mm.data_ptr())
            eng.shard_scatter_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, n, 0, mm.data_ptr(), tag, kf.data_ptr())
            eng.shard_decode_u16(kf.data_ptr(), padded, tag, u16.data_ptr())  # the merged frame (what the reduce-scatter leaves, chunk by chunk)
        eng.sync()
        torch.cuda.synchronize()
        for r in range(world):
            full = torch.full((padded,), 0x7777, dtype=torch.int16, device=dev)  # poison: a cell the rank was not given must not be read
            a, b = max(r * C - Hc, 0), min((r + 1) * C + Hc, padded)
            full[a:b] = u16[a:b]
            lo = -(-(r * C) // rect_h)
            hi = min(-(-((r + 1) * C) // rect_h), rect_w) if r < world - 1 else rect_w
            d = torch.zeros((cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
            bg = torch.zeros((cfg.proj_h, cfg.proj_w, 3), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            with torch.cuda.stream(stream):
                eng.shard_finish_u16_band(full.data_ptr(), lo, hi, d.data_ptr(), bg.data_ptr())
            eng.sync()
            torch.cuda.synchronize()
            # every pixel is written by exactly one rank: no overlap with what earlier ranks produced
            assert not bool(((acc_b.amax(dim=-1) > 0) & (bg.amax(dim=-1) > 0)).any()), r
            torch.maximum(acc_d, d.view(torch.int32), out=acc_d)
            torch.maximum(acc_b, bg, out=acc_b)
        assert np.array_equal(acc_d.view(torch.float32).cpu().numpy(), ref["depth"]) and np.array_equal(acc_b.cpu().numpy(), ref["bgr"])


@pytest.mark.parametrize("camera", [False, True])
def test_reduce_scatter_merge_through_the_processor_over_rccl(camera, tmp_path):
    """ShardedFrameProcessor(merge="reduce_scatter") over a real RCCL group (one rank here, the collectives are issued all the
    same), both views, incl. border tiles of the u16 frame kernel."""
    import torch
    import torch.distributed as dist
    from x_maps_amd.sharded import GpuShardProvider, ShardedFrameProcessor
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method=f"file://{tmp_path}/rdzv", rank=0, world_size=1, device_id=dev)
        created = True
    try:
        with XMapsEngine(tb, camera_perspective=camera) as eng:
            procs = [ShardedFrameProcessor(GpuShardProvider(eng, dev), dist, always_reduce=True, merge="reduce_scatter")]
            if not camera:  # ... and the band-sharded finish (one rank: one band, no halo exchange; the outputs are MAX-reduced all the same)
                procs.append(ShardedFrameProcessor(GpuShardProvider(eng, dev), dist, always_reduce=True, merge="bands"))
            for proc in procs:
              for f in range(3):
                evs = S.make_events(cfg, frame=60 + f, n=2500 + 400 * f, shuffled=(f == 1))
                sh = _soa_dev(torch, evs, dev) + (None,)
                torch.cuda.synchronize()
                depth, bgr = proc.process_shard(sh, 0)
                eng.sync()
                torch.cuda.synchronize()
                x, y, t, _ = S.to_soa(evs)
                r = O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t, camera_perspective=camera)
                assert np.array_equal(depth.cpu().numpy(), r["depth"]) and np.array_equal(bgr.cpu().numpy(), r["bgr"]), (proc.merge, f)
              assert proc.collectives_issued == 3 * (4 if proc.merge == "bands" else 3), proc.merge  # bands: extrema, reduce-scatter, depth, BGR
    finally:
        if created:
            dist.destroy_process_group()


def test_config4_sharded_processor_over_rccl_issues_both_collectives(c10m, tmp_path):
    """ShardedFrameProcessor + GpuShardProvider over a real RCCL group at C-10M.  One rank on this box, `always_reduce`
    makes it issue the 16-byte MIN and the 55.8 MB MAX all-reduce anyway (RCCL kernels run, data unchanged)."""
    import torch
    import torch.distributed as dist
    from x_maps_amd.sharded import GpuShardProvider, ShardedFrameProcessor
    cfg, tb, (x, y, t), ref = c10m
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method=f"file://{tmp_path}/rdzv", rank=0, world_size=1, device_id=dev)
        created = True
    try:
        sh = tuple(torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), t)) + (None,)
        torch.cuda.synchronize()
        with XMapsEngine(tb) as eng:
            proc = ShardedFrameProcessor(GpuShardProvider(eng, dev), dist, always_reduce=True)
            for rep in range(2):
                depth, bgr = proc.process_shard(sh, 0)
                eng.sync()
                torch.cuda.synchronize()
                assert np.array_equal(depth.cpu().numpy(), ref["depth"]) and np.array_equal(bgr.cpu().numpy(), ref["bgr"])
            assert proc.collectives_issued == 4
    finally:
        if created:
            dist.destroy_process_group()


def _rccl_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from x_maps_amd.sharded import GpuShardProvider, ShardedFrameProcessor, shard_bounds
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    cfg = S.C_10M
    tb = S.make_tables(cfg)
    x, y, t, _ = S.to_soa(S.make_events(cfg))
    a, b = shard_bounds(len(t), rank, world)
    sh = tuple(torch.from_numpy(v[a:b].copy()).to(dev) for v in (x.view(np.int16), y.view(np.int16), t)) + (None,)
    torch.cuda.synchronize()
    with XMapsEngine(tb, device=rank) as eng:
        proc = ShardedFrameProcessor(GpuShardProvider(eng, dev), dist)
        depth, bgr = proc.process_shard(sh, a)
        eng.sync()
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), depth=depth.cpu().numpy(), bgr=bgr.cpu().numpy())
    dist.destroy_process_group()


def test_config4_sharded_over_rccl_all_visible_gpus(c10m, tmp_path):
    """Every visible GPU is a rank (skipped on a single-GPU box): the C-10M frame sharded by index, merged over RCCL/xGMI."""
    import torch
    import torch.multiprocessing as mp
    world = torch.cuda.device_count()
    if world < 2:
        pytest.skip("one GPU visible: the multi-rank RCCL run needs >= 2 (the one-GPU shard test covers the arithmetic)")
    cfg, tb, _, ref = c10m
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_rccl_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        assert np.array_equal(got["depth"], ref["depth"]) and np.array_equal(got["bgr"], ref["bgr"]), r


# ---------------------------------------------------------------------------------------------------------------------
# config 3
# ---------------------------------------------------------------------------------------------------------------------
def test_config3_esl_like_replay_through_the_processor():
    """36 consecutive ESL-like frames (real calibration geometry, ~150 k events / frame, microsecond stamps, 3.6 ms gaps,
    10 % negative-polarity events, noise events inside some gaps) as 1/4-period packets through
    DepthReprojectionProcessor -> polarity filter -> trigger finder -> fused GPU frame -> window: >= 30 frames come out
    (the reference's trigger finder cannot cut the first and the last frame of a stream, and drops a buffer that holds a
    full period but only one pause, trigger_finder.py:146-189 -- a handful of frames are lost to that by design) and
    EVERY one equals the oracle run on the events the trigger finder cut."""
    from x_maps_amd import rig
    from x_maps_amd.depth_reprojection_processor import DepthReprojectionProcessor, RuntimeParams
    cp, tb, _, _ = rig.make_esl_like(row_stride=13)
    stream, rendered = rig.render_stream(cp, tb, n_frames=36, row_stride=13, seed=3)
    assert (np.diff(stream["t"]) >= 0).all() and 100_000 < np.mean([len(f) for f in rendered]) < 200_000
    params = RuntimeParams(camera_width=640, camera_height=480, projector_width=1080, projector_height=1920, projector_fps=60,
                           z_near=0.1, z_far=1.2, calib=None, projector_time_map=None, no_frame_dropping=True,
                           camera_perspective=False, tables=tb, device_ingest=False)  # (the host chain: the test spies on its trigger finder)
    cut, shown = [], []

    class Window:
        def should_close(self):
            return False

        def show_async(self, img):
            shown.append(img)

    with DepthReprojectionProcessor(params, window=Window()) as proc:
        orig = proc._pipe.process_ev_frame

        def spy(evs):
            cut.append(evs.copy())
            orig(evs)

        proc._pipe.trigger_finder.frame_callback = spy
        packet = int(1e6 / 60 / 4)
        edges = np.arange(stream["t"][0], stream["t"][-1] + packet, packet)
        cuts = np.searchsorted(stream["t"], edges)
        for a, b in zip(cuts[:-1], cuts[1:]):
            proc.process_events(stream[a:b])
        assert proc.stats_printer.counters["processed evs"] == len(stream)
    assert len(cut) == len(shown) >= 30
    t_prev = -1
    for evs, img in zip(cut, shown):
        assert (evs["p"] == 1).all() and len(evs) > 100_000 and evs["t"][0] > t_prev
        t_prev = evs["t"][-1]
        x, y, t, _ = S.to_soa(evs)
        ref = O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t)
        assert img.shape == (1920, 1080, 3) and np.array_equal(img, ref["bgr"])
    # no two shown frames are the same array / the same picture (fresh array per frame, scene offset changes)
    assert len({id(i) for i in shown}) == len(shown)
