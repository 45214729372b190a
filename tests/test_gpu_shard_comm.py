"""-m gpu: the library's own multi-process shard communicator (xm_shard_comm_*: one call per frame, kernels + RCCL collectives
enqueued by the library) on the one GPU of the box = a world of one rank: RCCL really runs (communicator from a unique id,
all-gather, SUM / MIN / MAX all-reduces), nothing crosses a link.  Frames == the C oracle for both merges, several frames in a
row on rotating outputs, two communicators (frames in flight on two engines) side by side, the flag of a frame that cannot be
handled, the id carried by torch.distributed.  (Worlds > 1 need one GPU per rank: the exchange itself is pinned rank by rank by
tests/test_gpu_shard_cols.py -- the same kernels with the collectives done by hand -- and by the gloo tests of the protocol.)"""
import numpy as np
import pytest

from c_oracle import COracle
from x_maps_amd import XMapsEngine
from x_maps_amd import synthetic as S
from x_maps_amd.sharded import ShardComm

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def c1m():
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    frames, refs = [], []
    for f in range(3):
        x, y, t, _ = S.to_soa(S.make_events(cfg, frame=f))
        ref = COracle(tb, False, omp=True).process_ev_frame(x, y, t)
        frames.append((x, y, t))
        refs.append({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in ref.items()})
    return cfg, tb, frames, refs


def _dev_shard(frame, dev):
    x, y, t = frame
    return tuple(torch.from_numpy(a.copy()).to(dev) for a in (x.view(np.int16), y.view(np.int16), t)) + (None,)


def test_frames_through_the_library_communicator_equal_the_oracle(c1m):
    cfg, tb, frames, refs = c1m
    dev = torch.device("cuda", 0)
    with XMapsEngine(tb) as eng, ShardComm(eng, ShardComm.new_id(), 0, 1, len(frames[0][2]), dev, n_out=3) as comm:
        assert comm.takes_columns and comm.cap_events % 8 == 0 and comm.world == 1
        shards = [_dev_shard(f, dev) for f in frames]
        res = [comm.resident(sh) for sh in shards]
        outs = [comm.frame(*r) for r in res]  # three frames enqueued back to back, three output buffers
        assert not comm.failed()              # (synchronises)
        for (d, b), ref in zip(outs, refs):
            assert np.array_equal(d.cpu().numpy(), ref["depth"]) and np.array_equal(b.cpu().numpy(), ref["bgr"])
        # the packed keys through the same communicator: same frames
        for sh, ref in zip(shards, refs):
            d, b = comm.frame_keys(sh, 0)
            eng.sync()
            assert np.array_equal(d.cpu().numpy(), ref["depth"]) and np.array_equal(b.cpu().numpy(), ref["bgr"])
        # depth only / nothing (a rank that does not need the result)
        d, b = comm.frame(*res[1], want_bgr=False)
        eng.sync()
        assert b is None and np.array_equal(d.cpu().numpy(), refs[1]["depth"])
        assert comm.frame(*res[2], want_depth=False, want_bgr=False) == (None, None)
        assert not comm.failed()


def test_two_communicators_keep_two_frames_in_flight(c1m):
    cfg, tb, frames, refs = c1m
    dev = torch.device("cuda", 0)
    n = len(frames[0][2])
    with XMapsEngine(tb) as e0, XMapsEngine(tb) as e1:
        lanes = [ShardComm(e, ShardComm.new_id(), 0, 1, n, dev) for e in (e0, e1)]
        res = [lanes[i % 2].resident(_dev_shard(frames[i], dev)) for i in range(3)]
        got = []
        for rep in range(4):
            for i in range(3):
                d, b = lanes[i % 2].frame(*res[i])
                if rep == 3:
                    lanes[i % 2].eng.sync()
                    got.append((d.cpu().numpy().copy(), b.cpu().numpy().copy()))
        assert not lanes[0].failed() and not lanes[1].failed()
        for (d, b), ref in zip(got, refs):
            assert np.array_equal(d, ref["depth"]) and np.array_equal(b, ref["bgr"])
        for c in lanes:
            c.close()


def test_a_frame_that_cannot_be_handled_raises_the_flag_and_the_keys_redo_it(c1m):
    cfg, tb, frames, refs = c1m
    dev = torch.device("cuda", 0)
    x, y, t = frames[0]
    perm = np.random.default_rng(3).permutation(len(t))
    shuffled = (x[perm], y[perm], t[perm])  # not time-sorted: the column tiles object
    ref = COracle(tb, False, omp=True).process_ev_frame(*shuffled)
    with XMapsEngine(tb) as eng, ShardComm(eng, ShardComm.new_id(), 0, 1, len(t), dev) as comm:
        sh = _dev_shard(shuffled, dev)
        comm.frame(*comm.resident(sh))
        assert comm.failed() and not comm.failed()  # reported once, then cleared
        d, b = comm.frame_keys(sh, 0)
        eng.sync()
        assert np.array_equal(d.cpu().numpy(), ref["depth"]) and np.array_equal(b.cpu().numpy(), ref["bgr"])


def test_a_rig_that_does_not_take_the_column_tiles_says_so_and_runs_on_the_keys():
    tb = S.make_tables_shared_cells()
    x, y, t, _ = S.to_soa(S.make_events(S.C_SHARED))
    ref = COracle(tb, False, omp=True).process_ev_frame(x, y, t)
    dev = torch.device("cuda", 0)
    with XMapsEngine(tb) as eng, ShardComm(eng, ShardComm.new_id(), 0, 1, len(t), dev) as comm:
        assert not comm.takes_columns
        sh = _dev_shard((x, y, t), dev)
        with pytest.raises(Exception, match="column tiles"):
            comm.frame((sh[0], sh[1], sh[2]), len(t))
        d, b = comm.frame_keys(sh, 0)
        eng.sync()
        assert np.array_equal(d.cpu().numpy(), ref["depth"]) and np.array_equal(b.cpu().numpy(), ref["bgr"])


def test_the_id_travels_over_torch_distributed(c1m, tmp_path):
    import torch.distributed as dist
    cfg, tb, frames, refs = c1m
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/rdzv", rank=0, world_size=1, device_id=dev)
    try:
        with XMapsEngine(tb) as eng, ShardComm.over_torch_dist(eng, dist, len(frames[0][2]), dev) as comm:
            d, b = comm.frame(*comm.resident(_dev_shard(frames[0], dev)))
            assert not comm.failed()
            assert np.array_equal(d.cpu().numpy(), refs[0]["depth"]) and np.array_equal(b.cpu().numpy(), refs[0]["bgr"])
    finally:
        dist.destroy_process_group()


def test_bad_arguments_are_refused():
    tb = S.make_tables(S.C_TINY)
    dev = torch.device("cuda", 0)
    with XMapsEngine(tb) as eng:
        with pytest.raises(Exception, match="rank"):
            ShardComm(eng, ShardComm.new_id(), 2, 2, 1000, dev)
