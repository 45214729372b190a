"""-m gpu: xm_create_sharded / xm_sharded_process_frame (last row: the C-ABI entry that owns the RCCL
communicators) -- one frame sharded by event index over the devices of ONE process, against the oracle: with one device (the
all-reduces run through RCCL on a one-rank communicator) and with every visible device when there is more than one."""
import numpy as np
import pytest

import xmaps_oracle as O
from conftest import xm_option
from x_maps_amd import synthetic as S
from x_maps_amd.sharded import ShardedDevices

pytestmark = pytest.mark.gpu


def _devices():
    import torch
    return list(range(torch.cuda.device_count()))


def _check(tb, sh, evs, camera=False, p=None):
    x, y, t, pp = S.to_soa(evs)
    depth, bgr, st = sh.process_frame(x, y, t, p=pp if p else None)
    if p:
        x, y, t = x[pp == 1], y[pp == 1], t[pp == 1]
    ref = O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t, camera_perspective=camera)
    assert np.array_equal(depth, ref["depth"]) and np.array_equal(bgr, ref["bgr"])
    if len(t):
        assert (st["t_min"], st["t_max"]) == (float(t.min()), float(t.max()))
    return st


@pytest.mark.parametrize("camera", [False, True])
def test_one_device(camera):
    tb = S.make_tables(S.C_TINY)
    with ShardedDevices(tb, devices=[0], camera_perspective=camera) as sh:
        assert sh.n_dev == 1 and sh.uses_rccl
        for f in range(3):
            st = _check(tb, sh, S.make_events(S.C_TINY, frame=f, n=30_000), camera)
        assert st["key_frame_all_reduce_ms"] > 0
        _check(tb, sh, S.make_events(S.C_TINY, frame=5, n=20_000)[::-1].copy(), camera)          # not sorted
        _check(tb, sh, S.make_events(S.C_TINY, frame=6, n=20_000, p_zero_fraction=0.3), camera, p=True)  # polarity column
        depth, bgr, _ = sh.process_frame(np.zeros(0, np.uint16), np.zeros(0, np.uint16), np.zeros(0, np.int64))  # defined: empty frame
        assert not depth.any() and (bgr == 255).all()


def test_float_time_stamps_one_device():
    tb = S.make_tables(S.C_TINY)
    evs = S.make_events(S.C_TINY, frame=2, n=25_000)
    x, y, t, _ = S.to_soa(evs)
    with ShardedDevices(tb, devices=[0]) as sh:
        for dt in (np.float64, np.float32):
            tf = t.astype(dt)
            depth, _, _ = sh.process_frame(x, y, tf, want_bgr=False)
            ref = O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), tf, want_bgr=False)
            assert np.array_equal(depth, ref["depth"]), dt


def test_c1m_frame_one_device_and_all_devices():
    """a C-1M frame: one device, then every visible device (on a one-GPU box the two are the same handle shape; on the driver's
    multi-GPU node the second run crosses xGMI) -- the same frame as the oracle's either way"""
    tb = S.make_tables(S.C_1M)
    evs = S.make_events(S.C_1M, frame=3)
    devs = _devices()
    for ids in ([0], devs):
        with ShardedDevices(tb, devices=ids) as sh:
            assert sh.n_dev == len(ids)
            _check(tb, sh, evs)


def test_which_exchange_a_frame_takes():
    """time-sorted int64 frames on an injective rig: the columns exchange (all-gather + SUM of u16 frames); a frame that objects is
    redone with the packed keys; polarity columns, float stamps, the camera view and XM_SHARDED_KEYS=1 take the keys -- the same
    frames as the oracle's every time"""
    tb = S.make_tables(S.C_1M)
    evs = S.make_events(S.C_1M, frame=4)
    with ShardedDevices(tb, devices=[0]) as sh:
        _check(tb, sh, evs)
        _check(tb, sh, S.make_events(S.C_1M, frame=5))  # (a second frame on the same buffers: nothing stale)
        assert sh.stats() == {"frames_columns": 2, "frames_keys": 0, "frames_redone": 0}
        shuffled = evs[np.random.default_rng(1).permutation(len(evs))]
        _check(tb, sh, shuffled)
        assert sh.stats() == {"frames_columns": 3, "frames_keys": 0, "frames_redone": 1}
        _check(tb, sh, evs)  # (and the columns again after the redo)
        _check(tb, sh, S.make_events(S.C_1M, frame=6, p_zero_fraction=0.2), p=True)
        x, y, t, _ = S.to_soa(evs)
        depth, _, _ = sh.process_frame(x, y, t.astype(np.float64), want_bgr=False)
        assert np.array_equal(depth, O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t.astype(np.float64), want_bgr=False)["depth"])
        assert sh.stats() == {"frames_columns": 4, "frames_keys": 2, "frames_redone": 1}
        xm_option("XM_SHARDED_KEYS", "1")
        _check(tb, sh, evs)
        assert sh.stats()["frames_keys"] == 3
    xm_option("XM_SHARDED_KEYS", "0")
    with ShardedDevices(tb, devices=[0], camera_perspective=True) as sh:
        _check(tb, sh, evs, camera=True)
        assert sh.stats() == {"frames_columns": 0, "frames_keys": 1, "frames_redone": 0}


def test_bad_device_lists_are_rejected():
    from x_maps_amd._native import XMapsNativeError
    tb = S.make_tables(S.C_TINY)
    for ids in ([0, 0], [99], []):
        with pytest.raises((XMapsNativeError, ValueError)):
            ShardedDevices(tb, devices=ids)


# ---- N > 1 through the C entry on a one-GPU box: virtual ranks (XM_SHARD_FAKE_RANKS) ---------------------------------------------
@pytest.mark.parametrize("W", [2, 4, 8])
def test_virtual_ranks_run_the_whole_exchange_through_the_c_entry(W):
    """xm_create_sharded with W virtual ranks on device 0: W handles and threads, shard bounds, the columns exchange (every rank's
    header + last events gathered, the predecessor's last column, SUM of the u16 frames), the redo with packed keys (MIN of the
    extrema, MAX of the keys), the agreement in front of every collective -- the collectives themselves emulated on the one device
    (RCCL refuses two ranks on one GPU).  Every frame == the oracle's."""
    xm_option("XM_SHARD_FAKE_RANKS", str(W))
    tb = S.make_tables(S.C_1M)
    evs = S.make_events(S.C_1M, frame=11)
    with ShardedDevices(tb, devices=[0]) as sh:
        assert sh.n_dev == W and not sh.uses_rccl
        _check(tb, sh, evs)
        _check(tb, sh, S.make_events(S.C_1M, frame=12, n=1_000_003))       # shards of unequal length
        assert sh.stats() == {"frames_columns": 2, "frames_keys": 0, "frames_redone": 0}
        shuffled = evs[np.random.default_rng(W).permutation(len(evs))]
        _check(tb, sh, shuffled)                                            # every piece objects: redone with the keys
        assert sh.stats()["frames_redone"] == 1
        _check(tb, sh, S.make_events(S.C_1M, frame=13, p_zero_fraction=0.2), p=True)   # polarity column: keys
        x, y, t, _ = S.to_soa(evs)
        depth, _, _ = sh.process_frame(x, y, t.astype(np.float64), want_bgr=False)      # float stamps: keys, MIN over doubles
        assert np.array_equal(depth, O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t.astype(np.float64), want_bgr=False)["depth"])
        _check(tb, sh, evs[:W * 64 + 5])                                    # hardly an event per rank
        _check(tb, sh, evs[:3])                                             # fewer events than ranks: empty shards
        _check(tb, sh, evs)                                                 # and the columns again


@pytest.mark.timeout(120)
@pytest.mark.parametrize("where", ["1:1", "2:2", "0:2", "1:3", "0:3"])
def test_a_rank_failing_in_front_of_a_collective_fails_the_frame_on_every_rank(where):
    """XM_SHARD_FAIL_AT = <rank>:<collective>: that rank's thread fails before entering the collective; the agreement barrier makes
    every other rank skip it too -- the call returns an error naming the device instead of hanging.  Point 3: the rank leaves
    BEHIND the first agreement (as after an RCCL call that returned an error): it never arrives at the next agreement point,
    where its peers wait -- the barrier is poisoned on its way out and they return too (round 5: they waited for ever)."""
    from x_maps_amd._native import XMapsNativeError
    xm_option("XM_SHARD_FAKE_RANKS", "4")
    xm_option("XM_SHARD_FAIL_AT", where)
    tb = S.make_tables(S.C_1M)
    evs = S.make_events(S.C_1M, frame=14, n=200_000)
    x, y, t, _ = S.to_soa(evs)
    with ShardedDevices(tb, devices=[0]) as sh:
        for tt in (t, t.astype(np.float64)):  # the columns exchange, then the keys
            with pytest.raises(XMapsNativeError, match="injected failure"):
                sh.process_frame(x, y, tt)
    xm_option("XM_SHARD_FAIL_AT", None)
    with ShardedDevices(tb, devices=[0]) as sh:  # (a fresh handle in the same process works)
        _check(tb, sh, evs)
