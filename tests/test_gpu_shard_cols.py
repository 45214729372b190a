"""-m gpu: shards on the column tiles (xm_shard_cols_*: every time column processed by ONE rank, plain u16 frames merged by SUM)
with all ranks played by one GPU -- the collectives are done by hand on the ranks' buffers (the send buffers packed into one
gathered buffer, SUM of the u16 frames as packed int32 pairs), exactly what RCCL computes.  Result == the C oracle for
C-10M in 1 / 2 / 4 / 8 shards, two frames in a row (no stale cells), C-1M with lopsided shards; shards that cannot be handled
(without events, inside one column, not sorted) raise the flag instead of producing a wrong frame."""
import numpy as np
import pytest

import xmaps_oracle as O
from c_oracle import COracle
from x_maps_amd import XMapsEngine
from x_maps_amd import synthetic as S
from x_maps_amd.sharded import shard_bounds

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _run(eng, dev, shards, n_frame, out_shape, finish=True):
    """shards: list of (x, y, t) numpy pieces of ONE frame in order.  Returns ([(depth, bgr)] of two runs, failed)."""
    info = eng.shard_cols_info(n_frame)
    assert info is not None
    cap, W, sb = info["cap_events"], len(shards), info["send_bytes"]
    stream = torch.cuda.ExternalStream(eng.stream(0), device=dev)
    P = lambda buf: buf.data_ptr() + (cap + 8) * buf.element_size()  # the shard's first own event (an empty slice has no data_ptr)
    res, frames = [], []
    for (x, y, t) in shards:
        r = []
        for a, dt in ((x.view(np.int16), torch.int16), (y.view(np.int16), torch.int16), (t, torch.int64)):
            buf = torch.zeros(cap + 8 + len(a) + 8, dtype=dt, device=dev)  # headroom | own events | slack
            buf[cap + 8:cap + 8 + len(a)] = torch.from_numpy(a).to(dev)
            r.append(buf)
        res.append(r)
        frames.append(torch.zeros(info["frame_bytes"], dtype=torch.uint8, device=dev))
    gathered = torch.zeros(W * sb, dtype=torch.uint8, device=dev)
    depth = torch.zeros(out_shape, dtype=torch.float32, device=dev)
    bgr = torch.zeros(out_shape + (3,), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    outs = []
    for rep in range(2):  # two frames in a row on the same buffers: every cell is rewritten, nothing stale
        with torch.cuda.stream(stream):
            for r in range(W):  # every rank packs straight into its slot of the gathered buffer = the all-gather
                eng.shard_cols_pack(P(res[r][0]), P(res[r][1]), P(res[r][2]), len(shards[r][2]), gathered[r * sb:].data_ptr(), cap)
            for r in range(W):
                eng.shard_cols_scatter(P(res[r][0]), P(res[r][1]), P(res[r][2]), len(shards[r][2]), n_frame, gathered.data_ptr(), sb, r, W,
                                       cap, frames[r].data_ptr())
            nred = info["reduce_u32"] * 4
            merged = frames[0].clone()
            acc = merged[:nred].view(torch.int32)
            for r in range(1, W):
                acc += frames[r][:nred].view(torch.int32)  # = all_reduce(SUM)
            if finish:
                eng.shard_finish_u16(merged.data_ptr(), depth.data_ptr(), bgr.data_ptr())
        eng.sync()
        torch.cuda.synchronize()
        outs.append((depth.cpu().numpy().copy(), bgr.cpu().numpy().copy()))
    failed = eng.shard_cols_failed()
    return outs, failed


@pytest.fixture(scope="module")
def c10m():
    cfg = S.C_10M
    tb = S.make_tables(cfg)
    x, y, t, _ = S.to_soa(S.make_events(cfg))
    ref = COracle(tb, False, omp=True).process_ev_frame(x, y, t)
    ref = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in ref.items()}
    return cfg, tb, (x, y, t), ref


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_c10m_in_column_shards_on_one_gpu(c10m, world):
    cfg, tb, (x, y, t), ref = c10m
    dev = torch.device("cuda", 0)
    n = len(t)
    shards = [tuple(a[lo:hi] for a in (x, y, t)) for lo, hi in (shard_bounds(n, r, world) for r in range(world))]
    with XMapsEngine(tb) as eng:
        outs, failed = _run(eng, dev, shards, n, (cfg.proj_h, cfg.proj_w))
    assert not failed
    for depth, bgr in outs:
        assert np.array_equal(depth, ref["depth"]) and np.array_equal(bgr, ref["bgr"]), world


def test_c1m_lopsided_shards_and_ties_on_the_cut():
    """shards of very different sizes, cuts placed INSIDE runs of equal time stamps (C-1M has ~77 events per microsecond: the
    last column of a shard always goes on in the next one)"""
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    x, y, t, _ = S.to_soa(S.make_events(cfg, frame=4))
    ref = O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t)
    n = len(t)
    dev = torch.device("cuda", 0)
    for cuts in ([0, n // 10, n // 10 + 5000, (n * 65) // 100, n], [0, 3000, n - 3000, n], [0, n // 2 + 1, n]):
        shards = [tuple(a[lo:hi] for a in (x, y, t)) for lo, hi in zip(cuts[:-1], cuts[1:])]
        with XMapsEngine(tb) as eng:
            outs, failed = _run(eng, dev, shards, n, (cfg.proj_h, cfg.proj_w))
        assert not failed, cuts
        for depth, bgr in outs:
            assert np.array_equal(depth, ref["depth"]) and np.array_equal(bgr, ref["bgr"]), cuts


def test_pieces_that_cannot_be_handled_raise_the_flag():
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    x, y, t, _ = S.to_soa(S.make_events(cfg, frame=2))
    n = len(t)
    dev = torch.device("cuda", 0)
    cases = {
        "a shard without events": [0, n // 3, n // 3, n],
        "a shard inside one time column": [0, n // 2, n // 2 + 200, n],
    }
    for name, cuts in cases.items():
        shards = [tuple(a[lo:hi] for a in (x, y, t)) for lo, hi in zip(cuts[:-1], cuts[1:])]
        with XMapsEngine(tb) as eng:
            _, failed = _run(eng, dev, shards, n, (cfg.proj_h, cfg.proj_w))
        assert failed, name
    # a stream that is not sorted: the per-event verification of the tiles objects
    xs, ys, ts = x.copy(), y.copy(), t.copy()
    ts[n // 4:n // 4 + 5000] = ts[n // 4:n // 4 + 5000][::-1] + 4000
    shards = [tuple(a[lo:hi] for a in (xs, ys, ts)) for lo, hi in ((0, n // 2), (n // 2, n))]
    with XMapsEngine(tb) as eng:
        _, failed = _run(eng, dev, shards, n, (cfg.proj_h, cfg.proj_w))
        assert failed
        # the flag is cleared by the check: a good frame afterwards passes
        shards = [tuple(a[lo:hi] for a in (x, y, t)) for lo, hi in ((0, n // 2), (n // 2, n))]
        outs, failed = _run(eng, dev, shards, n, (cfg.proj_h, cfg.proj_w))
        assert not failed
    ref = O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t)
    assert np.array_equal(outs[1][0], ref["depth"])


def test_rigs_that_do_not_take_the_tiles_say_so():
    tb = S.make_tables_shared_cells()  # several time columns per cell: owner tiles, not this path
    with XMapsEngine(tb) as eng:
        assert eng.shard_cols_info(100_000) is None
