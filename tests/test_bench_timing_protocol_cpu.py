"""bench.py's timing protocol (benchmodes/common.py: Timer) without a GPU: every timed block is bracketed by
[engine sync + torch.cuda.synchronize + barrier] in front and [engine sync + torch.cuda.synchronize] behind, a block runs EXACTLY the
steps it was given, the per-block time is the MAX over ranks, and every rank derives the same number of blocks from an agreed
estimate.  Fakes stand in for torch / torch.distributed and record the order of calls."""
import numpy as np

from benchmodes.common import Timer, n_blocks_for


class _Log(list):
    def add(self, what):
        self.append(what)


class _FakeCuda:
    def __init__(self, log):
        self.log = log

    def synchronize(self):
        self.log.add("cuda.synchronize")


class _FakeTensor:
    def __init__(self, values):
        self.v = np.array(values, dtype=np.float64)

    def item(self):
        return float(self.v[0])

    def cpu(self):
        return self.v


class _FakeTorch:
    float64 = "f64"

    def __init__(self, log):
        self.cuda = _FakeCuda(log)

    def tensor(self, values, dtype=None, device=None):
        return _FakeTensor(values)


class _FakeDist:
    class ReduceOp:
        MAX = "max"

    def __init__(self, log, other_rank_scale):
        self.log, self.scale = log, other_rank_scale

    def barrier(self):
        self.log.add("dist.barrier")

    def all_reduce(self, t, op=None):
        assert op == self.ReduceOp.MAX
        self.log.add("dist.all_reduce(MAX)")
        t.v = np.maximum(t.v, t.v * self.scale)  # (another rank that was `scale` times slower)


def test_blocks_are_bracketed_and_run_exactly_k_steps():
    log = _Log()
    tm = Timer(_FakeTorch(log), _FakeDist(log, 3.0), "dev", lambda: log.add("engine.sync"))
    steps_run = []
    K, R = 7, 4

    def run_block():
        log.add("block")
        steps_run.append(K)

    el, enq = tm.blocks(run_block, R)
    assert steps_run == [K] * R and len(el) == len(enq) == R
    per_block = ["engine.sync", "cuda.synchronize", "dist.barrier", "block", "engine.sync", "cuda.synchronize"]
    assert list(log) == per_block * R + ["dist.barrier", "dist.all_reduce(MAX)"]
    assert (enq <= el / 3.0 + 1e-9).all()  # `el` is the MAX over ranks (the fake other rank is 3 x slower), the enqueue time is this rank's


def test_single_rank_needs_no_collective():
    log = _Log()
    tm = Timer(_FakeTorch(log), None, "dev", lambda: log.add("engine.sync"))
    el, _ = tm.blocks(lambda: log.add("block"), 2)
    assert "dist.barrier" not in log and list(log).count("block") == 2 and len(el) == 2
    assert tm.agree(0.123) == 0.123


def test_every_rank_derives_the_same_number_of_blocks():
    log = _Log()
    tm = Timer(_FakeTorch(log), _FakeDist(log, 2.0), "dev", lambda: None)
    est = tm.agree(1e-4)  # MAX over ranks of the pre-warm's seconds per step
    assert est == 2e-4

    class A:
        single_block, steps = False, 20
    assert n_blocks_for(A, est) == int(min(400, max(3, round(0.30 / (20 * est)))))
    A.single_block = True
    assert n_blocks_for(A, est) == 1
