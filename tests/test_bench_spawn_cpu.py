"""bench.py's launch logic without a GPU (XM_BENCH_DRY=1: the ranks meet over gloo and do no GPU work): `--gpus N` without a
launcher around it re-executes itself under torch.distributed.run with N ranks; a launcher whose rank count disagrees with
--gpus is an error, never a silent N = 1 run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = dict(os.environ, XM_BENCH_DRY="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_spawns_two_ranks_by_itself():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["spawned_by_bench"] is True
    assert line["steps"] == 3 and line["warmup"] == 1


def test_one_gpu_needs_no_launcher():
    r = _run([])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["spawned_by_bench"] is False


def test_a_launcher_with_another_rank_count_is_an_error():
    r = _run(["--gpus", "4"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0
    assert "error" in json.loads(r.stdout.strip().splitlines()[-1])


def test_more_ranks_than_gpus_fails_loudly():
    """without XM_BENCH_DRY: a box with fewer GPUs than --gpus (this container has none) gets an error line and a non-zero exit"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "XM_BENCH_DRY"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0
    err = json.loads(r.stdout.strip().splitlines()[-1])
    assert err["n_gpus_requested"] == 64 and err["n_gpus_visible"] < 64


# ---- the guard around the leg in which the ranks wait for one another (benchmodes/guard.py) -------------------------------------
def _last_json(r):
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert lines, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    return json.loads(lines[-1])


def test_leg_behind_the_replicas_runs_under_the_guard():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"XM_BENCH_DRY_LEG": "ok", "XM_BENCH_LEG_TIMEOUT_S": "60"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["other_modes"]["one_frame_sharded_over_the_ranks"] == {"dry_leg": "ok"}


def test_a_rank_that_hangs_inside_the_leg_does_not_cost_the_line():
    """rank 1 goes to sleep between two collectives of the leg: rank 0 waits in the second one; after XM_BENCH_LEG_TIMEOUT_S the
    guard prints the replicas' line with the leg's error in it and every rank leaves"""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"XM_BENCH_DRY_LEG": "hang:1", "XM_BENCH_LEG_TIMEOUT_S": "6"}, timeout=120)
    assert time.time() - t0 < 90
    line = _last_json(r)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["steps"] == 3
    assert "timeout" in line["other_modes"]["one_frame_sharded_over_the_ranks"]["error"]
    assert r.returncode == 0, r.stderr[-2000:]


def test_a_rank_that_dies_inside_the_leg_does_not_cost_the_line():
    """rank 1 leaves the process mid-leg: the launcher terminates rank 0 (SIGTERM) or its collective fails -- either way rank 0
    has printed the replicas' line, with the leg's error, before it goes"""
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"XM_BENCH_DRY_LEG": "die:1", "XM_BENCH_LEG_TIMEOUT_S": "30"}, timeout=180)
    line = _last_json(r)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2
    assert "error" in line["other_modes"]["one_frame_sharded_over_the_ranks"]
