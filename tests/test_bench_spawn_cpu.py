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
