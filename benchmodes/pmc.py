"""HBM traffic of the default workload's kernels measured IN the bench run (roofline.traffic), when rocprofv3 is on the box.

Two short child runs of bench.py under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (one counter group per
run, --kernel-trace only: the combination the GPU pool allows and /opt/skills/guides/MI355X_MICROARCH.md prescribes), one group
of 32 frames at a time; per launch: HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB -- FETCH_SIZE reports half the bytes of a wide
coalesced read on gfx950 (the guide's correction).  Anything that goes wrong (no rocprofv3, a timeout, an unreadable database)
returns None and bench.py falls back to the committed profiles/pmc_traffic.json, saying so and how old that file is.
"""
from __future__ import annotations

import glob
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time


def _logical(name: str):
    if "k_minmax" in name or "k_cols_bounds" in name:
        return "k_minmax"
    if "k_scatter" in name:
        return "k_scatter"
    if "k_frame" in name:
        return "k_frame"
    return None


def measure_traffic(bench_py: str, child_flags: list[str], timeout_s: float = 150.0):
    """-> ({logical kernel: HBM bytes per launch}, {details}) or (None, {"error": ...})"""
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, {"error": "rocprofv3 not found"}
    t0 = time.perf_counter()
    tmp = tempfile.mkdtemp(prefix="xm_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", XM_BENCH_PREWARM_S="0.05")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    vals = {}
    try:
        for cn in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--kernel-trace", "--pmc", cn, "-d", tmp, "-o", "pmc_" + cn, "--", sys.executable, bench_py] + child_flags
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            dbs = glob.glob(os.path.join(tmp, "**", f"pmc_{cn}*_results.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None, {"error": f"rocprofv3 --pmc {cn} failed (rc {r.returncode}): {(r.stderr or '')[-200:]}"}
            for k, c, v in sqlite3.connect(dbs[0]).execute("select kernel_name, counter_name, avg(value) from counters_collection group by 1, 2"):
                lg = _logical(k) if "xm::" in k and "batch" in k or "pipe" in k else None
                if lg and c == cn:
                    key = (lg, k.split("(")[0].replace("void ", ""))
                    vals.setdefault(key, {})[cn] = float(v)
    except Exception as e:  # (timeout, sqlite, ...)
        return None, {"error": repr(e)[:200]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out, kern = {}, {}
    for (lg, name), d in vals.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            b = int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024)
            if b > out.get(lg, -1):
                out[lg], kern[lg] = b, {"kernel": name, "FETCH_SIZE_KB": round(d["FETCH_SIZE"], 1), "WRITE_SIZE_KB": round(d["WRITE_SIZE"], 1)}
    if not out:
        return None, {"error": "no counters for the workload's kernels in the databases"}
    return out, {"kernels": kern, "seconds": round(time.perf_counter() - t0, 1),
                 "source": "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate child runs of bench.py, "
                           "one group of 32 frames at a time), HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB (gfx950 FETCH_SIZE "
                           "correction, MI355X_MICROARCH.md)"}
