#!/usr/bin/env python3
"""One-off soak: the randomized differential tests (tests/test_gpu_fuzz.py, tests/test_gpu_cols.py, tests/test_gpu_own.py) over many more seeds than the
suite runs:  python tools/fuzz_soak.py [first_seed=120] [n_seeds=1000] [XM_OPTION=value ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as F
import test_gpu_cols as C
import test_gpu_own as W
first = int(sys.argv[1]) if len(sys.argv) > 1 else 120
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
from x_maps_amd import _native as _N; _N.debug_option("XM_COLS", "2")
for kv in sys.argv[3:]:  # library variant switches (e.g. XM_OWN_ROW_PASSES=3)
    _N.debug_option(*kv.split("=", 1))
t0 = time.time()
bad, skipped = [], 0
for seed in range(first, first + n):
    for name, fn in (("fuzz", F.test_random_tables_and_streams), ("cols", C.test_random_rigs_and_streams_on_the_tiles),
                     ("own", W.test_random_shared_cell_rigs_and_streams)):
        try:
            fn(seed)
        except ValueError as e:
            if "high - low" in repr(e):  # the case generator drew an empty range for this seed: not a case
                skipped += 1
                continue
            bad.append((name, seed, repr(e)[:200]))
            print("FAILED", name, seed, repr(e)[:300], flush=True)
        except BaseException as e:  # noqa
            bad.append((name, seed, repr(e)[:200]))
            print("FAILED", name, seed, repr(e)[:300], flush=True)
print(f"{n} seeds x 3 tests in {time.time() - t0:.1f} s, failures: {len(bad)}, seeds the generator rejected: {skipped}")
sys.exit(1 if bad else 0)
