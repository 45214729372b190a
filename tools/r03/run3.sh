#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_bench; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_cols.py tests/test_gpu_own.py tests/test_gpu_fused.py tests/test_gpu_a4_bruteforce.py tests/test_gpu_configs.py tests/test_gpu_sparse_groups.py -x -q 2>&1 | tail -5
for pipe in 1 0; do
XM_K2_PIPE=$pipe XM_BENCH_PREWARM_S=0.2 timeout 300 python bench.py --no-cpu-baseline --no-host-path --no-other-modes > $OUT/def_pipe$pipe.json 2> $OUT/def_pipe$pipe.err; tail -c 200 $OUT/def_pipe$pipe.err
XM_K2_PIPE=$pipe XM_BENCH_PREWARM_S=0.2 timeout 300 python bench.py --esl --no-cpu-baseline --no-host-path --no-other-modes > $OUT/esl_pipe$pipe.json 2> $OUT/esl_pipe$pipe.err; tail -c 200 $OUT/esl_pipe$pipe.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_bench/*_pipe*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "NO JSON", e); continue
    r = d.get("roofline") or {}
    print(os.path.basename(f), "value", d.get("value"), "ms/step", d.get("ms_per_step"), "| us/frame", {k: v.get("us_per_frame") for k, v in (r.get("kernels") or {}).items()}, d.get("parity"))
PY
