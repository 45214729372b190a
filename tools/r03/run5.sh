#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_bench; mkdir -p $OUT
bash tools/r03/gputests.sh tests/test_gpu_adaptive.py tests/test_gpu_api.py tests/test_gpu_bench_contract.py
XM_BENCH_PREWARM_S=0.2 timeout 300 python bench.py --no-cpu-baseline --no-host-path > $OUT/def_modes.json 2> $OUT/def_modes.err; tail -c 300 $OUT/def_modes.err
XM_BENCH_PREWARM_S=0.2 timeout 300 python bench.py --batch 0 --no-cpu-baseline --no-host-path --no-other-modes > $OUT/single_adaptive.json 2> $OUT/single_adaptive.err; tail -c 300 $OUT/single_adaptive.err
python - <<'PY'
import json, os
R = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_bench/"
for f in ("def_modes.json", "single_adaptive.json"):
    d = json.loads(open(R + f).read().strip().splitlines()[-1])
    print(f, "value", d["value"], "paths", d["config"]["k1_paths_frames"], "other", {k: (v.get("value"), v.get("k1_paths")) for k, v in (d.get("other_modes") or {}).items() if isinstance(v, dict)})
PY
