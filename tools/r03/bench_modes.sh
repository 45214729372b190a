#!/bin/bash
# round 3: every bench mode once (short), JSON lines into gpurun_out/r03_bench/
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_bench; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
export XM_BENCH_PREWARM_S=${XM_BENCH_PREWARM_S:-0.2}
Q="--no-cpu-baseline"
timeout 300 python bench.py --esl $Q > $OUT/esl.json 2> $OUT/esl.err; tail -c 600 $OUT/esl.err
timeout 300 python bench.py --esl --batch 0 $Q --no-host-path > $OUT/esl_single.json 2> $OUT/esl_single.err
timeout 300 python bench.py --graph $Q > $OUT/graph.json 2> $OUT/graph.err; tail -c 300 $OUT/graph.err
timeout 300 python bench.py --sharded $Q > $OUT/sharded.json 2> $OUT/sharded.err; tail -c 300 $OUT/sharded.err
timeout 300 python bench.py $Q --no-host-path > $OUT/default.json 2> $OUT/default.err; tail -c 300 $OUT/default.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_bench/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "NO JSON", e); continue
    r = d.get("roofline") or {}
    print(os.path.basename(f), "value", d.get("value"), "ms/step", d.get("ms_per_step"), "| roofline:", r.get("kernel"), "frac", r.get("frac"),
          "counter", r.get("frac_counter_bytes"), "evstream", r.get("event_stream_read_roofline_frac"), "| us/frame",
          {k: v.get("us_per_frame") for k, v in (r.get("kernels") or {}).items()}, "| other", {k: v.get("value") for k, v in (d.get("other_modes") or {}).items() if isinstance(v, dict)})
PY
