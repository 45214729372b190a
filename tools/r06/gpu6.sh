#!/bin/bash
# round 6 (second session), first GPU call: the whole GPU suite with durations, the driver's default line, the ESL-like line, the
# tile_stream ubench -- the state this session starts from (the earlier session's gpurun_out/ was lost with its container)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
hipcc --offload-arch=gfx950 -O3 tools/ubench/tile_stream.hip -o tools/ubench/tile_stream 2>/dev/null
( time timeout 1500 python -m pytest tests -q -m gpu -x --durations=25 ) > gpurun_out/r06/t6.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06/t6.log; tail -40 gpurun_out/r06/t6.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench6.json 2> gpurun_out/r06/bench6.err; echo "bench rc $?"
timeout 600 python bench.py --esl --steps 20 > gpurun_out/r06/esl6.json 2> gpurun_out/r06/esl6.err; echo "esl rc $?"
tools/ubench/tile_stream > gpurun_out/r06/tile_stream.txt 2>&1; tail -30 gpurun_out/r06/tile_stream.txt
python - <<'PY'
import json
for fn in ("bench6", "esl6"):
    try:
        d = json.loads(open(f"gpurun_out/r06/{fn}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(fn, "unreadable", e); continue
    print(fn, "value", d["value"], d["unit"], "ms/step", d["ms_per_step"], "roofline", json.dumps(d["roofline"])[:700])
    sl = d.get("stream_legs") or (d.get("other_configs") or {}).get("esl", {}).get("stream_legs") or {}
    for k, v in sl.items():
        if isinstance(v, dict) and "Mevents_per_s_end_to_end" in v:
            print("  ", k, v["Mevents_per_s_end_to_end"], v.get("ms_per_cut_frame", v.get("ms_per_shown_frame")))
    ch = sl.get("in_a_process_without_torch", {})
    for k, v in ch.items():
        print("   child", k, json.dumps(v)[:300])
    print("  other_modes", json.dumps(d.get("other_modes"))[:1500])
PY
