#!/bin/bash
# round 6: the whole GPU suite, smoke(), the driver's line and the ESL-like line on the round's final build
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
( time timeout 1500 python -m pytest tests -q -m gpu -x --durations=40 ) > gpurun_out/r06/t16.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06/t16.log; tail -24 gpurun_out/r06/t16.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/smoke16.log 2>&1; echo "smoke rc $?"; tail -4 gpurun_out/r06/smoke16.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench16.json 2> gpurun_out/r06/bench16.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench16.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "cb", d["roofline"].get("frac_counter_bytes"), d["roofline"]["avg_launch_us"])
e = d["other_configs"]["esl"]
print("esl", e["value"], e["ms_per_step"], "ingest", e["ingest_path"]["Mevents_per_s_end_to_end"], "default params", e["full_replay_through_processor_default_params"]["Mevents_per_s_end_to_end"], e["full_replay_through_processor_default_params"].get("ms_per_shown_frame"))
print("graph60", d["other_configs"]["graph60"]["value"], "sharded", d["other_configs"]["sharded_c10m"]["value"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores_c_openmp"]["value"])
PY
