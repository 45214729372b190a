#!/bin/bash
# round 6, first GPU call: the default call pattern (device ingest + owned result buffers) -- tests, then the ESL-like bench legs
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_evt2.py tests/test_gpu_evt3.py tests/test_gpu_api.py tests/test_gpu_configs.py tests/test_gpu_activity.py -x -q -m gpu > gpurun_out/r06/t1.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06/t1.log
tail -5 gpurun_out/r06/t1.log
timeout 600 python bench.py --esl --steps 20 --no-cpu-baseline --no-other-modes > gpurun_out/r06/esl1.json 2> gpurun_out/r06/esl1.err
echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/esl1.json").read().strip().splitlines()[-1])
sl = d["stream_legs"]
for k, v in sl.items():
    if isinstance(v, dict) and "Mevents_per_s_end_to_end" in v:
        print(k, v["Mevents_per_s_end_to_end"], v.get("ms_per_cut_frame", v.get("ms_per_shown_frame")), v.get("same_frames_as_host_trigger_finder", v.get("same_frames_as_host_path")), v.get("passes_ms"))
ch = sl.get("in_a_process_without_torch", {})
for k, v in ch.items():
    print("child", k, v)
print("value", d["value"], d["ms_per_step"])
PY
