#!/bin/bash
# round 6, second GPU call: new tests (ingest defaults, on-arrival), then the whole GPU suite, then the ESL bench legs
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_on_arrival.py -x -q -m gpu > gpurun_out/r06/t2a.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06/t2a.log; tail -4 gpurun_out/r06/t2a.log
timeout 1500 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/r06/t2.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06/t2.log; tail -40 gpurun_out/r06/t2.log
timeout 600 python bench.py --esl --steps 20 --no-cpu-baseline --no-other-modes > gpurun_out/r06/esl2.json 2> gpurun_out/r06/esl2.err
echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/esl2.json").read().strip().splitlines()[-1])
sl = d["stream_legs"]
for k, v in sl.items():
    if isinstance(v, dict) and "Mevents_per_s_end_to_end" in v:
        print(k, v["Mevents_per_s_end_to_end"], v.get("ms_per_cut_frame", v.get("ms_per_shown_frame")), v.get("same_frames_as_host_trigger_finder", v.get("same_frames_as_host_path")), v.get("passes_ms"), v.get("result_buffer_pool"))
ch = sl.get("in_a_process_without_torch", {})
for k, v in ch.items():
    print("child", k, v)
PY
