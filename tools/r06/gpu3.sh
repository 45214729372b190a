#!/bin/bash
# round 6, third GPU call: the failing tests again; tile_stream ubench (headline K1's access-pattern floor); K2 on the ESL rig with parts
# switched off (variants/libxmaps_abl.so = -DXM_ABLATE build: results wrong on purpose)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_on_arrival.py tests/test_gpu_api.py tests/test_gpu_configs.py tests/test_gpu_sharded_cabi.py -q -m gpu > gpurun_out/r06/t3.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06/t3.log; tail -12 gpurun_out/r06/t3.log
tools/ubench/tile_stream > gpurun_out/r06/tile_stream.txt 2>&1; cat gpurun_out/r06/tile_stream.txt
Q="--esl --no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --no-parity --groups-in-flight 1 --steps 10 --warmup 2"
for A in 0 4096 8192 12288; do for NB in "" "--no-bgr"; do
  XM_LIB=variants/libxmaps_abl.so python bench.py $Q $NB --lib-option XM_ABLATE=$A 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ablate $A $NB : K0b/K1/K2 us', d.get('roofline',{}).get('avg_launch_us'), 'step ms', d.get('ms_per_step'), 'bgr-only', (d.get('other_modes') or {}).get('groups_bgr_only',{}).get('ms_per_step'), d.get('error'))"
done; done 2>&1 | tee gpurun_out/r06/k2_ablate.txt
python bench.py --esl --no-cpu-baseline --no-host-path --no-pmc --no-other-configs --steps 20 2>/dev/null | tail -1 > gpurun_out/r06/esl3.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/esl3.json").read())
print("esl value", d["value"], d["ms_per_step"], d["roofline"].get("avg_launch_us"), (d.get("other_modes") or {}).get("groups_bgr_only"))
PY
