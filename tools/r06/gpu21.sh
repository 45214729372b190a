#!/bin/bash
# round 6: the cells' turns with empty packets in between (tests, soak); the driver's line five times: how its ESL-like ingest leg spreads
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_activity.py tests/test_gpu_ingest.py tests/test_gpu_evt3.py tests/test_gpu_evt2.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python tools/ingest_soak.py 3000 800 2>&1 | tail -2
for rep in 1 2 3 4 5; do
  timeout 400 python bench.py --steps 20 --warmup 5 --cpu-seconds 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
e=d.get('other_configs',{}).get('esl',{})
print('default line rep $rep: value', d['value'], 'esl', e.get('value'), 'esl ingest_path', e.get('ingest_path',{}).get('Mevents_per_s_end_to_end'), 'off', e.get('ingest_path_filter_off',{}).get('Mevents_per_s_end_to_end'), 'default params', e.get('full_replay_through_processor_default_params',{}).get('Mevents_per_s_end_to_end'), 'child', e.get('in_a_process_without_torch',{}).get('ingest_path',{}).get('Mevents_per_s_end_to_end'))"
done 2>&1 | tee gpurun_out/r06/default_line_ingest.txt
