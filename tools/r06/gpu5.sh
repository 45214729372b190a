#!/bin/bash
# round 6, fifth GPU call: activity filter's first pass on the copy side (tests, soak, A/B), K2's output stores non-temporal (A/B), CPU baseline
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_activity.py tests/test_gpu_ingest.py tests/test_gpu_evt2.py tests/test_gpu_evt3.py tests/test_gpu_k2pipe.py tests/test_gpu_on_arrival.py -q -m gpu -x > gpurun_out/r06/t5.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06/t5.log; tail -6 gpurun_out/r06/t5.log
timeout 600 python tools/ingest_soak.py 0 600 > gpurun_out/r06/soak5.log 2>&1; tail -3 gpurun_out/r06/soak5.log
for rep in 1 2; do for OPT in "" "--lib-option XM_INGEST_ACT_ON_LAUNCH_SIDE=1"; do
  python bench.py --esl --no-cpu-baseline --no-other-modes --no-pmc --no-other-configs --steps 10 $OPT 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
sl=d.get('stream_legs') or {}
ch=sl.get('in_a_process_without_torch') or {}
print('[$OPT] rep $rep: ingest_path', d.get('ingest_path',{}).get('Mevents_per_s_end_to_end'), 'off', (sl.get('ingest_path_filter_off') or {}).get('Mevents_per_s_end_to_end'), 'fresh', (sl.get('ingest_path_fresh_arrays') or {}).get('Mevents_per_s_end_to_end'), 'evt3', (sl.get('from_evt3_words_period_chunks') or {}).get('Mevents_per_s_end_to_end'), 'default params', (sl.get('full_replay_through_processor_default_params') or {}).get('Mevents_per_s_end_to_end'), '| child ingest', (ch.get('ingest_path') or {}).get('Mevents_per_s_end_to_end'), 'child default', (ch.get('full_replay_through_processor_default_params') or {}).get('Mevents_per_s_end_to_end'), d.get('error'))"
done; done 2>&1 | tee gpurun_out/r06/act_copy_side_ab.txt
Q="--esl --no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --steps 20"
for rep in 1 2; do for L in "" variants/libxmaps_nt.so; do
  XM_LIB=$L python bench.py $Q 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('lib [$L] rep $rep: K0b/K1/K2 us', d.get('roofline',{}).get('avg_launch_us'), 'step ms', d.get('ms_per_step'), 'bgr-only', (d.get('other_modes') or {}).get('groups_bgr_only',{}).get('ms_per_step'), d.get('error'))"
done; done 2>&1 | tee gpurun_out/r06/k2_nt_ab.txt
XM_LIB=variants/libxmaps_nt.so python bench.py --no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --steps 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C-1M nt: K0b/K1/K2 us', d.get('roofline',{}).get('avg_launch_us'), 'step ms', d.get('ms_per_step'), d.get('error'))" | tee -a gpurun_out/r06/k2_nt_ab.txt
python bench.py --no-other-modes --no-host-path --no-pmc --no-other-configs --steps 20 --cpu-seconds 6 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C-1M: K0b/K1/K2 us', d.get('roofline',{}).get('avg_launch_us'), 'step ms', d.get('ms_per_step'), 'cpu', json.dumps(d.get('cpu_baseline'))[:600])" | tee gpurun_out/r06/cpu_baseline.txt
