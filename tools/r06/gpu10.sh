#!/bin/bash
# round 6: the ingest with the next packet's k_act_first riding on k_ing_count (tests, soak, A/B); K2 on the ESL-like rig with parts switched off
# (the K2 ablation legs need variants/libxmaps_abl.so: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DXM_ABLATE x_maps_amd/csrc/xmaps_hip.hip -o variants/libxmaps_abl.so)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_activity.py tests/test_gpu_ingest.py tests/test_gpu_evt2.py tests/test_gpu_evt3.py tests/test_gpu_on_arrival.py tests/test_gpu_configs.py -q -m gpu -x > gpurun_out/r06/t10.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06/t10.log; tail -6 gpurun_out/r06/t10.log
timeout 600 python tools/ingest_soak.py 0 500 > gpurun_out/r06/soak10.log 2>&1; tail -3 gpurun_out/r06/soak10.log
for rep in 1 2 3; do for OPT in "" "--lib-option XM_INGEST_ACT_FUSE=0"; do
  python bench.py --esl --no-cpu-baseline --no-other-modes --no-pmc --no-other-configs --steps 10 $OPT 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
sl=d.get('stream_legs') or {}
ch=sl.get('in_a_process_without_torch') or {}
ip=d.get('ingest_path',{})
print('[$OPT] rep $rep: ingest_path', ip.get('Mevents_per_s_end_to_end'), 'passes', ip.get('passes_ms'), 'off', (sl.get('ingest_path_filter_off') or {}).get('Mevents_per_s_end_to_end'), 'fresh', (sl.get('ingest_path_fresh_arrays') or {}).get('Mevents_per_s_end_to_end'), 'evt3', (sl.get('from_evt3_words_period_chunks') or {}).get('Mevents_per_s_end_to_end'), 'default params', (sl.get('full_replay_through_processor_default_params') or {}).get('Mevents_per_s_end_to_end'), 'paced p50/p99', ((sl.get('paced') or {}).get('real_time') or {}).get('push_to_frame_visible_ms'), '| child ingest', (ch.get('ingest_path') or {}).get('Mevents_per_s_end_to_end'), 'child off', (ch.get('ingest_path_filter_off') or {}).get('Mevents_per_s_end_to_end'), 'child default', (ch.get('full_replay_through_processor_default_params') or {}).get('Mevents_per_s_end_to_end'), d.get('error'))"
done; done 2>&1 | tee gpurun_out/r06/act_fuse_ab.txt
Q="--esl --no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --no-parity --groups-in-flight 1 --steps 10 --warmup 2"
for A in 0 4096 8192 12288 262144 524288 1048576 1835008 1847296; do for NB in "" "--no-bgr"; do
  XM_LIB=variants/libxmaps_abl.so python bench.py $Q $NB --lib-option XM_ABLATE=$A 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ablate $A $NB : K1a/K1b/K2 us', d.get('roofline',{}).get('avg_launch_us'), 'step ms', d.get('ms_per_step'), d.get('error'))"
done; done 2>&1 | tee gpurun_out/r06/k2_ablate.txt
