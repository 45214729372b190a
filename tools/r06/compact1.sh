#!/bin/bash
# round 6, compact frames stage 1: K1's column tiles write compact frames (XM_COMPACT=1), k_cols_expand restores the rectified frame in front of K2 -- parity, and what K1 alone gains
# (variants/libxmaps_cmp*.so = builds of tools/r06/patches/compact_frames.patch applied to the commit; cmp4: with -DXM_K2C_LATE=1)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
export XM_LIB=variants/libxmaps_cmp1.so
( XM_COMPACT_ENV=1 timeout 900 python -m pytest tests/test_gpu_cols.py tests/test_gpu_fused.py tests/test_gpu_configs.py tests/test_gpu_adaptive.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2
timeout 600 python tools/fuzz_soak.py 9000 150 XM_COMPACT=1 2>&1 | tail -1
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --steps 40 --warmup 10"
for rep in 1 2 3; do for C in 0 1; do for M in "--groups-in-flight 1"; do
  python bench.py $Q $M --lib-option XM_COMPACT=$C 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('compact $C [$M] rep $rep: step ms', d.get('ms_per_step'), 'value', d.get('value'), d.get('roofline',{}).get('avg_launch_us'), 'exact', (d.get('parity') or {}).get('depth_bit_exact'), d.get('error'))"
done; done; done ) 2>&1 | tee gpurun_out/r06/compact1.txt
