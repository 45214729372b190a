#!/bin/bash
# round 6: what would a COMPACT disparity frame buy?  XM_ABLATE bit 21 (variants/libxmaps_abl.so, -DXM_ABLATE) makes K2's patch loader read 41 % of a
# patch's quads, contiguous per tile -- the traffic of a frame that stores only the cells an X-map column can reach.  Results are wrong on purpose (--no-parity).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --no-parity --steps 40 --warmup 10"
for rep in 1 2 3; do for A in 0 2097152; do for G in "" "--groups-in-flight 1"; do
  XM_LIB=variants/libxmaps_abl.so python bench.py $Q $G --lib-option XM_ABLATE=$A 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ablate $A [$G] rep $rep: step ms', d.get('ms_per_step'), 'value', d.get('value'), d.get('roofline',{}).get('avg_launch_us'), d.get('error'))"
done; done; done 2>&1 | tee gpurun_out/r06/k2_compact_bound.txt
