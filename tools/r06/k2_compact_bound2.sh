#!/bin/bash
# as k2_compact_bound.sh, on a regular build with the emulation compiled in (variants/libxmaps_emuc.so: -DXM_K2P_EMU_COMPACT=1) against the product library
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --no-parity --steps 40 --warmup 10"
for rep in 1 2 3; do for L in "" variants/libxmaps_emuc.so; do for G in "" "--groups-in-flight 1"; do
  XM_LIB=$L python bench.py $Q $G 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('lib [$L] [$G] rep $rep: step ms', d.get('ms_per_step'), 'value', d.get('value'), d.get('roofline',{}).get('avg_launch_us'), d.get('error'))"
done; done; done 2>&1 | tee gpurun_out/r06/k2_compact_bound2.txt
