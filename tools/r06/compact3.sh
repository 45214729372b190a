#!/bin/bash
# round 6: which side of the compact frames costs the pipelined step?  0 = rectified, 2 = compact, 3 = K1 writes both / K2 compact, 4 = K1 writes both / K2 rectified
# (variants/libxmaps_cmp*.so = builds of tools/r06/patches/compact_frames.patch applied to the commit; cmp4: with -DXM_K2C_LATE=1)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
export XM_LIB=${1:-variants/libxmaps_cmp3.so}
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --steps 40 --warmup 10"
( for rep in 1 2 3; do for C in 0 2 3 4; do  # (3 / 4: K1 writes both frames, K2 reads the compact / the rectified one) for M in "" ; do
  python bench.py $Q $M --lib-option XM_COMPACT=$C 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('compact $C [$M] rep $rep: step ms', d.get('ms_per_step'), 'value', d.get('value'), d.get('roofline',{}).get('avg_launch_us'), 'exact', (d.get('parity') or {}).get('depth_bit_exact'), d.get('error'))"
done; done; done ) 2>&1 | tee gpurun_out/r06/compact3.txt
