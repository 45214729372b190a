#!/bin/bash
# K2's duration for groups of F frames with one / two pixels per thread (XM_K2_PPT): where should k2_ppt() switch?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for F in 1 2 3 4 6 8 16; do
  for P in 1 2; do
    OUT=gpurun_out/pptthr_${F}_$P; mkdir -p $OUT
    XM_K2_PPT=$P timeout 100 rocprofv3 --kernel-trace --stats -d $OUT -o t -- python tools/batch_probe.py $F 12 2 > $OUT/log 2>&1
    python tools/rocprof_summary.py $OUT/t_results.db 2>/dev/null | grep -E "k_frame_proj" | awk -v F=$F -v P=$P -F'|' '{printf "F=%s ppt=%s %s calls=%s avg=%s min=%s  per frame %.2f\n", F, P, $2, $3, $4, $5, $4/F}'
    rm -rf $OUT
  done
done
for P in 1 2; do
  XM_K2_PPT=$P timeout 150 python bench.py --no-cpu-baseline --no-other-modes --no-host-path --batch 0 --steps 2000 --warmup 200 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one frame per call, XM_K2_PPT=$P', d['value'], d['roofline']['avg_launch_us'])"
done
