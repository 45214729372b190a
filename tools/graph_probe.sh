#!/bin/bash
# rocprofv3 kernel trace of the 60 x 1M graph (multi-frame launches) with 60 and with 8 key frames -> gpurun_out/graph_probe/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/graph_probe; mkdir -p $OUT
for S in 60 8; do
  rocprofv3 --kernel-trace --stats -d $OUT -o s$S -- python bench.py --graph --slots $S --steps 600 --no-cpu-baseline > $OUT/s$S.json 2> $OUT/s$S.err
done
python bench.py --graph --no-cpu-baseline > $OUT/plain60.json 2> $OUT/plain60.err
python bench.py --graph --slots 8 --no-cpu-baseline > $OUT/plain8.json 2> $OUT/plain8.err
python bench.py --graph --slots 16 --no-cpu-baseline > $OUT/plain16.json 2> $OUT/plain16.err
python bench.py --graph --assume-sorted --no-cpu-baseline > $OUT/plain60_sorted.json 2> $OUT/plain60_sorted.err
XM_GRAPH_PER_FRAME=1 python bench.py --graph --slots 8 --no-cpu-baseline > $OUT/perframe8.json 2> $OUT/perframe8.err
ls $OUT
for f in $OUT/*.db; do python tools/rocprof_summary.py $f > ${f%.db}_summary.md 2>&1; done
ls $OUT | head -40
