#!/bin/bash
# round 3: owner tiles on the ESL-like rig -- parity tests, then group / single throughput with and without the tiles, then a
# rocprofv3 kernel trace of the group mode
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_esl
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_own.py -x -q 2>&1 | tail -25 > $OUT/test_own.log
cat $OUT/test_own.log
if [ "$1" != "quick" ]; then
timeout 1200 python -m pytest tests/test_gpu_cols.py tests/test_gpu_fused.py tests/test_gpu_a4_bruteforce.py tests/test_gpu_configs.py -x -q 2>&1 | tail -8 > $OUT/test_regr.log
cat $OUT/test_regr.log
fi
B=16 G=3 timeout 600 python tools/esl_groups.py 2>&1 | tail -5 | tee $OUT/groups_own.log
XM_COLS=2 B=16 G=3 ONLY=single timeout 600 python tools/esl_groups.py 2>&1 | tail -3 | tee $OUT/single_own.log
XM_COLS=0 B=16 G=3 timeout 600 python tools/esl_groups.py 2>&1 | tail -5 | tee $OUT/groups_direct.log
cd /tmp && export TMPDIR=/tmp
ONLY=groups B=16 G=3 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_own -o own -- python $GRAFT_REPO_ROOT/tools/esl_groups.py > $OUT/prof_own.log 2>&1
XM_COLS=0 ONLY=groups B=16 G=3 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_direct -o direct -- python $GRAFT_REPO_ROOT/tools/esl_groups.py > $OUT/prof_direct.log 2>&1
cd $GRAFT_REPO_ROOT
for d in prof_own prof_direct; do
  f=$(find $OUT/$d -name "*.db" | head -1)
  echo "== $d $f"; python tools/rocprof_summary.py "$f" | tee $OUT/$d.md | cut -c1-200
  rm -rf $OUT/$d
done
