cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/bp32; mkdir -p $OUT
timeout 120 rocprofv3 --kernel-trace --stats -d $OUT -o t2 -- python tools/batch_probe.py 60 4 2 > $OUT/t2.log 2>&1
timeout 120 rocprofv3 --kernel-trace --stats -d $OUT -o t1 -- python tools/batch_probe.py 60 4 1 > $OUT/t1.log 2>&1
for f in $OUT/*.db; do python tools/rocprof_summary.py $f | head -8; done
rm -f $OUT/*.db
