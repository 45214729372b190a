#!/bin/bash
# Runs on the MI355X box (gpurun): kernel trace + PMC passes of the default bench workload (projector + camera view).
# Each --pmc group is its own rocprofv3 run with --kernel-trace only (the GPU pool refuses other combinations).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
OUT=gpurun_out/$TAG; mkdir -p $OUT
BASE="python bench.py --slots 1 --steps 200 --warmup 20 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT -o trace_proj -- $BASE > $OUT/trace_proj.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT -o trace_cam -- $BASE --camera-perspective > $OUT/trace_cam.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT -o trace_pipe8 -- python bench.py --steps 400 --no-cpu-baseline > $OUT/trace_pipe8.log 2>&1
PM="python bench.py --slots 1 --steps 60 --warmup 10 --no-cpu-baseline"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT -o pmc_proj_$i -- $PM > $OUT/pmc_proj_$i.log 2>&1 || echo "pmc pass $i failed"
done
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT -o pmc_cam_$i -- $PM --camera-perspective > $OUT/pmc_cam_$i.log 2>&1 || echo "pmc pass $i failed"
done
python bench.py --host-path > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --camera-perspective --no-cpu-baseline > $OUT/bench_camera.json 2>/dev/null
python bench.py --graph --slots 8 --no-cpu-baseline > $OUT/bench_graph60.json 2>/dev/null
python tools/scale_probe.py > $OUT/scale_probe.txt 2>/dev/null
ls $OUT | head -50
