#!/bin/bash
# Runs on the MI355X box (gpurun): kernel traces + PMC passes of the bench workloads as of the end of round 2 (a step = one
# group of 16 frames through xm_process_batch; K1 = column tiles).  Every profiler run is bounded by its own timeout; every
# --pmc group is its own run with --kernel-trace only (FETCH_SIZE and WRITE_SIZE separately).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r02c}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export XM_BENCH_PREWARM_S=0.05
Q="--no-cpu-baseline --no-other-modes --no-host-path"
T="timeout 240"
$T rocprofv3 --kernel-trace --stats -d $OUT -o trace_groups -- python bench.py $Q > $OUT/trace_groups.log 2>&1
$T rocprofv3 --kernel-trace --stats -d $OUT -o trace_serial -- python bench.py --groups-in-flight 1 $Q > $OUT/trace_serial.log 2>&1
$T rocprofv3 --kernel-trace --stats -d $OUT -o trace_single -- python bench.py --batch 0 --slots 1 --steps 200 --warmup 20 $Q > $OUT/trace_single.log 2>&1
$T rocprofv3 --kernel-trace --stats -d $OUT -o trace_cam -- python bench.py --groups-in-flight 1 --camera-perspective $Q > $OUT/trace_cam.log 2>&1
$T rocprofv3 --kernel-trace --stats -d $OUT -o trace_batch60 -- python tools/batch_probe.py 60 4 2 > $OUT/trace_batch60.log 2>&1
PM="python bench.py --groups-in-flight 1 --steps 10 --warmup 2 $Q"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE TA_TA_BUSY_sum"; do
  i=$((i+1))
  $T rocprofv3 --kernel-trace --pmc $set -d $OUT -o pmc_groups_$i -- $PM > $OUT/pmc_groups_$i.log 2>&1 || echo "pmc_groups pass $i failed: $set"
done
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  $T rocprofv3 --kernel-trace --pmc $set -d $OUT -o pmc_camg_$i -- $PM --camera-perspective > $OUT/pmc_camg_$i.log 2>&1 || echo "pmc_camg pass $i failed"
done
unset XM_BENCH_PREWARM_S
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 200 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
timeout 200 python bench.py --steps 2000 $Q > $OUT/bench_steps2000.json 2> $OUT/bench_steps2000.err
timeout 200 python bench.py --batch 0 --steps 2000 --warmup 200 $Q > $OUT/bench_one_frame_per_call.json 2> $OUT/bench_one_frame_per_call.err
timeout 200 python bench.py --graph > $OUT/bench_graph60.json 2> $OUT/bench_graph60.err
timeout 200 python bench.py --sharded > $OUT/bench_sharded.json 2> $OUT/bench_sharded.err
timeout 200 python bench.py --esl > $OUT/bench_esl.json 2> $OUT/bench_esl.err
timeout 200 python bench.py --camera-perspective --no-cpu-baseline --no-other-modes > $OUT/bench_camera.json 2> $OUT/bench_camera.err
python tools/collect_profiles_r02b.py $TAG --to-scratch > $OUT/collect.log 2>&1
rm -f $OUT/*.db $OUT/*.csv
ls $OUT | wc -l; du -sh $OUT gpurun_out/${TAG}_profiles
