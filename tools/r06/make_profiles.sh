#!/bin/bash
# Runs on the MI355X box (gpurun): kernel traces + PMC passes of every bench workload as of round 6, plus the plain bench lines.
# Each profiler run is bounded by its own timeout; each --pmc group is its own run with --kernel-trace only.
#   bash tools/r06/make_profiles.sh [tag]     -> gpurun_out/<tag>/ (dbs, removed at the end) and gpurun_out/<tag>_profiles/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r06p}
OUT=gpurun_out/$TAG
python3 -c "import shutil, sys; shutil.rmtree(sys.argv[1], ignore_errors=True)" "$OUT"; mkdir -p "$OUT"
export XM_BENCH_PREWARM_S=0.05
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc"
T="timeout 240"
run_set () {  # name, bench flags...
  local name=$1; shift
  $T rocprofv3 --kernel-trace --stats -d $OUT -o trace_$name -- python bench.py "$@" $Q > $OUT/trace_$name.log 2>&1 || echo "trace $name failed"
  local i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" \
             "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "GRBM_GUI_ACTIVE TA_TA_BUSY_sum"; do
    i=$((i+1))
    [ -n "$PMC_SETS" ] && [ $i -gt $PMC_SETS ] && break
    $T rocprofv3 --kernel-trace --pmc $set -d $OUT -o pmc_${name}_$i -- python bench.py "$@" --steps 10 --warmup 2 $Q > $OUT/pmc_${name}_$i.log 2>&1 || echo "pmc $name pass $i failed: $set"
  done
}
run_set groups --groups-in-flight 1
run_set esl --esl --groups-in-flight 1
PMC_SETS=3 run_set camg --groups-in-flight 1 --camera-perspective
# (a frame at a time from Python: every kernel of the chain alone on the GPU, as bench.py's own timing pass takes them)
PMC_SETS=2 run_set sharded --sharded --lanes 1 --comm torch
PMC_SETS=2 run_set shardedkeys --sharded --merge all_reduce --lanes 1 --comm torch
$T rocprofv3 --kernel-trace --stats -d $OUT -o trace_sharded2 -- python bench.py --sharded $Q > $OUT/trace_sharded2.log 2>&1  # the default: 2 lanes, the library's communicators
PMC_SETS=2 run_set single --batch 0 --slots 1
$T rocprofv3 --kernel-trace --stats -d $OUT -o trace_groups3 -- python bench.py $Q > $OUT/trace_groups3.log 2>&1
$T rocprofv3 --kernel-trace --stats -d $OUT -o trace_graph -- python bench.py --graph $Q > $OUT/trace_graph.log 2>&1
$T rocprofv3 --kernel-trace --stats -d $OUT -o trace_evt3 -- python tools/evt3_probe.py 2000000 20 > $OUT/trace_evt3.log 2>&1
$T rocprofv3 --kernel-trace --stats -d $OUT -o trace_ingest -- python bench.py --no-cpu-baseline --no-other-modes --no-other-configs --no-pmc > $OUT/trace_ingest.log 2>&1
$T rocprofv3 --kernel-trace --stats -d $OUT -o trace_eslstream -- python tools/r05/act_probe.py 1 3 > $OUT/trace_eslstream.log 2>&1        # activity filter ON (the default)
$T rocprofv3 --kernel-trace --stats -d $OUT -o trace_eslstream_off -- python tools/r05/act_probe.py 0 3 > $OUT/trace_eslstream_off.log 2>&1  # ... and off
$T rocprofv3 --kernel-trace --stats -d $OUT -o trace_b1 -- python bench.py --batch 1 --groups-in-flight 1 --steps 200 --warmup 20 $Q --no-other-configs > $OUT/trace_b1.log 2>&1  # a group of ONE frame through the column tiles
unset XM_BENCH_PREWARM_S
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
timeout 200 python bench.py --graph > $OUT/bench_graph60.json 2> $OUT/bench_graph60.err
timeout 200 python bench.py --sharded > $OUT/bench_sharded.json 2> $OUT/bench_sharded.err
timeout 200 python bench.py --sharded --lanes 4 --no-cpu-baseline > $OUT/bench_sharded_4_lanes.json 2> $OUT/bench_sharded4.err
timeout 300 python bench.py --esl > $OUT/bench_esl.json 2> $OUT/bench_esl.err
timeout 200 python bench.py --batch 0 --no-cpu-baseline --no-host-path --no-other-modes --no-pmc > $OUT/bench_one_frame_per_call.json 2> $OUT/bench_one.err
timeout 200 python bench.py --camera-perspective --no-cpu-baseline --no-other-modes --no-host-path --no-pmc > $OUT/bench_camera.json 2> $OUT/bench_camera.err
python tools/r05/collect_profiles.py $TAG 6 > $OUT/collect.log 2>&1; tail -3 $OUT/collect.log
python3 -c "import glob, os, sys; [os.remove(f) for p in ('*.db', '*.csv') for f in glob.glob(os.path.join(sys.argv[1], p))]" "$OUT"
ls $OUT | wc -l; du -sh $OUT gpurun_out/${TAG}_profiles
