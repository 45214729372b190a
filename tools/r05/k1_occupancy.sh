#!/bin/bash
# Item 6 of round 4's review: both occupancy walls of the headline K1 (column tiles, C-1M groups) lowered TOGETHER.
#   LDS wall:  the LUT band w_x = 16 camera columns (48.6 KB per block: 3 blocks per CU) -> w_x = 11 (39.0 KB: 4 blocks);
#              events outside the narrower band take the global LUT (the kernel's existing slow path)
#   VGPR wall: __launch_bounds__(512, 6) (80 VGPRs: 6 waves per SIMD = 3 blocks of 8 waves) -> (512, 8) (64 VGPRs: 4 blocks)
# Per variant: bench.py's own per-kernel timing (HIP events per dispatch, one group at a time) + the pipelined step, then
# SQ counters of the K1 dispatches.     bash tools/r05/k1_occupancy.sh     (variants/libxmaps_hip_w8.so built beforehand)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05/k1occ; mkdir -p $OUT
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs"
cp x_maps_amd/libxmaps_hip.so /tmp/lib_w6.so
for lib in w6 w8; do
  [ $lib = w8 ] && cp variants/libxmaps_hip_w8.so x_maps_amd/libxmaps_hip.so || cp /tmp/lib_w6.so x_maps_amd/libxmaps_hip.so
  touch x_maps_amd/libxmaps_hip.so
  for wx in 16 11; do
    python bench.py $Q --lib-option XM_K1_WX=$wx 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('waves/EU ${lib#w}  w_x $wx :', d['value'], 'Mev/s', d['ms_per_step'], 'ms/step  K0b/K1/K2 us', r['avg_launch_us'], 'parity', d['parity'].get('depth_bit_exact'), d['parity'].get('bgr_equal'))"
    XM_BENCH_PREWARM_S=0.05 timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT -o pmc_${lib}_$wx -- python bench.py $Q --groups-in-flight 1 --steps 10 --warmup 2 --lib-option XM_K1_WX=$wx > /dev/null 2>&1
    python - <<PY
import sqlite3, glob
for db in glob.glob("$OUT/**/pmc_${lib}_${wx}_results.db", recursive=True):
    rows = {}
    for k, cn, v in sqlite3.connect(db).execute("select kernel_name, counter_name, avg(value) from counters_collection group by 1, 2"):
        if "k_scatter_cols_batch" in k: rows[cn] = v
    if rows:
        w = rows.get("SQ_WAVES", 1)
        print("   K1 counters per dispatch: waves %.0f, wave-cycles/busy-cycles (mean waves in flight per SQ) %.2f, WAIT_ANY %.0f %% of wave cycles, WAIT_INST_ANY %.0f %%, ACTIVE_INST_ANY %.0f %%, GRBM_GUI_ACTIVE %.0f" % (
            w, rows.get("SQ_WAVE_CYCLES", 0) / max(rows.get("SQ_BUSY_CYCLES", 1), 1), 100 * rows.get("SQ_WAIT_ANY", 0) / max(rows.get("SQ_WAVE_CYCLES", 1), 1),
            100 * rows.get("SQ_WAIT_INST_ANY", 0) / max(rows.get("SQ_WAVE_CYCLES", 1), 1), 100 * rows.get("SQ_ACTIVE_INST_ANY", 0) / max(rows.get("SQ_WAVE_CYCLES", 1), 1), rows.get("GRBM_GUI_ACTIVE", 0)))
PY
  done
done
cp /tmp/lib_w6.so x_maps_amd/libxmaps_hip.so
rm -rf $OUT/*/ 2>/dev/null; find $OUT -name "*.db" -delete
