#!/bin/bash
# Owner tiles (ESL-like groups, one group at a time): what stalls k_scatter_own_batch -- the workgroup dispatcher's resource-allocation
# stalls (which resource is full when a block could not be placed), the waves' own wait / issue cycles, instruction fetch.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05/ownpmc; mkdir -p $OUT
Q="--esl --groups-in-flight 1 --steps 10 --warmup 2 --no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs"
i=0
for set in "SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_RES_STALL_CSN SPI_RA_SGPR_SIMD_FULL_CSN SPI_RA_BAR_CU_FULL_CSN SPI_RA_TGLIM_CU_FULL_CSN" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM GRBM_GUI_ACTIVE SPI_CSN_BUSY"; do
  i=$((i+1))
  XM_BENCH_PREWARM_S=0.05 timeout 200 rocprofv3 --kernel-trace --pmc $set -d $OUT -o p$i -- python bench.py $Q $XM_SWEEP_OPTS > /dev/null 2>&1
done
python - <<PY
import sqlite3, glob
rows = {}
for db in glob.glob("$OUT/**/p*_results.db", recursive=True):
    try:
        for k, cn, v in sqlite3.connect(db).execute("select kernel_name, counter_name, avg(value) from counters_collection group by 1, 2"):
            if "k_scatter_own_batch" in k or "k_frame_proj_pipe" in k or "k_cols_bounds_batch" in k:
                rows.setdefault(k.split("(")[0][:40], {})[cn] = v
    except Exception as e:
        print(db, e)
for k, d in rows.items():
    print(k)
    for cn in sorted(d): print("   %-32s %16.0f" % (cn, d[cn]))
PY
rm -rf $OUT
