#!/bin/bash
# round 6: dynamic instruction counts of the pipelined K2 on the ESL-like rig, product library against variants/libxmaps_k2m.so
# (the variant libraries are builds of tools/r06/patches/k2_branch_free.patch applied to xmaps_k2pipe.hpp: its first half = libxmaps_k2m.so, all of it = libxmaps_k2bf.so;
#  variants/libxmaps_base.so = a build of the commit itself: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared x_maps_amd/csrc/xmaps_hip.hip -o <lib>)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r06/k2m_pmc; rm -rf $OUT; mkdir -p $OUT
Q="--esl --groups-in-flight 1 --steps 10 --warmup 2 --no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs"
for L in base k2m; do
  [ $L = k2m ] && export XM_LIB=variants/libxmaps_k2m.so || unset XM_LIB
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
             "SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH" \
             "GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_I8"; do
    i=$((i+1))
    timeout 240 rocprofv3 --kernel-trace --pmc $set -d $OUT -o ${L}_$i --output-format csv -- python bench.py $Q > $OUT/${L}_$i.log 2>&1 || echo "pass $L $i failed"
  done
done
python3 - <<'PY'
import csv, glob, collections, os
out = "gpurun_out/r06/k2m_pmc"
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    tag = os.path.basename(f).split("_")[0]
    for r in csv.DictReader(open(f)):
        if "k_frame_proj_pipe" in r["Kernel_Name"]:
            res[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
for tag in sorted(res):
    print(tag, {k: round(sum(v) / len(v)) for k, v in sorted(res[tag].items())})
PY
python3 -c "import glob, os; [os.remove(f) for f in glob.glob('gpurun_out/r06/k2m_pmc/**/*', recursive=True) if f.endswith(('.db', '.csv'))]"
