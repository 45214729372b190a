#!/bin/bash
# round 6: the pipelined K2 without per-lane regions (variants/libxmaps_k2bf.so) against the round's K2 (variants/libxmaps_base.so): tests, timing, counters
# (the variant libraries are builds of tools/r06/patches/k2_branch_free.patch applied to xmaps_k2pipe.hpp: its first half = libxmaps_k2m.so, all of it = libxmaps_k2bf.so;
#  variants/libxmaps_base.so = a build of the commit itself: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared x_maps_amd/csrc/xmaps_hip.hip -o <lib>)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
V=${1:-variants/libxmaps_k2bf.so}
( XM_LIB=$V timeout 900 python -m pytest tests/test_gpu_k2pipe.py tests/test_gpu_fused.py tests/test_gpu_configs.py tests/test_gpu_own.py tests/test_gpu_cols.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2
XM_LIB=$V timeout 600 python tools/fuzz_soak.py 9000 150 2>&1 | tail -1
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --steps 40 --warmup 10"
for rep in 1 2 3; do for L in variants/libxmaps_base.so $V; do for M in "" "--groups-in-flight 1" "--esl" "--esl --groups-in-flight 1"; do
  XM_LIB=$L python bench.py $Q $M 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('lib [$L] [$M] rep $rep: step ms', d.get('ms_per_step'), 'value', d.get('value'), d.get('roofline',{}).get('avg_launch_us'), 'exact', (d.get('parity') or {}).get('depth_bit_exact', (d.get('parity') or {}).get('group_last_frame_depth_bit_exact')), d.get('error'))"
done; done; done
OUT=gpurun_out/r06/k2bf_pmc; rm -rf $OUT; mkdir -p $OUT
for M in esl c1m; do
[ $M = esl ] && QQ="--esl" || QQ=""
for L in variants/libxmaps_base.so $V; do
  T=$(basename $L .so)_$M
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_BUSY_CYCLES" \
             "SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES" \
             "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    XM_LIB=$L timeout 240 rocprofv3 --kernel-trace --pmc $set -d $OUT -o ${T}_$i --output-format csv -- python bench.py $QQ --groups-in-flight 1 --steps 10 --warmup 2 --no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs > $OUT/${T}_$i.log 2>&1 || echo "pass $T $i failed"
  done
done; done
python3 - <<'PY'
import csv, glob, collections, os
out = "gpurun_out/r06/k2bf_pmc"
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    tag = "_".join(os.path.basename(f).split("_")[:-3])
    for r in csv.DictReader(open(f)):
        if "k_frame_proj_pipe" in r["Kernel_Name"]:
            res[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
for tag in sorted(res):
    print(tag, {k: round(sum(v) / len(v)) for k, v in sorted(res[tag].items())})
PY
python3 -c "import glob, os; [os.remove(f) for f in glob.glob('gpurun_out/r06/k2bf_pmc/**/*', recursive=True) if f.endswith(('.db', '.csv'))]"
) 2>&1 | tee gpurun_out/r06/k2_bf.txt
