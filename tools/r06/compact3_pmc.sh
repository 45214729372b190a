#!/bin/bash
# round 6: HBM bytes per group under overlap (four groups in flight), compact frames (2) against the rectified frame (0)
# (variants/libxmaps_cmp*.so = builds of tools/r06/patches/compact_frames.patch applied to the commit; cmp4: with -DXM_K2C_LATE=1)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r06/cmp3_pmc; rm -rf $OUT; mkdir -p $OUT
export XM_LIB=${1:-variants/libxmaps_cmp3.so}
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --steps 20 --warmup 5"
for C in 0 2; do for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $set | cut -c1-5)
  timeout 240 rocprofv3 --kernel-trace --pmc $set -d $OUT -o c${C}_$n --output-format csv -- python bench.py $Q --lib-option XM_COMPACT=$C > $OUT/c${C}_$n.log 2>&1
done; done
python3 - <<'PY'
import csv, glob, collections, os
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r06/cmp3_pmc/**/*counter_collection.csv", recursive=True):
    tag = os.path.basename(f).split("_")[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "K1" if "k_scatter_cols_batch" in k else "K2" if "k_frame_proj_pipe" in k else "K0b" if "k_cols_bounds" in k else None
        if name: res[tag][(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
for tag in sorted(res):
    print(tag, {f"{k[0]} {k[1]}": round(sum(v) / len(v)) for k, v in sorted(res[tag].items())})
PY
python3 -c "import glob, os; [os.remove(f) for f in glob.glob('gpurun_out/r06/cmp3_pmc/**/*', recursive=True) if f.endswith(('.db', '.csv'))]"
