#!/bin/bash
# round 6: kernel durations under overlap (four groups in flight), compact frames against the rectified frame
# (variants/libxmaps_cmp*.so = builds of tools/r06/patches/compact_frames.patch applied to the commit; cmp4: with -DXM_K2C_LATE=1)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r06/cmp2_trace; rm -rf $OUT; mkdir -p $OUT
export XM_LIB=${1:-variants/libxmaps_cmp2.so}
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --steps 40 --warmup 10 --single-block"
for C in 0 2 0 2; do
  timeout 240 rocprofv3 --kernel-trace -d $OUT -o c$C --output-format csv -- python bench.py $Q --lib-option XM_COMPACT=$C > $OUT/c$C.log 2>&1
  grep '^{"metric' $OUT/c$C.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('compact $C: step', d['ms_per_step'], d['roofline']['avg_launch_us'])"
  python3 - <<PY
import csv,glob,collections
for f in glob.glob('$OUT/**/c${C}_kernel_trace.csv', recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if any(k in r['Kernel_Name'] for k in ('k_scatter_cols_batch','k_frame_proj_pipe','k_cols_bounds_batch'))]
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    rows=rows[len(rows)//2:]   # the steady state
    t0=int(rows[0]['Start_Timestamp']); t1=max(int(r['End_Timestamp']) for r in rows)
    d=collections.defaultdict(list)
    for r in rows: d[r['Kernel_Name'].split('(')[0][-40:]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    n=len(rows)/3
    print('   span per group us', round((t1-t0)/1e3/n,2), {k:(len(v), round(sum(v)/len(v),1)) for k,v in d.items()})
    # busy union
    ev=sorted([(int(r['Start_Timestamp']),1) for r in rows]+[(int(r['End_Timestamp']),-1) for r in rows])
    busy=0;depth=0;last=None;w=collections.Counter()
    for t,s in ev:
        if last is not None: w[depth]+=t-last
        depth+=s;last=t
    tot=sum(w.values()); print('   concurrency share', {k:round(v/tot,3) for k,v in sorted(w.items())})
PY
done
python3 -c "import glob, os; [os.remove(f) for f in glob.glob('$OUT/**/*', recursive=True) if f.endswith(('.db', '.csv'))]"
