#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_bench; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_own.py tests/test_gpu_api.py -x -q 2>&1 | tail -5
XM_BENCH_PREWARM_S=0.2 timeout 300 python bench.py --esl --no-cpu-baseline --no-host-path > $OUT/esl.json 2> $OUT/esl.err; tail -c 300 $OUT/esl.err
python -c "
import json; d=json.loads(open('$OUT/esl.json').read().strip().splitlines()[-1]); r=d['roofline']
print('esl value', d['value'], 'us/frame', d['config']['us_per_frame'], r['kernel'], 'frac', r['frac'], {k:(v['us_per_frame'], v['frac_algorithmic']) for k,v in r['kernels'].items()}, d['other_modes'], d['config']['k1_geometry'])"
XM_COLS=2 XM_BENCH_PREWARM_S=0.2 timeout 300 python bench.py --batch 0 --no-cpu-baseline --no-host-path --no-other-modes > $OUT/single_cols.json 2> $OUT/single_cols.err
python -c "
import json; d=json.loads(open('$OUT/single_cols.json').read().strip().splitlines()[-1]); r=d['roofline']
print('C-1M one frame per call, XM_COLS=2: value', d['value'], {k:(v['us_per_frame']) for k,v in r['kernels'].items()}, d['config']['k1_paths_frames'])"
