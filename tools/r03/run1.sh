#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_bench; mkdir -p $OUT
XM_BENCH_PREWARM_S=0.2 timeout 300 python bench.py --esl --no-cpu-baseline --no-host-path > $OUT/esl.json 2> $OUT/esl.err; tail -c 300 $OUT/esl.err
python -c "
import json; d=json.loads(open('$OUT/esl.json').read().strip().splitlines()[-1]); r=d['roofline']
print('esl value', d['value'], 'us/frame', d['config']['us_per_frame'], r['kernel'], 'frac', r['frac'], {k:(v['us_per_frame'], v['frac_algorithmic']) for k,v in r['kernels'].items()}, d['other_modes'])"
ONLY=groups B=32 G=3 bash tools/r03/pmc.sh r03_pmc_esl 32 python tools/esl_groups.py
