#!/bin/bash
# ESL-like groups of 32 (owner tiles): the bench leg + a kernel trace of one group at a time.   bash tools/r05/esl_groups_probe.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-esl}
mkdir -p gpurun_out/r05
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc"
XM_BENCH_PREWARM_S=0.05 timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r05/prof_$TAG -o t -- python bench.py --esl --groups-in-flight 1 $Q > /dev/null 2>&1
python tools/rocprof_summary.py gpurun_out/r05/prof_$TAG/t_results.db | grep -E "kernel|own|bounds|frame_proj" | head -8
python bench.py --esl $Q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('esl groups:', d['value'], 'Mev/s', d['ms_per_step'], 'ms/step', d['roofline']['avg_launch_us'], d['parity'].get('depth_bit_exact'))"
