#!/bin/bash
# round 6: what the boundary pass K0b costs in the pipelined C-1M step: a build that stops launching it after eight un-profiled groups.
# variants/libxmaps_skipk0b.so = the library with these lines in front of K0b's launch in host/xm_batch.hpp (column tiles' branch):
#   static const int skip_after = dbg_opt("XM_SKIP_K0B") ? atoi(dbg_opt("XM_SKIP_K0B")) : 0;
#   static std::atomic<int> groups_seen{0};
#   if (!(skip_after > 0 && !prof && groups_seen.fetch_add(1) >= skip_after))
# (legitimate only because bench.py feeds every slot the same frame again: the slots' boundaries are still valid; not in the tree)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r06
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --no-parity --steps 40 --warmup 10"
for rep in 1 2 3; do for OPT in "" "--lib-option XM_SKIP_K0B=8"; do
  XM_LIB=variants/libxmaps_skipk0b.so python bench.py $Q $OPT 2>>gpurun_out/r06/skip_err.txt | tail -1 | tee -a gpurun_out/r06/skip_raw.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[$OPT] rep $rep: step ms', d.get('ms_per_step'), 'value', d.get('value'), d.get('roofline',{}).get('avg_launch_us'), d.get('error'))"
done; done 2>&1 | tee gpurun_out/r06/skip_k0b.txt
