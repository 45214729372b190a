#!/bin/bash
# Owner tiles: what the phases cost -- K1 of one ESL-like group at a time with parts switched off (-DXM_ABLATE build in variants/;
# the results are wrong on purpose: --no-parity).  bit 4 (16): no flush stores, bit 5 (32): no LUT / X-map gathers, bit 6 (64): no
# event loads, bit 7 (128): no per-event work at all (head, clear, flush only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
Q="--esl --no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --no-parity --groups-in-flight 1 --steps 10 --warmup 2"
for E in ${1:-4}; do for P in ${2:-1}; do for A in ${3:-0 16 32 64 96 112 128 144}; do
  XM_LIB=variants/libxmaps_abl.so python bench.py $Q --lib-option XM_OWN_EPT=$E --lib-option XM_OWN_ROW_PASSES=$P --lib-option XM_ABLATE=$A 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('EPT $E passes $P ablate $A : K0b/K1/K2 us', d.get('roofline',{}).get('avg_launch_us'), d.get('error'))"
done; done; done
