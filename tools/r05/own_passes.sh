#!/bin/bash
# Owner tiles: events per thread (XM_OWN_EPT = 4 / 8) x row passes of a tile's LDS slots (XM_OWN_ROW_PASSES = 1..4): per-kernel
# times + the pipelined step (ESL-like groups).   bash tools/r05/own_passes.sh "4 8" "1 2 3"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for E in ${1:-4 8}; do for P in ${2:-1 2 3 4}; do
  XM_SWEEP_OPTS="--lib-option XM_OWN_ROW_PASSES=$P --lib-option XM_OWN_EPT=$E" bash tools/r05/own_sweep.sh 8 2>&1 | sed "s/^/EPT $E P $P /" | cut -c1-140
done; done
