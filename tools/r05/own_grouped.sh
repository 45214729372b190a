#!/bin/bash
# Owner tiles: ownership per 8-row group (16-byte flush, halo grows by the X-map's slant over 8 rows) against per row (XM_OWN_GROUPED=0),
# tile widths 8 / 16, row passes: per-kernel times (one ESL-like group at a time) + the pipelined step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for G in 0 1; do for W in 8 16; do for P in ${@:-2 3 4}; do
  [ $G = 0 ] && [ $W = 16 ] && continue
  XM_SWEEP_OPTS="--lib-option XM_OWN_ROW_PASSES=$P --lib-option XM_OWN_GROUPED=$G" bash tools/r05/own_sweep.sh $W 2>&1 | sed "s/^/grouped $G P $P /" | cut -c1-260
done; done; done
