cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for W in 12 16 20 24; do for P in 2 3; do
  XM_SWEEP_OPTS="--lib-option XM_OWN_ROW_PASSES=$P" bash tools/r05/own_sweep.sh $W 2>&1 | sed "s/^/grouped P $P /" | cut -c1-230
done; done
