cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for W in 4 8; do for E in 4 8; do for P in 1 2 3; do
XM_SWEEP_OPTS="--lib-option XM_OWN_ROW_PASSES=$P --lib-option XM_OWN_EPT=$E" bash tools/r05/own_sweep.sh $W 2>&1 | grep "gif 1" | sed "s/^/EPT $E P $P /" | cut -c1-150
done; done; done
