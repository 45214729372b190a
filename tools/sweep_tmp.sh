cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/flags
for f in 1 0; do
XM_K2_FLAGS=$f rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/flags -o f${f}_fetch -- python bench.py --slots 1 --steps 60 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
XM_K2_FLAGS=$f rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum -d gpurun_out/flags -o f${f}_tcc -- python bench.py --slots 1 --steps 60 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
done
