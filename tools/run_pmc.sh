set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/pmc
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc/fetch gpurun_out/pmc/write
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc/fetch -o f -- python $R/tools/pmc_probe.py > $R/gpurun_out/pmc/fetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc/write -o w -- python $R/tools/pmc_probe.py > $R/gpurun_out/pmc/write.log 2>&1 )
