set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/pmc
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc/gfetch gpurun_out/pmc/gwrite
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc/gfetch -o f -- python $R/tools/pmc_gemm.py > $R/gpurun_out/pmc/gfetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc/gwrite -o w -- python $R/tools/pmc_gemm.py > $R/gpurun_out/pmc/gwrite.log 2>&1 )
