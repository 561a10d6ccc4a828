set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/pmc gpurun_out/prof
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc/fetch -o f -- python $R/tools/pmc_probe.py > $R/gpurun_out/pmc/fetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc/write -o w -- python $R/tools/pmc_probe.py > $R/gpurun_out/pmc/write.log 2>&1 )
for wl in cfg1 cfg3 cfg4; do timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_$wl.json; done
timeout 300 python bench.py --mode decode 2>/dev/null | tail -1 > gpurun_out/bench_decode.json
timeout 300 python bench.py --precision 0 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_cfg2_f32.json
