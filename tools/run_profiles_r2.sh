# Round-2 evidence in one box session (outputs under gpurun_out/r02/, copied to profiles/ by hand afterwards):
#   1. rocprofv3 --kernel-trace --stats of the default driver command (python bench.py) -> kernel stats + step timeline
#   2. FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, kernel-trace only) over tools/pmc_probe.py -> HBM bytes per launch
set -u
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $O/stats -o cfg2 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.log )
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r02_cfg2_train_decode_kernel_stats.csv
db=$(find $O/stats -name "*.db" | head -1); [ -n "$db" ] && python tools/prof_timeline.py $db -1 > $O/r02_cfg2_step_timeline.txt 2>&1
[ -n "$db" ] && python tools/prof_stats.py $db > $O/r02_cfg2_train_decode_kernel_stats.txt 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f -- python $R/tools/pmc_probe.py > $O/fetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w -- python $R/tools/pmc_probe.py > $O/write.log 2>&1 )
fd=$(find $O/fetch -name "*.db" | head -1); wd=$(find $O/write -name "*.db" | head -1)
[ -n "$fd" ] && [ -n "$wd" ] && python tools/pmc_to_json.py $fd $wd $O/r02_pmc_hbm_traffic.json > $O/pmc.log 2>&1
( timeout 120 ./tools/mb_step.bin > $O/r02_mb_step.txt 2>&1 )
rm -rf $O/stats/*/*.db $O/fetch $O/write 2>/dev/null
ls -la $O
