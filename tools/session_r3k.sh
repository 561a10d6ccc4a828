set -u
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3k; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o ry -- python $R/bench.py --workload ref_yaml --steps 20 --warmup 3 --no-cpu-baseline --no-decode > $O/bench_ref_yaml_rocprof.json 2> $O/stats.log )
db=$(find $O/stats -name "*.db" | head -1)
[ -n "$db" ] && python tools/prof_stats.py $db > $O/r03_ref_yaml_kernel_stats.txt 2>&1
[ -n "$db" ] && python tools/prof_timeline.py $db -1 > $O/r03_ref_yaml_step_timeline.txt 2>&1
rm -rf $O/stats/*/*.db
head -45 $O/r03_ref_yaml_kernel_stats.txt | cut -c1-150
