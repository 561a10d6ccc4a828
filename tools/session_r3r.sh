set -u
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3r; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for wl in ref_yaml cfg4; do
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats_$wl -o p -- python $R/bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/bench_${wl}_rocprof.json 2> $O/stats_$wl.log )
  db=$(find $O/stats_$wl -name "*.db" | head -1)
  [ -n "$db" ] && python tools/prof_stats.py $db > $O/r03_${wl}_kernel_stats.txt 2>&1
  [ -n "$db" ] && python tools/prof_timeline.py $db -1 > $O/r03_${wl}_step_timeline.txt 2>&1
  rm -rf $O/stats_$wl
done
timeout 400 python tools/squat_stress.py --workload cfg4 --steps 60 --out $O/r03_squat_stress_cfg4.json > $O/squat_cfg4.log 2>&1; echo "squat cfg4 rc=$?"
head -12 $O/r03_cfg4_kernel_stats.txt | cut -c1-140; cat $O/r03_squat_stress_cfg4.json
