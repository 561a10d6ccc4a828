set -u
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/step0; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
( cd /tmp && CTCN_FWD_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o cfg2 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/bench.json 2> $O/stats.log )
db=$(find $O/stats -name "*.db" | head -1); [ -n "$db" ] && python tools/prof_timeline.py $db -1 > $O/timeline.txt 2>&1
rm -rf $O/stats
grep "rnn_fwd_tagged\|^# " $O/timeline.txt | cut -c1-120
