# kernel stats of tools/conv_bench.py (cfg3 front-end convolutions, MFMA vs direct)
set -u
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/conv; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o cb -- python $R/tools/conv_bench.py > $O/bench.log 2> $O/stats.log )
db=$(find $O/stats -name "*.db" | head -1); [ -n "$db" ] && python tools/prof_stats.py $db > $O/conv_kernel_stats.txt 2>&1
rm -rf $O/stats
cat $O/bench.log; head -30 $O/conv_kernel_stats.txt
