# rocprofv3 kernel trace of the decode leg (bench.py --mode decode): do the searches of two streams overlap? -> gpurun_out/dec/trace.txt
set -u
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dec; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/stats -o dec -- python $R/bench.py --mode decode --steps 3 > $O/bench.json 2> $O/stats.log )
db=$(find $O/stats -name "*.db" | head -1)
python - "$db" > $O/trace.txt <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, stream_id, start, end from kernels where name like '%beam%' or name like '%fillBuffer%' order by start"))
t0 = rows[0][2]
for n, s, a, b in rows:
    print("%10.1f %10.1f  dur %8.1f  stream %s  %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, s, n.split("(")[0][-40:]))
PY
rm -rf $O/stats
grep -c . $O/trace.txt; grep "beam_fast" $O/trace.txt | tail -24
