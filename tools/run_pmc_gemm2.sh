set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/pmc
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc/g2_*
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc/g2_$i -o p -- python $R/tools/pmc_gemm.py > $R/gpurun_out/pmc/g2_$i.log 2>&1 )
  f=$(find gpurun_out/pmc/g2_$i -name "*.db" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name order by 1"))
for name, cname, n, avg, dur in rows:
    if "gemm_planes" in name or "split_rows" in name:
        print("%-40s %-32s n=%3d avg=%16.1f dur=%.1f us" % (name.replace("(anonymous namespace)::", "").split("(")[0][-40:], cname, n, avg, dur / 1e3))
PY
done
