cd /tmp; export TMPDIR=/tmp
for cfg in "X=0" "CTCN_OPT_GEMM_BF16_SINGLE=1" "CTCN_FWD_OVERLAP=0"; do
  rm -rf /tmp/pp; env $cfg timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-decode --no-cpu-baseline --no-others --no-pmc > /tmp/b.json 2>/tmp/p.log
  db=$(find /tmp/pp -name "*.db" | head -1)
  echo "== $cfg: $(python -c "import json; print(json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])['ms_per_step'])") ms/step"
  PROF_STEP=8 python $GRAFT_REPO_ROOT/tools/prof_timeline.py $db -1 | grep "rnn_fwd_tagged\|rnn_bwd_scatter\|^#"
done
