set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3n; rm -rf $O; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
CTCN_OPT_FWD_RSV_LDS=1 timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider -k "rnn or model_three or fused_dropout or overlap_equals or shipped or elementwise" > $O/pytest_rsv.log 2>&1; echo "pytest (rsv) rc=$?"; tail -4 $O/pytest_rsv.log
for cfg in "FWD_RSV_LDS=0" "FWD_RSV_LDS=1" "FWD_RSV_LDS=0" "FWD_RSV_LDS=1" "FWD_RSV_LDS=1 TAG_POLL_DELAY=6" "FWD_RSV_LDS=1 TAG_POLL_DELAY=10"; do
  envs=""; for kv in $cfg; do envs="$envs CTCN_OPT_${kv%%=*}=${kv##*=}"; done
  for wl in cfg2 ref_yaml cfg4; do
    r=$(env $envs timeout 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-decode 2>$O/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  fwd %.3f bwd %.3f us' % (d['ms_per_step'], d['recurrence']['fwd_us_per_timestep'], d['recurrence']['bwd_us_per_timestep']))" 2>&1)
    echo "[$cfg] $wl: $r"
  done
done
