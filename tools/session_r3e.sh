set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3e; rm -rf $O; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
for cfg in "BWD_ITEM_GATHER=0" "BWD_ITEM_GATHER=2" "BWD_ITEM_GATHER=2 BWD_POLL_DELAY=24"; do
  envs=""; for kv in $cfg; do envs="$envs CTCN_OPT_${kv%%=*}=${kv##*=}"; done
  for wl in cfg4 ref_yaml cfg2; do
    r=$(env $envs timeout 300 python bench.py --workload $wl --steps 15 --warmup 3 --no-cpu-baseline --no-decode 2>$O/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  fwd %.3f bwd %.3f us  %s %s' % (d['ms_per_step'], d['recurrence']['fwd_us_per_timestep'], d['recurrence']['bwd_us_per_timestep'], d['recurrence']['fwd_kernel'], d['recurrence']['bwd_kernel']))" 2>&1)
    echo "[$cfg] $wl: $r"
  done
done
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider -k "rnn or model_three or fused_dropout or side_stream or elementwise or shipped or large_shape or foreign" > $O/pytest_rnn.log 2>&1; echo "pytest rnn (default policy) rc=$?"
tail -3 $O/pytest_rnn.log
CTCN_OPT_BWD_ITEM_GATHER=2 timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider -k "rnn or model_three or fused_dropout or side_stream or elementwise or shipped or large_shape" > $O/pytest_rnn_ig2.log 2>&1; echo "pytest rnn (item gather everywhere) rc=$?"
tail -3 $O/pytest_rnn_ig2.log
