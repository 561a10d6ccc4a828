# round-3 session B: rnn_bwd_scatter2 (item-wave gather) -- correctness on the recurrent tests, then A/B against rnn_bwd_scatter
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3b; rm -rf $O; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider -k "rnn or model_three or fused_dropout or side_stream or elementwise or shipped or large_shape" > $O/pytest_rnn.log 2>&1; echo "pytest rnn rc=$?"
tail -15 $O/pytest_rnn.log
for cfg in "BWD_ITEM_GATHER=0" "BWD_ITEM_GATHER=1" "BWD_ITEM_GATHER=1 BWD_POLL_DELAY=4" "BWD_ITEM_GATHER=1 BWD_POLL_DELAY=8" "BWD_ITEM_GATHER=1 BWD_POLL_DELAY=16"; do
  envs=""; for kv in $cfg; do envs="$envs CTCN_OPT_${kv%%=*}=${kv##*=}"; done
  for wl in cfg2 ref_yaml cfg1; do
    r=$(env $envs timeout 200 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-decode 2>$O/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  fwd %.3f bwd %.3f us  %s' % (d['ms_per_step'], d['recurrence']['fwd_us_per_timestep'], d['recurrence']['bwd_us_per_timestep'], d['recurrence']['bwd_kernel']))" 2>&1)
    echo "[$cfg] $wl: $r"
  done
done
