set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3g; rm -rf $O; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
for cfg in "TAG_POLL_DELAY=4" "TAG_POLL_DELAY=8" "TAG_POLL_DELAY=12" "TAG_POLL_DELAY=16" "TAG_POLL_DELAY=20"; do
  envs=""; for kv in $cfg; do envs="$envs CTCN_OPT_${kv%%=*}=${kv##*=}"; done
  for wl in cfg4 ref_yaml cfg1; do
    r=$(env $envs timeout 300 python bench.py --workload $wl --steps 12 --warmup 3 --no-cpu-baseline --no-decode 2>$O/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  fwd %.3f bwd %.3f us  %s %s' % (d['ms_per_step'], d['recurrence']['fwd_us_per_timestep'], d['recurrence']['bwd_us_per_timestep'], d['recurrence']['fwd_kernel'], d['recurrence']['bwd_kernel']))" 2>&1)
    echo "[$cfg] $wl: $r"
  done
done
