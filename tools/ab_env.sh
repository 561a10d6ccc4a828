# A/B of environment settings on bench workloads inside ONE box session: tools/ab_env.sh "<workload>" "VAR=V VAR2=V" "VAR=V" ...
cd "$GRAFT_REPO_ROOT"
wl="$1"; shift
for cfg in "$@"; do
  for rep in 1 2; do
    r=$(env $cfg python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-decode 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  fwd %.3f bwd %.3f us' % (d['ms_per_step'], d['recurrence']['fwd_us_per_timestep'], d['recurrence']['bwd_us_per_timestep']))")
    echo "[$wl $cfg] rep $rep: $r"
  done
done
