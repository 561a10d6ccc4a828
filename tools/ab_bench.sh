# A/B of library options on the training step inside ONE box session: tools/ab_bench.sh "NAME=V NAME2=V" "..." ...
cd "$GRAFT_REPO_ROOT"
for cfg in "$@"; do
  envs=""
  for kv in $cfg; do envs="$envs CTCN_OPT_${kv%%=*}=${kv##*=}"; done
  for rep in 1 2; do
    r=$(env $envs python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-decode 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  fwd %.3f bwd %.3f us' % (d['ms_per_step'], d['recurrence']['fwd_us_per_timestep'], d['recurrence']['bwd_us_per_timestep']))")
    echo "[$cfg] rep $rep: $r"
  done
done
