# forward poll delay (option tag_poll_delay: 64-cycle sleeps before an exchange wave's first poll) per workload, inside one box session
cd "$GRAFT_REPO_ROOT"
for wl in "$@"; do
  for pd in 8 12 14; do
    r=$(env CTCN_OPT_tag_poll_delay=$pd python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-decode 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  fwd %.3f bwd %.3f us' % (d['ms_per_step'], d['recurrence']['fwd_us_per_timestep'], d['recurrence']['bwd_us_per_timestep']))")
    echo "[$wl tag_poll_delay=$pd] $r"
  done
done
