cd "$GRAFT_REPO_ROOT"
for wl in ref_yaml cfg1; do
  for mi in 2097152 500000 200000; do
    for ov in 1 0; do
    r=$(env CTCN_SIDE_MIN_ITEMS=$mi CTCN_FWD_OVERLAP=$ov python bench.py --workload $wl --steps 40 --warmup 5 --no-cpu-baseline --no-decode 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  fwd %.3f bwd %.3f us' % (d['ms_per_step'], d['recurrence']['fwd_us_per_timestep'], d['recurrence']['bwd_us_per_timestep']))")
    echo "[$wl min_items=$mi fwd_overlap=$ov] $r"
    done
  done
done
