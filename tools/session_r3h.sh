set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3h; rm -rf $O; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
timeout 100 ./tools/mb_bwd2.bin 320 32 800 | grep -E "round 2|gather formulation"
timeout 100 ./tools/mb_bwd2.bin 128 8 300 | grep -E "round 2"
for rep in 1 2; do for wl in cfg2 cfg1; do
  r=$(timeout 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-decode 2>$O/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  fwd %.3f bwd %.3f us  %s %s' % (d['ms_per_step'], d['recurrence']['fwd_us_per_timestep'], d['recurrence']['bwd_us_per_timestep'], d['recurrence']['fwd_kernel'], d['recurrence']['bwd_kernel']))" 2>&1)
  echo "$wl: $r"
done; done
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider -k "rnn or model_three or fused_dropout or side_stream or elementwise" > $O/pytest_rnn.log 2>&1; echo "pytest rnn rc=$?"
tail -3 $O/pytest_rnn.log
