set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3m; rm -rf $O; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
for ss in 0 1 0 1; do
  for wl in ref_yaml cfg1; do
    r=$(CTCN_SMALL_SPLIT=$ss timeout 300 python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-decode 2>$O/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step (median %.3f)  epoch loop %.3f' % (d['ms_per_step'], d['ms_per_step_median'], d['epoch_loop']['ms_per_step']))" 2>&1)
    echo "[small_split $ss] $wl: $r"
  done
done
timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider -k "model_three or run_epoch or shipped or side_stream or end_to_end or data_parallel" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
