set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3l; rm -rf $O; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider -k "batchnorm or conv_front or model_three or sync_batch or shipped" > $O/pytest_bn.log 2>&1; echo "pytest bn rc=$?"; tail -3 $O/pytest_bn.log
for mi in 2097152 600000 300000; do
  for wl in ref_yaml cfg1 cfg3; do
    r=$(CTCN_SIDE_MIN_ITEMS=$mi timeout 300 python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-decode 2>$O/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step (median %.3f)  epoch loop %.3f' % (d['ms_per_step'], d['ms_per_step_median'], d['epoch_loop']['ms_per_step']))" 2>&1)
    echo "[min_items $mi] $wl: $r"
  done
done
