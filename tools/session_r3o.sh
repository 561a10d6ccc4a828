set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3o; rm -rf $O; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1800 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"
tail -4 $O/pytest_all.log
for wl in cfg1 cfg3 cfg4 ref_yaml; do
  timeout 400 python bench.py --workload $wl --steps 20 --warmup 3 --no-decode > $O/r03_bench_$wl.json 2> $O/bench_$wl.err
done
timeout 600 python bench.py --steps 20 --warmup 3 > $O/r03_bench_default.json 2> $O/bench_default.err
for f in r03_bench_default r03_bench_cfg1 r03_bench_cfg3 r03_bench_cfg4 r03_bench_ref_yaml; do python - "$O/$f.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "ms/step %.3f (median %.3f) value %.0f  fwd %.3f bwd %.3f  %s/%s epoch %.3f" % (d["ms_per_step"], d["ms_per_step_median"], d["value"], d["recurrence"]["fwd_us_per_timestep"], d["recurrence"]["bwd_us_per_timestep"], d["recurrence"]["fwd_kernel"], d["recurrence"]["bwd_kernel"], d["epoch_loop"]["ms_per_step"]), "decode", (d.get("decode") or {}).get("value"), (d.get("decode") or {}).get("value_flat"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
