# round-3 session A: new tests (squatter, element-wise full-size parity, shipped YAML, rank-invariant overlap) + whole GPU suite + bench lines
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3a; rm -rf $O; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -s -k "elementwise or shipped or foreign or rank_invariant" > $O/pytest_new.log 2>&1; echo "pytest new rc=$?"
grep -E "^\[|passed|failed|Error|error" $O/pytest_new.log | tail -40
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"
tail -5 $O/pytest_all.log
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 300 python bench.py --workload ref_yaml --steps 20 --warmup 3 --no-decode > $O/bench_ref_yaml.json 2> $O/bench_ref_yaml.err; echo "bench ref_yaml rc=$?"
timeout 300 python tools/squat_stress.py --steps 200 --out $O/squat_stress_cfg2.json > $O/squat_cfg2.log 2>&1; echo "squat cfg2 rc=$?"
timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-decode > $O/bench_gpus2.log 2>&1; echo "bench --gpus 2 on a 1-GPU box rc=$? (expected non-zero, from inside the runtime)"
tail -c 1500 $O/bench_gpus2.log | grep -iE "error|invalid|ordinal|assert" | head -5
python - <<'PY'
import json
for f in ("bench_default", "bench_ref_yaml"):
    try:
        d = json.loads(open("gpurun_out/r3a/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "ms/step %.3f median %.3f value %.0f" % (d["ms_per_step"], d.get("ms_per_step_median") or -1, d["value"]), d.get("recurrence", {}).get("fwd_us_per_timestep"), d.get("recurrence", {}).get("bwd_us_per_timestep"), d.get("roofline", {}).get("kernel"), d.get("comm"))
    except Exception as e:
        print(f, "unreadable", e)
PY
cat $O/squat_stress_cfg2.json 2>/dev/null | head -20
