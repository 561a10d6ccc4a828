set -u
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06b; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -30 > $O/gpu_suite.txt
tail -3 $O/gpu_suite.txt
timeout 600 python bench.py --steps 20 --warmup 3 > $O/r06_bench_default.json 2> $O/bench_default.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o cfg2 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-others --no-pmc --no-ragged --no-sync-bn-cost > $O/r06_bench_under_rocprof.json 2> $O/stats.log )
db=$(find $O/stats -name "*.db" | head -1)
[ -n "$db" ] && python tools/prof_timeline.py $db -1 all > $O/r06_cfg2_step_timeline.txt 2>&1
[ -n "$db" ] && python tools/prof_stats.py $db > $O/r06_cfg2_train_decode_kernel_stats.txt 2>&1
rm -rf $O/stats
{ for r in peaky flat; do timeout 120 python tools/mb_beam.py run $r 2>&1 | grep -v amdgpu.ids; done; for r in peaky flat; do timeout 200 python tools/mb_beam.py generic $r 200 2>&1 | grep -v amdgpu.ids; done; } > $O/r06_mb_beam.txt 2>&1
WB_CONFIGS=0:0,512:1 WB_NS=3,12 WB_REPS=16 timeout 140 python tools/wide_beam_probe.py 200 2>&1 | grep -v amdgpu.ids > $O/wide_beam_probe_final.txt
python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r06b")
d = json.loads(open(os.path.join(O, "r06_bench_default.json")).read().strip().splitlines()[-1])
print("ms/step %.3f value %.0f decode %.0f %.0f" % (d["ms_per_step"], d["value"], d["decode"]["value"], d["decode"]["value_flat"]))
print(json.dumps(d["decode"]["wide_beam"])[:1200])
print({k: round(v.get("ms_per_step", -1), 3) for k, v in d["other_workloads"].items()})
PY
cat $O/wide_beam_probe_final.txt
