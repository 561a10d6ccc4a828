set -u
cd "$GRAFT_REPO_ROOT"
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out/r3x
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3x/pytest_gpu.txt
cat gpurun_out/r3x/pytest_gpu.txt
timeout 300 python bench.py > gpurun_out/r3x/bench_default.json 2> gpurun_out/r3x/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3x/bench_default.json').read().strip().splitlines()[-1])
print('cfg2 ms/step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'value', d['value'])
print('recurrence', {k: d['recurrence'][k] for k in ('fwd_us_per_timestep','bwd_us_per_timestep','fwd_kernel','bwd_kernel') if k in d['recurrence']})
print('gemm', d['roofline_gemm']['achieved'], 'decode', d['decode']['value'], d['decode'].get('value_flat'))
PY
