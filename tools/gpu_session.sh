#!/bin/bash
# One GPU-box session: smoke -> parity tests -> short bench -> rocprofv3 kernel stats.  Logs under gpurun_out/.
# usage: tools/gpu_session.sh [tests|bench|prof|all] (default all)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
what="${1:-all}"
export PYTHONDONTWRITEBYTECODE=1
if [[ "$what" == "all" || "$what" == "tests" ]]; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/summary.log
  timeout 1500 python -m pytest tests -m gpu -q -n 1 --timeout 300 -p no:cacheprovider ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?" | tee -a gpurun_out/summary.log
  tail -n 60 gpurun_out/pytest_gpu.log
fi
if [[ "$what" == "all" || "$what" == "bench" ]]; then
  timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" | tee -a gpurun_out/summary.log
  tail -n 3 gpurun_out/bench.log; tail -n 5 gpurun_out/bench.err
fi
if [[ "$what" == "all" || "$what" == "prof" ]]; then
  export TMPDIR=/tmp
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o cfg2 -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof.log" 2>&1 ); echo "prof rc=$?" | tee -a gpurun_out/summary.log
  find gpurun_out/prof -name "*kernel_stats*" | head -3
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 25 "$f"
fi
