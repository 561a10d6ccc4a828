#!/bin/bash
# Does the persistent recurrences' timing respond to the code generator's scheduling switches?  (Round 6: a dead code path cost the forward kernel 10 %,
# profiles/r06_early_sum_ab.txt -- so the opposite question is worth one box session.)  rnn.hip is recompiled with each flag set, linked with the other
# objects of the in-tree build into /tmp, and the cfg2 step + both recurrences are timed through CTCN_LIBCTCN.  usage: tools/flag_lottery.sh <outdir>
set -u
cd "$(dirname "$0")/.."; R=$PWD; O=${1:-$R/gpurun_out/lottery}; mkdir -p $O
OBJ=$R/ctc_pytorch_amd/csrc/_obj
others=$(ls $OBJ/*.o | grep -v "/rnn.o")
i=0
while IFS= read -r flags; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $R/ctc_pytorch_amd/csrc/rnn.hip -o /tmp/rnn_v$i.o 2> $O/build_$i.err || { echo "variant $i [$flags]: build failed" | tee -a $O/lottery.txt; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o /tmp/libctcn_v$i.so $others /tmp/rnn_v$i.o 2>> $O/build_$i.err || { echo "variant $i: link failed" | tee -a $O/lottery.txt; continue; }
  for rep in 1 2; do
    CTCN_LIBCTCN=/tmp/libctcn_v$i.so timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-decode --no-pmc --no-ragged --no-sync-bn-cost --no-others > $O/bench_${i}_$rep.json 2> $O/bench_${i}_$rep.err
    python - "$O/bench_${i}_$rep.json" "$i" "$flags" <<'PY' | tee -a $O/lottery.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("variant %s [%s]: cfg2 %.3f ms (median %.3f)  fwd %.3f bwd %.3f us/step  final loss %r" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["ms_per_step_median"],
          d["recurrence"]["fwd_us_per_timestep"], d["recurrence"]["bwd_us_per_timestep"], d["final_loss"]))
except Exception as e:
    print("variant %s [%s]: unreadable %r" % (sys.argv[2], sys.argv[3], e))
PY
  done
done <<'FLAGS'

-mllvm -amdgpu-schedule-metric-bias=5
-mllvm -amdgpu-schedule-metric-bias=20
-mllvm -enable-post-misched=0
-mllvm -amdgpu-use-amdgpu-trackers=1
-mllvm -amdgpu-disable-unclustered-high-rp-reschedule
-mllvm -greedy-reverse-local-assignment=1
-O2
FLAGS
