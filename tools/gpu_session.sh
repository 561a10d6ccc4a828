#!/bin/bash
# The command list of ONE GPU-box session of round 6 (outputs under gpurun_out/s<N>/, which is scratch -- what is kept is copied to
# profiles/ by hand).  usage: tools/gpu_session.sh <N>
set -u
cd "$(dirname "$0")/.."; R=$PWD; S=${1:-1}; O=$R/gpurun_out/s$S; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
case $S in
1)
  #  the full parity suite twice in the suite's own order: as shipped, and with "xcd_interleave" forced onto the launches
  # that take every XCD (cfg4) -- the arrangement of the one divergent session of round 5.  Every test's process-state snapshot goes to
  # state_*.jsonl, every squat_stress trajectory (with the state it ran in) to traj_*.jsonl.
  CTCN_STATE_LOG=$O/state_shipped.jsonl CTCN_TRAJ_LOG=$O/traj_shipped.jsonl timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout 600 -p no:cacheprovider -p no:randomly > $O/pytest_shipped.log 2>&1; echo "pytest shipped rc=$?" > $O/summary.log
  tail -n 25 $O/pytest_shipped.log
  CTCN_OPT_XCD_INTERLEAVE_FORCE=1 CTCN_STATE_LOG=$O/state_forced.jsonl CTCN_TRAJ_LOG=$O/traj_forced.jsonl timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout 600 -p no:cacheprovider -p no:randomly > $O/pytest_forced.log 2>&1; echo "pytest forced rc=$?" >> $O/summary.log
  tail -n 8 $O/pytest_forced.log
  timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/summary.log
  cat $O/summary.log
  ;;
2)
  # the cfg4 nondeterminism of session 1 (three trajectories in one process): which tensor is it, and does dirtied memory move it
  for v in "" "--poison-ws nan" "--poison-ws rand" "--poison-pool nan" "--poison-pool rand" "--squat"; do
    n=40; [ -n "$v" ] && n=8; [ "$v" = "--squat" ] && n=30
    echo "== nondet_probe $v" >> $O/nondet.txt
    timeout 600 python tools/nondet_probe.py --workload cfg4 --reps $n $v 2>&1 | grep -v amdgpu.ids >> $O/nondet.txt
  done
  tail -n 40 $O/nondet.txt
  timeout 900 python -m pytest tests -m gpu -q --maxfail=8 --timeout 600 -p no:cacheprovider -k "bench_ or xcd_order or beam or dropout or conv_front" > $O/pytest_sub.log 2>&1; echo "pytest rc=$?" > $O/summary.log; tail -n 6 $O/pytest_sub.log
  timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/summary.log; tail -c 600 $O/bench_default.err
  cat $O/summary.log
  ;;
3)
  # hunting the cfg4 nondeterminism: traced trajectories (device-side checksums per step) in the contexts where it showed and where it did not
  for i in 1 2 3; do
    CTCN_TRAJ_LOG=$O/traj_foreign_$i.jsonl timeout 400 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -k "foreign" > $O/pytest_foreign_$i.log 2>&1; echo "foreign $i rc=$?" >> $O/summary.log
  done
  CTCN_TRAJ_LOG=$O/traj_full.jsonl timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout 600 -p no:cacheprovider > $O/pytest_full.log 2>&1; echo "full rc=$?" >> $O/summary.log
  ( timeout 400 python tools/traj_bisect.py cfg4 12 8 2>&1 | grep -v amdgpu.ids ) > $O/bisect_fresh.txt
  ( timeout 600 python tools/traj_bisect.py cfg4 12 8 squat cfg2 60 2>&1 | grep -v amdgpu.ids ) > $O/bisect_prelude.txt
  python tools/traj_compare.py $O/traj_*.jsonl > $O/traj_compare.txt 2>&1
  cat $O/summary.log $O/bisect_fresh.txt $O/bisect_prelude.txt $O/traj_compare.txt | cut -c1-300
  ;;
4)
  # statistics for the rare cfg4 divergence: 150 traced 12-step runs in one process on a fresh box (first thing in the session), squatters on every other run
  rocm-smi --showclocks --showtemp --showpower > $O/smi_before.txt 2>&1
  ( timeout 1500 python tools/traj_bisect.py cfg4 12 150 squat 2>&1 | grep -v "amdgpu.ids\|Warning" ) > $O/bisect_150.txt
  rocm-smi --showclocks --showtemp --showpower > $O/smi_after.txt 2>&1
  tail -n 12 $O/bisect_150.txt
  ;;
5)
  # the cfg4 divergence showed in 1 of 3 FULL suites (and never in 190 stand-alone runs): is it the tests that spawn second processes on the GPU
  # right before the squatter tests?  traced trajectories throughout
  for i in 1 2 3 4 5; do
    CTCN_TRAJ_LOG=$O/traj_sub_$i.jsonl timeout 500 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider -k "two_ranks or bench_ or rank_invariant or foreign" > $O/pytest_sub_$i.log 2>&1; echo "subset $i rc=$?" >> $O/summary.log
  done
  for i in 1 2 3; do
    CTCN_TRAJ_LOG=$O/traj_full_$i.jsonl timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout 600 -p no:cacheprovider > $O/pytest_full_$i.log 2>&1; echo "full $i rc=$?" >> $O/summary.log
  done
  rocm-smi --showrasinfo all > $O/ras.txt 2>&1
  python tools/traj_compare.py $O/traj_*.jsonl > $O/traj_compare.txt 2>&1
  cat $O/summary.log $O/traj_compare.txt | cut -c1-300
  ;;
6)
  # the cfg4 divergence localised: full suites with the finer trace (layer-0 buffers per (timestep, direction), buffer addresses)
  for i in 1 2 3 4; do
    CTCN_TRAJ_LOG=$O/traj_full_$i.jsonl timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout 600 -p no:cacheprovider > $O/pytest_full_$i.log 2>&1; echo "full $i rc=$?" >> $O/summary.log
    python tools/traj_compare.py $O/traj_full_$i.jsonl > $O/traj_compare_$i.txt 2>&1
    tail -n 3 $O/pytest_full_$i.log | cut -c1-200
  done
  python tools/traj_compare.py $O/traj_*.jsonl > $O/traj_compare.txt 2>&1
  cat $O/summary.log $O/traj_compare.txt | cut -c1-400
  # keep the logs of the runs that deviated, drop the bulky rest
  for i in 1 2 3 4; do grep -q "first difference" $O/traj_compare_$i.txt || rm -f $O/traj_full_$i.jsonl; done
  ;;
7)
  # the mitigation under test (option rnn_proj_order = 1, default): full traced suites -- the divergence showed in 2 of 4 of them with the old order
  for i in 1 2 3; do
    CTCN_TRAJ_LOG=$O/traj_full_$i.jsonl timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout 600 -p no:cacheprovider --durations=12 > $O/pytest_full_$i.log 2>&1; echo "full $i rc=$?" >> $O/summary.log
    python tools/traj_compare.py $O/traj_full_$i.jsonl > $O/traj_compare_$i.txt 2>&1
    tail -n 3 $O/pytest_full_$i.log | cut -c1-200
  done
  python tools/traj_compare.py $O/traj_*.jsonl > $O/traj_compare.txt 2>&1
  cat $O/summary.log $O/traj_compare.txt | cut -c1-400
  for i in 1 2 3; do grep -q "first difference" $O/traj_compare_$i.txt || rm -f $O/traj_full_$i.jsonl; done
  timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/summary.log
  CTCN_OPT_RNN_PROJ_ORDER=0 timeout 600 python bench.py --no-cpu-baseline --no-decode --no-pmc --no-ragged --no-sync-bn-cost > $O/bench_order0.json 2> $O/bench_order0.err
  ;;
8)
  # round-6 evidence (profiles/r06_*) + three more traced full suites with the projection order fix
  bash tools/run_profiles_r6.sh > $O/run_profiles.log 2>&1
  for i in 1 2 3; do
    CTCN_TRAJ_LOG=$O/traj_full_$i.jsonl timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout 600 -p no:cacheprovider --durations=6 > $O/pytest_full_$i.log 2>&1; echo "full $i rc=$?" >> $O/summary.log
    python tools/traj_compare.py $O/traj_full_$i.jsonl > $O/traj_compare_$i.txt 2>&1
    tail -n 3 $O/pytest_full_$i.log | cut -c1-200
    grep -q "first difference" $O/traj_compare_$i.txt || rm -f $O/traj_full_$i.jsonl
  done
  cat $O/summary.log $O/traj_compare_*.txt | cut -c1-300
  tail -n 12 $O/run_profiles.log | cut -c1-400
  ;;
9)
  # the kernel-stats pass of the driver command without the ragged-epoch legs (every recurrence launch is the T = 800 one), and two more suites
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o cfg2 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-others --no-pmc --no-ragged --no-sync-bn-cost > $O/r06_bench_under_rocprof.json 2> $O/stats.log )
  db=$(find $O/stats -name "*.db" | head -1)
  [ -n "$db" ] && python tools/prof_timeline.py $db -1 all > $O/r06_cfg2_step_timeline.txt 2>&1
  [ -n "$db" ] && python tools/prof_stats.py $db > $O/r06_cfg2_train_decode_kernel_stats.txt 2>&1
  rm -rf $O/stats
  head -n 6 $O/r06_cfg2_train_decode_kernel_stats.txt | cut -c1-160
  for i in 1 2; do
    CTCN_TRAJ_LOG=$O/traj_full_$i.jsonl timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout 600 -p no:cacheprovider --durations=6 > $O/pytest_full_$i.log 2>&1; echo "full $i rc=$?" >> $O/summary.log
    python tools/traj_compare.py $O/traj_full_$i.jsonl > $O/traj_compare_$i.txt 2>&1
    tail -n 3 $O/pytest_full_$i.log | cut -c1-200
    grep -q "first difference" $O/traj_compare_$i.txt || rm -f $O/traj_full_$i.jsonl
  done
  python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.log
  cat $O/summary.log $O/traj_compare_*.txt | cut -c1-300
  ;;
10)
  # option rnn_early_sum (item waves sum the parked partial tiles as the exchange waves finish them): parity subset, then A/B of the step
  timeout 900 python -m pytest tests -m gpu -q --maxfail=8 --timeout 600 -p no:cacheprovider -k "rnn or model_three or large_shape or lstm or gru or run_epoch or foreign or soak or stateless" > $O/pytest_sub.log 2>&1; echo "pytest rc=$?" > $O/summary.log; tail -n 5 $O/pytest_sub.log | cut -c1-200
  for rep in 1 2; do for v in 1 0; do
    CTCN_OPT_RNN_EARLY_SUM=$v timeout 600 python bench.py --no-cpu-baseline --no-decode --no-pmc --no-ragged --no-sync-bn-cost > $O/bench_early${v}_$rep.json 2> $O/bench_early${v}_$rep.err
  done; done
  python - $O <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_early*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "cfg2 %.3f ms (median %.3f)  fwd %.3f bwd %.3f us/step" % (d["ms_per_step"], d["ms_per_step_median"], d["recurrence"]["fwd_us_per_timestep"], d["recurrence"]["bwd_us_per_timestep"]),
              {k: (round(v["ms_per_step"], 3), round(v["fwd_us_per_timestep"], 3), round(v["bwd_us_per_timestep"], 3)) for k, v in d["other_workloads"].items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
  cat $O/summary.log
  ;;
11)
  # closing verification at HEAD: smoke, the full parity suite twice (traced), the driver's command
  python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.log
  for i in 1 2; do
    CTCN_TRAJ_LOG=$O/traj_full_$i.jsonl timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout 600 -p no:cacheprovider > $O/pytest_full_$i.log 2>&1; echo "full $i rc=$?" >> $O/summary.log
    python tools/traj_compare.py $O/traj_full_$i.jsonl > $O/traj_compare_$i.txt 2>&1
    tail -n 2 $O/pytest_full_$i.log | cut -c1-200
    grep -q "first difference" $O/traj_compare_$i.txt || rm -f $O/traj_full_$i.jsonl
  done
  timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/summary.log
  cat $O/summary.log $O/traj_compare_*.txt | cut -c1-300
  python - $O/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("cfg2 %.3f ms (median %.3f) value %.0f; roofline frac %.4f us/launch %.1f traffic %s; decode %s; wide %s / %s ms" % (
    d["ms_per_step"], d["ms_per_step_median"], d["value"], d["roofline"]["frac"], d["roofline"]["us_per_launch"], d["roofline"]["traffic"],
    {k: round(v["value"]) for k, v in d["decode"]["regimes"].items()}, round(d["decode"]["wide_beam"]["peaky"]["ms_per_batch"], 2), round(d["decode"]["wide_beam"]["flat"]["ms_per_batch"], 2)))
PY
  ;;
12)
  bash tools/flag_lottery.sh $O > $O/run.log 2>&1; cat $O/lottery.txt | cut -c1-250
  ;;
13)
  # a reproducer of the stale first reads outside the parity suite? (tools/stale_probe.py)
  { for args in "--iters 400" "--iters 300 --lag 0.02" "--iters 150 --lag 0.1" "--iters 200 --lag 0.05 --dirty 30" "--iters 100 --lag 0.3 --dirty 30" "--iters 200 --lag 0.05 --dirty 30 --order 1"; do
      echo "== stale_probe $args"; timeout 300 python tools/stale_probe.py $args 2>&1 | grep -v amdgpu.ids | tail -n 8
    done; } > $O/stale_probe.txt 2>&1
  cat $O/stale_probe.txt | cut -c1-250
  ;;
14)
  # the order fix A/B-ed IN the process that has just run the full suite (where the divergence lives), twice
  for i in 1 2; do
    CTCN_AFTER_SUITE=$R/tools/after_suite_ab.py CTCN_AFTER_SUITE_OUT=$O/after_suite_ab.json CTCN_AFTER_SUITE_N=25 timeout 1800 python -m pytest tests -m gpu -q --maxfail=20 --timeout 600 -p no:cacheprovider -s > $O/pytest_full_$i.log 2>&1; echo "full $i rc=$?" >> $O/summary.log
    grep "after_suite_ab" $O/pytest_full_$i.log | cut -c1-600
  done
  cat $O/summary.log
  ;;
15)
  # which process condition does the cfg4 divergence need?  after the suite: as-is | GC off | emptied allocator pool | as-is again; and a FRESH process whose host lags
  CTCN_AFTER_SUITE=$R/tools/after_suite_ab.py CTCN_AFTER_SUITE_OUT=$O/after_suite_ab.json CTCN_AFTER_SUITE_N=16 timeout 1800 python -m pytest tests -m gpu -q --maxfail=20 --timeout 600 -p no:cacheprovider -s > $O/pytest_full_1.log 2>&1; echo "full rc=$?" >> $O/summary.log
  grep "after_suite_ab" $O/pytest_full_1.log | cut -c1-1500
  ( CTCN_STEP_LAG=0.04 timeout 600 python tools/traj_bisect.py cfg4 12 40 squat 2>&1 | grep -v "amdgpu.ids\|Warning" ) > $O/bisect_lag.txt; tail -n 6 $O/bisect_lag.txt | cut -c1-300
  ;;
16)
  # which tests put the process into the state where cfg4 trajectories deviate?  (24 traced runs after each subset, in the subset's process)
  SPAWN="two_ranks or bench_ or rank_invariant or rank_failure or torchrun"
  i=0
  for sel in "not ($SPAWN)" "$SPAWN" "beam or decode or greedy or join" "rnn or lstm or gru"; do
    i=$((i+1))
    CTCN_AFTER_SUITE=$R/tools/after_suite_ab.py CTCN_AFTER_SUITE_OUT=$O/after_$i.json CTCN_AFTER_SUITE_N=24 CTCN_AFTER_SUITE_PHASES=order1 timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -s -k "$sel" > $O/pytest_$i.log 2>&1
    echo "subset $i [$sel]: $(grep -o '"summary": "[^"]*"' $O/pytest_$i.log)  $(tail -n 1 $O/pytest_$i.log | cut -c1-100)" | tee -a $O/summary.log
  done
  ;;
17)
  # narrowing: which of the recurrent-layer tests; and does the workspace's content matter (fresh process, garbage in the workspaces before every run)?
  i=0
  for sel in "item_gather or persistent_equals or batch_chunks or stateless or 4gb or unaligned or xcd_order" "model_three or rnn_layer or rnn_module or side_stream_equals or projection_overlap"; do
    i=$((i+1))
    CTCN_AFTER_SUITE=$R/tools/after_suite_ab.py CTCN_AFTER_SUITE_OUT=$O/after_$i.json CTCN_AFTER_SUITE_N=24 CTCN_AFTER_SUITE_PHASES=order1 timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -s -k "$sel" > $O/pytest_$i.log 2>&1
    echo "subset $i [$sel]: $(grep -o '"summary": "[^"]*"' $O/pytest_$i.log)  $(tail -n 1 $O/pytest_$i.log | cut -c1-100)" | tee -a $O/summary.log
  done
  ( CTCN_WS_FILL=rand timeout 600 python tools/traj_bisect.py cfg4 12 30 2>&1 | grep -v "amdgpu.ids\|Warning" ) > $O/bisect_wsrand.txt; echo "fresh process, random workspaces: $(tail -n 1 $O/bisect_wsrand.txt)" | tee -a $O/summary.log
  ;;
18)
  i=0
  for sel in "item_gather_equals or item_gather_edge" "persistent_equals or batch_chunks" "stateless or 4gb or unaligned" "xcd_order"; do
    i=$((i+1))
    CTCN_AFTER_SUITE=$R/tools/after_suite_ab.py CTCN_AFTER_SUITE_OUT=$O/after_$i.json CTCN_AFTER_SUITE_N=30 CTCN_AFTER_SUITE_PHASES=order1 timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -s -k "$sel" > $O/pytest_$i.log 2>&1
    echo "subset $i [$sel]: $(grep -o '"summary": "[^"]*"' $O/pytest_$i.log)  $(tail -n 1 $O/pytest_$i.log | cut -c1-100)" | tee -a $O/summary.log
  done
  ;;
19)
  # does an earlier persistent launch of cfg4's geometry at another T poison the process?  (T mod 4 decides the tags it leaves in the hand-off tiles)
  for T in 30 40 31 0; do ( timeout 400 python tools/prelude_ab.py $T 6 gru 30 2>&1 | grep -v "amdgpu.ids\|Warning" | tail -n 2 ) | tee -a $O/prelude.txt; done
  ;;
20)
  i=0
  for sel in "item_gather_equals" "item_gather_edge" "persistent_equals" "batch_chunks_equal" "batch_chunks_into_flat" "xcd_order"; do
    i=$((i+1))
    CTCN_AFTER_SUITE=$R/tools/after_suite_ab.py CTCN_AFTER_SUITE_OUT=$O/after_$i.json CTCN_AFTER_SUITE_N=30 CTCN_AFTER_SUITE_PHASES=order1 timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -s -k "$sel" > $O/pytest_$i.log 2>&1
    echo "subset $i [$sel]: $(grep -o '"summary": "[^"]*"' $O/pytest_$i.log)  $(tail -n 1 $O/pytest_$i.log | cut -c1-80)" | tee -a $O/summary.log
    grep "process state" $O/pytest_$i.log | cut -c1-400 | tee -a $O/summary.log
  done
  ;;
21)
  # WHERE does a deviating bottom layer leave the others (element pattern), is it the projection or the recurrence, and which arm removes it
  CTCN_AFTER_SUITE=$R/tools/first_rows_probe.py CTCN_PROBE_OUT=$O/probe.jsonl CTCN_PROBE_N=${PROBE_N:-1000} CTCN_PROBE_RUNS=${PROBE_RUNS:-30} timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -s -k "item_gather_edge or batch_chunks_equal" > $O/pytest.log 2>&1
  grep "^\[probe\]" $O/pytest.log | cut -c1-1800; tail -n 3 $O/pytest.log | cut -c1-300
  ;;
22)
  # does session 20's condition come back with the rnn_dbg build (a) after_suite_ab as in session 20, (b) the probe's runs part alone, squatters on every other run
  i=0
  for sel in "item_gather_edge" "batch_chunks_equal"; do
    i=$((i+1))
    CTCN_AFTER_SUITE=$R/tools/after_suite_ab.py CTCN_AFTER_SUITE_OUT=$O/after_$i.json CTCN_AFTER_SUITE_N=30 CTCN_AFTER_SUITE_PHASES=order1 timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -s -k "$sel" > $O/pytest_$i.log 2>&1
    echo "subset $i [$sel]: $(grep -o '"summary": "[^"]*"' $O/pytest_$i.log)  $(tail -n 1 $O/pytest_$i.log | cut -c1-80)" | tee -a $O/summary.log
  done
  CTCN_AFTER_SUITE=$R/tools/first_rows_probe.py CTCN_PROBE_OUT=$O/probe.jsonl CTCN_PROBE_PARTS=runs CTCN_PROBE_ARMS="base:" CTCN_PROBE_SQUAT=alt CTCN_PROBE_RUNS=40 timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -s -k "item_gather_edge or batch_chunks_equal" > $O/pytest.log 2>&1
  grep "^\[probe\]" $O/pytest.log | cut -c1-2500; tail -n 1 $O/pytest.log | cut -c1-300
  ;;
23)
  # is the condition a property of the BUILD?  the previous library (HEAD~rnn_dbg) and the rnn_dbg one, alternating on one box (tools/libctcn_{prev,dbg}.so are built by hand)
  cp ctc_pytorch_amd/libctcn.so $O/libctcn_keep.so
  i=0
  for rep in 1 2; do for lib in prev dbg; do for sel in "item_gather_edge" "batch_chunks_equal"; do
    i=$((i+1))
    cp tools/libctcn_$lib.so ctc_pytorch_amd/libctcn.so
    CTCN_AFTER_SUITE=$R/tools/after_suite_ab.py CTCN_AFTER_SUITE_OUT=$O/after_$i.json CTCN_AFTER_SUITE_N=30 CTCN_AFTER_SUITE_PHASES=order1 timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -s -k "$sel" > $O/pytest_$i.log 2>&1
    echo "$i lib $lib [$sel]: $(grep -o '"summary": "[^"]*"' $O/pytest_$i.log)  $(tail -n 1 $O/pytest_$i.log | cut -c1-80)" | tee -a $O/summary.log
  done; done; done
  cp $O/libctcn_keep.so ctc_pytorch_amd/libctcn.so; rm -f $O/libctcn_keep.so
  ;;
24)
  # the element pattern of a deviating bottom layer, with the library that shows the condition (tools/libctcn_prev.so)
  cp ctc_pytorch_amd/libctcn.so $O/libctcn_keep.so
  cp tools/libctcn_${LIBV:-prev}.so ctc_pytorch_amd/libctcn.so
  CTCN_AFTER_SUITE=$R/tools/first_rows_probe.py CTCN_PROBE_OUT=$O/probe.jsonl CTCN_PROBE_PARTS=runs CTCN_PROBE_ARMS="${ARMS:-base:,rsv0:fwd_rsv_lds=0,order0:rnn_proj_order=0}" CTCN_PROBE_SQUAT=alt CTCN_PROBE_RUNS=${PROBE_RUNS:-40} timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -s -k "batch_chunks_equal" > $O/pytest.log 2>&1
  cp $O/libctcn_keep.so ctc_pytorch_amd/libctcn.so; rm -f $O/libctcn_keep.so
  grep "^\[probe\]" $O/pytest.log | cut -c1-3000; tail -n 1 $O/pytest.log | cut -c1-300
  ;;
25)
  # a fast reproducer?  the stand-alone stress parts with the library that shows the condition, two alternating inputs (which older content does a stale read return)
  cp ctc_pytorch_amd/libctcn.so $O/libctcn_keep.so
  cp tools/libctcn_${LIBV:-prev}.so ctc_pytorch_amd/libctcn.so
  CTCN_AFTER_SUITE=$R/tools/first_rows_probe.py CTCN_PROBE_OUT=$O/probe.jsonl CTCN_PROBE_PARTS=${PARTS:-gemm,rec2,layer2} CTCN_PROBE_N=${PROBE_N:-4000} timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -s -k "batch_chunks_equal" > $O/pytest.log 2>&1
  cp $O/libctcn_keep.so ctc_pytorch_amd/libctcn.so; rm -f $O/libctcn_keep.so
  grep "^\[probe\]" $O/pytest.log | cut -c1-3000; tail -n 1 $O/pytest.log | cut -c1-300
  ;;
26)
  # the root cause under test: the SLOW instantiation (item waves delayed at every step) on the single-buffer build (must FAIL) and on the shipped
  # double-buffered one (must pass); then the after-suite statistics of session 23 for both
  cp ctc_pytorch_amd/libctcn.so $O/libctcn_keep.so
  for lib in single keep; do
    [ $lib = single ] && cp tools/libctcn_single.so ctc_pytorch_amd/libctcn.so || cp $O/libctcn_keep.so ctc_pytorch_amd/libctcn.so
    timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -k "slow_item_waves" > $O/pytest_slow_$lib.log 2>&1
    echo "lib $lib, test_rnn_fwd_tagged_with_slow_item_waves: $(tail -n 1 $O/pytest_slow_$lib.log | cut -c1-120)" | tee -a $O/summary.log
    grep "^E .*differs" $O/pytest_slow_$lib.log | cut -c1-200 | head -n 6 | tee -a $O/summary.log
  done
  i=0
  for lib in keep single keep single; do for sel in "batch_chunks_equal"; do
    i=$((i+1))
    [ $lib = single ] && cp tools/libctcn_single.so ctc_pytorch_amd/libctcn.so || cp $O/libctcn_keep.so ctc_pytorch_amd/libctcn.so
    CTCN_AFTER_SUITE=$R/tools/after_suite_ab.py CTCN_AFTER_SUITE_OUT=$O/after_$i.json CTCN_AFTER_SUITE_N=40 CTCN_AFTER_SUITE_PHASES=order1 timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -s -k "$sel" > $O/pytest_$i.log 2>&1
    echo "$i lib $lib [$sel]: $(grep -o '"summary": "[^"]*"' $O/pytest_$i.log)  $(tail -n 1 $O/pytest_$i.log | cut -c1-80)" | tee -a $O/summary.log
  done; done
  cp $O/libctcn_keep.so ctc_pytorch_amd/libctcn.so; rm -f $O/libctcn_keep.so
  ;;
27)
  # verification at HEAD (double-buffered parked tiles): smoke, the full parity suite (traced), the driver's command
  python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.log
  for i in 1; do
    CTCN_TRAJ_LOG=$O/traj_full_$i.jsonl timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout 600 -p no:cacheprovider --durations=8 > $O/pytest_full_$i.log 2>&1; echo "full $i rc=$?" >> $O/summary.log
    python tools/traj_compare.py $O/traj_full_$i.jsonl > $O/traj_compare_$i.txt 2>&1
    tail -n 12 $O/pytest_full_$i.log | cut -c1-200
    grep -q "first difference" $O/traj_compare_$i.txt || rm -f $O/traj_full_$i.jsonl
  done
  timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/summary.log
  cat $O/summary.log $O/traj_compare_*.txt | cut -c1-300
  python - $O/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("cfg2 %.3f ms (median %.3f) value %.0f; roofline frac %.4f us/launch %.1f traffic %s; decode %s; wide %s / %s ms" % (
    d["ms_per_step"], d["ms_per_step_median"], d["value"], d["roofline"]["frac"], d["roofline"]["us_per_launch"], d["roofline"]["traffic"],
    {k: round(v["value"]) for k, v in d["decode"]["regimes"].items()}, round(d["decode"]["wide_beam"]["peaky"]["ms_per_batch"], 2), round(d["decode"]["wide_beam"]["flat"]["ms_per_batch"], 2)))
print("recurrence", d["recurrence"], {k: (round(v["ms_per_step"], 3), round(v["fwd_us_per_timestep"], 3), round(v["bwd_us_per_timestep"], 3)) for k, v in d["other_workloads"].items()})
PY
  ;;
28)
  # A/B of a forward-recurrence edit: parity subset, then cfg4 / cfg2 steps (LIBB = the library to compare with, tools/libctcn_<LIBB>.so)
  timeout 900 python -m pytest tests -m gpu -q --maxfail=8 --timeout 600 -p no:cacheprovider -k "rnn or model_three or large_shape or lstm or gru or run_epoch or foreign or soak or stateless or full_size" > $O/pytest_sub.log 2>&1; echo "pytest rc=$?" > $O/summary.log; tail -n 3 $O/pytest_sub.log | cut -c1-200
  cp ctc_pytorch_amd/libctcn.so $O/libctcn_keep.so
  for rep in 1 2; do for lib in keep ${LIBB:-base}; do
    [ $lib = keep ] && cp $O/libctcn_keep.so ctc_pytorch_amd/libctcn.so || cp tools/libctcn_$lib.so ctc_pytorch_amd/libctcn.so
    for wl in cfg4 cfg2; do
      timeout 400 python bench.py --workload $wl --steps 20 --warmup 3 --no-decode --no-cpu-baseline --no-others --no-pmc --no-ragged --no-sync-bn-cost > $O/bench_${lib}_${wl}_$rep.json 2> $O/bench_${lib}_${wl}_$rep.err
    done
  done; done
  cp $O/libctcn_keep.so ctc_pytorch_amd/libctcn.so; rm -f $O/libctcn_keep.so
  python - $O <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "%.3f ms (median %.3f)  fwd %.3f bwd %.3f us/step" % (d["ms_per_step"], d["ms_per_step_median"], d["recurrence"]["fwd_us_per_timestep"], d["recurrence"]["bwd_us_per_timestep"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
  cat $O/summary.log
  ;;
29)
  # cfg4 forward: sweep of the exchange waves' pause before their first poll (option tag_poll_delay, x 64 cycles; 8 was tuned at cfg2)
  for v in ${SWEEP:-8 12 16 20 24 8}; do
    CTCN_OPT_TAG_POLL_DELAY=$v timeout 400 python bench.py --workload ${WL:-cfg4} --steps 20 --warmup 3 --no-decode --no-cpu-baseline --no-others --no-pmc --no-ragged --no-sync-bn-cost > $O/bench_$v.json 2> $O/bench_$v.err
    python - $O/bench_$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("tag_poll_delay %s: %.3f ms (median %.3f)  fwd %.3f bwd %.3f us/step" % (sys.argv[2], d["ms_per_step"], d["ms_per_step_median"], d["recurrence"]["fwd_us_per_timestep"], d["recurrence"]["bwd_us_per_timestep"]))
PY
  done
  ;;
36)
  # after the fix: the statistics of session 20 (traced cfg4 runs in the process of the subsets that showed the divergence) at HEAD
  i=0
  for sel in "item_gather_equals" "item_gather_edge" "batch_chunks_equal" "xcd_order or batch_chunks_into_flat"; do
    i=$((i+1))
    CTCN_AFTER_SUITE=$R/tools/after_suite_ab.py CTCN_AFTER_SUITE_OUT=$O/after_$i.json CTCN_AFTER_SUITE_N=40 CTCN_AFTER_SUITE_PHASES=order1 timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -s -k "$sel" > $O/pytest_$i.log 2>&1
    echo "subset $i [$sel]: $(grep -o '"summary": "[^"]*"' $O/pytest_$i.log)  $(tail -n 1 $O/pytest_$i.log | cut -c1-80)" | tee -a $O/summary.log
  done
  ;;
37)
  # code-placement lottery: rnn.hip rebuilt with block / function alignment switches, cfg2 step, alternating on one box
  cp ctc_pytorch_amd/libctcn.so $O/libctcn_keep.so
  for rep in 1 2; do for lib in keep ${LIBS:-fn4k nft6 nola}; do
    [ $lib = keep ] && cp $O/libctcn_keep.so ctc_pytorch_amd/libctcn.so || cp tools/libctcn_$lib.so ctc_pytorch_amd/libctcn.so
    for wl in ${WLS:-cfg2}; do
      timeout 400 python bench.py --workload $wl --steps 20 --warmup 3 --no-decode --no-cpu-baseline --no-others --no-pmc --no-ragged --no-sync-bn-cost > $O/bench_${lib}_${wl}_$rep.json 2> $O/bench_${lib}_${wl}_$rep.err
    done
  done; done
  cp $O/libctcn_keep.so ctc_pytorch_amd/libctcn.so; rm -f $O/libctcn_keep.so
  python - $O <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "%.3f ms (median %.3f)  fwd %.3f bwd %.3f us/step  loss %r" % (d["ms_per_step"], d["ms_per_step_median"], d["recurrence"]["fwd_us_per_timestep"], d["recurrence"]["bwd_us_per_timestep"], d.get("final_loss")))
    except Exception as e:
        print(f, "unreadable", e)
PY
  ;;
esac
