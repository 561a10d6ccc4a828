# Round-6 evidence in one box session (outputs under gpurun_out/r06/, copied to profiles/ by hand afterwards):
#   1. rocprofv3 --kernel-trace --stats of the default driver command (python bench.py; --no-pmc: no profiler inside the profiler; --no-ragged: every
#      recurrence launch of the pass is the T = 800 launch the roofline object times -- the ragged epochs' shorter launches would pull the average down)
#      -> kernel stats + step timeline (main and side stream) of the cfg2 step and the cfg5 decode leg
#   2. the same for cfg4 (every stream listed, so that who waits for whom can be read off the absolute times), cfg3, ref_yaml
#   3. the default driver command as the driver runs it (roofline.traffic measured in-run by two rocprofv3 --pmc child passes, other_workloads)
#   4. FETCH_SIZE / WRITE_SIZE PMC passes over tools/pmc_probe.py (every kernel whose traffic DESIGN.md quotes) -> r06_pmc_hbm_traffic.json
#   5. fresh cycle accounts of the shipped recurrences and of the beam search, the conv phase probe, a short soak
set -u
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o cfg2 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-others --no-pmc --no-ragged --no-sync-bn-cost > $O/r06_bench_under_rocprof.json 2> $O/stats.log )
db=$(find $O/stats -name "*.db" | head -1)
[ -n "$db" ] && python tools/prof_timeline.py $db -1 all > $O/r06_cfg2_step_timeline.txt 2>&1
[ -n "$db" ] && python tools/prof_stats.py $db > $O/r06_cfg2_train_decode_kernel_stats.txt 2>&1
rm -rf $O/stats
for wl in cfg4 cfg3; do
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_$wl -o p -- python $R/bench.py --workload $wl --steps 8 --warmup 2 --no-decode --no-cpu-baseline --no-others --no-pmc > $O/r06_${wl}_under_rocprof.json 2> $O/prof_$wl.log )
  db=$(find $O/prof_$wl -name "*.db" | head -1)
  [ -n "$db" ] && python tools/prof_timeline.py $db -1 all > $O/r06_${wl}_step_timeline.txt 2>&1 && python tools/prof_stats.py $db > $O/r06_${wl}_kernel_stats.txt 2>&1
  rm -rf $O/prof_$wl
done
timeout 900 python bench.py --steps 20 --warmup 3 > $O/r06_bench_default.json 2> $O/bench_default.err
timeout 400 python bench.py --workload cfg2 --precision 0 --steps 10 --warmup 3 --no-decode --no-cpu-baseline --no-others --no-pmc > $O/r06_bench_cfg2_f32.json 2> $O/bench_f32.err
for wl in cfg1 cfg3 cfg4 ref_yaml; do
  timeout 400 python bench.py --workload $wl --steps 20 --warmup 3 --no-decode > $O/r06_bench_$wl.json 2> $O/bench_$wl.err
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f -- python $R/tools/pmc_probe.py > $O/fetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w -- python $R/tools/pmc_probe.py > $O/write.log 2>&1 )
( cd /tmp && PMC_PROBE_SET=scatter2 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch2 -o f -- python $R/tools/pmc_probe.py > $O/fetch2.log 2>&1 )
( cd /tmp && PMC_PROBE_SET=scatter2 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write2 -o w -- python $R/tools/pmc_probe.py > $O/write2.log 2>&1 )
fd=$(find $O/fetch -name "*.db" | head -1); wd=$(find $O/write -name "*.db" | head -1); fd2=$(find $O/fetch2 -name "*.db" | head -1); wd2=$(find $O/write2 -name "*.db" | head -1)
[ -n "$fd" ] && [ -n "$wd" ] && python tools/pmc_to_json.py $fd $wd $O/r06_pmc_hbm_traffic.json $fd2 $wd2 > $O/pmc.log 2>&1
rm -rf $O/fetch $O/write $O/fetch2 $O/write2
{ echo "# tools/mb_step.bin (cfg2 layer: H 320, B 32, T 800): in-kernel clock64 stamps of the shipped forward / backward recurrences, round 6 HEAD"; timeout 200 ./tools/mb_step.bin; } > $O/r06_mb_step.txt 2>&1
( timeout 200 ./tools/mb_bwd2.bin 320 32 800 > $O/r06_mb_bwd2_cfg2.txt 2>&1 ); ( timeout 100 ./tools/mb_bwd2.bin 384 8 400 > $O/r06_mb_bwd2_ref_yaml.txt 2>&1 ); ( timeout 100 ./tools/mb_bwd2.bin 512 64 1200 > $O/r06_mb_bwd2_h512.txt 2>&1 )
{ for r in peaky flat; do timeout 120 python tools/mb_beam.py run $r 2>&1 | grep -v amdgpu.ids; done; for r in peaky flat; do timeout 200 python tools/mb_beam.py generic $r 200 2>&1 | grep -v amdgpu.ids; done; } > $O/r06_mb_beam.txt 2>&1
{ echo "# tools/conv_phase_probe.py: us per ctcn_conv2d_fwd launch with phases switched off (conv_dbg: 1 = no window load, 2 = no MFMA loop, 4 = no output phase)"; timeout 200 python tools/conv_phase_probe.py 2>&1 | grep -v amdgpu.ids; } > $O/r06_conv_phase_probe.txt
timeout 900 python tools/soak.py --cfg2 600 --cfg4 300 --ref-yaml 600 --cfg1 600 --cfg3 400 --decode 100 --out $O/r06_soak.json > $O/soak.out 2> $O/soak.err; echo "soak rc=$?"
ls -la $O
for f in r06_bench_default r06_bench_cfg1 r06_bench_cfg3 r06_bench_cfg4 r06_bench_ref_yaml r06_bench_cfg2_f32; do python - "$O/$f.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "ms/step %.3f  value %.0f  fwd %.3f bwd %.3f  %s" % (d["ms_per_step"], d["value"], d["recurrence"]["fwd_us_per_timestep"], d["recurrence"]["bwd_us_per_timestep"], d["recurrence"]["bwd_kernel"]), "decode", (d.get("decode") or {}).get("value"), (d.get("decode") or {}).get("value_flat"),
          {k: round(v.get("ms_per_step", -1), 3) for k, v in (d.get("other_workloads") or {}).items()}, (d.get("roofline") or {}).get("traffic"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
