# Round-4 evidence in one box session (outputs under gpurun_out/r04/, copied to profiles/ by hand afterwards):
#   1. rocprofv3 --kernel-trace --stats of the default driver command (python bench.py) -> kernel stats + step timeline
#   2. FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, kernel-trace only) over tools/pmc_probe.py -> HBM bytes per launch
#      (+ a second pair over the ref_yaml layer: rnn_bwd_scatter2)
#   3. bench lines of every workload, phase accounting of the backward recurrences, MFMA / store-path microbenchmarks
set -u
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o cfg2 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r04_bench_under_rocprof.json 2> $O/stats.log )
db=$(find $O/stats -name "*.db" | head -1)
[ -n "$db" ] && python tools/prof_timeline.py $db -1 > $O/r04_cfg2_step_timeline.txt 2>&1
[ -n "$db" ] && python tools/prof_stats.py $db > $O/r04_cfg2_train_decode_kernel_stats.txt 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f -- python $R/tools/pmc_probe.py > $O/fetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w -- python $R/tools/pmc_probe.py > $O/write.log 2>&1 )
( cd /tmp && PMC_PROBE_SET=scatter2 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch2 -o f -- python $R/tools/pmc_probe.py > $O/fetch2.log 2>&1 )
( cd /tmp && PMC_PROBE_SET=scatter2 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write2 -o w -- python $R/tools/pmc_probe.py > $O/write2.log 2>&1 )
fd=$(find $O/fetch -name "*.db" | head -1); wd=$(find $O/write -name "*.db" | head -1); fd2=$(find $O/fetch2 -name "*.db" | head -1); wd2=$(find $O/write2 -name "*.db" | head -1)
[ -n "$fd" ] && [ -n "$wd" ] && python tools/pmc_to_json.py $fd $wd $O/r04_pmc_hbm_traffic.json $fd2 $wd2 > $O/pmc.log 2>&1
for wl in cfg1 cfg3 cfg4 ref_yaml; do
  timeout 400 python bench.py --workload $wl --steps 20 --warmup 3 --no-decode > $O/r04_bench_$wl.json 2> $O/bench_$wl.err
done
timeout 400 python bench.py --workload cfg2 --precision 0 --steps 10 --warmup 3 --no-decode --no-cpu-baseline > $O/r04_bench_cfg2_f32.json 2> $O/bench_f32.err
timeout 600 python bench.py --steps 20 --warmup 3 > $O/r04_bench_default.json 2> $O/bench_default.err
( timeout 200 ./tools/mb_bwd2.bin 320 32 800 > $O/r04_mb_bwd2_cfg2.txt 2>&1 ); ( timeout 100 ./tools/mb_bwd2.bin 384 8 400 > $O/r04_mb_bwd2_ref_yaml.txt 2>&1 ); ( timeout 100 ./tools/mb_bwd2.bin 512 64 1200 > $O/r04_mb_bwd2_h512.txt 2>&1 )
( timeout 60 ./tools/mb_mfma16.bin > $O/r04_mb_mfma16.txt 2>&1 ); ( timeout 60 ./tools/mb_store.bin > $O/r04_mb_store.txt 2>&1 )
{ echo "# tools/mb_gemm_pp.bin: cycles per barrier site (work before the barrier | wait inside it), waves 0 and 4 of block 0, instrumented build"
  echo "## plane tile 256 x 256 (gemm_planes_nt256pp_kernel<2>)"; timeout 60 ./tools/mb_gemm_pp.bin 76800 3072 1024 0
  echo "## float32-A tile 256 x 128 (dx GEMM of cfg2)"; timeout 60 ./tools/mb_gemm_pp.bin 25600 640 2560 1
  echo "## float32-A tile 256 x 256 (the bench probe shape)"; timeout 60 ./tools/mb_gemm_pp.bin 25600 1280 640 2
  echo "## TN tile 256 x 256 (dW_ih of cfg4)"; timeout 100 ./tools/mb_gemm_pp.bin 3072 1024 76800 3
  echo "## bare MFMA issue rate (tools/mb_mfma.bin)"; timeout 60 ./tools/mb_mfma.bin; } > $O/r04_mb_gemm_pp.txt 2>&1
{ for r in peaky flat; do timeout 120 python tools/mb_beam.py run $r 2>&1 | grep -v amdgpu.ids; done; } > $O/r04_mb_beam.txt 2>&1
if [ -f tools/libbeam_r3.so ] && [ -f tools/libbeam_r4.so ]; then
  { echo "# kernel time of one cfg5 batch (128 x 800 x 62, W = 20): the round-3 build of decode.hip against this round's, same box, same inputs"
    timeout 200 python tools/mb_beam.py time tools/libbeam_r3.so tools/libbeam_r4.so 2>&1 | grep -v amdgpu.ids
    echo "# fuzz batches (small alphabets, wide beams, 200 classes, empty utterances): both builds against the C oracle"
    timeout 200 python tools/beam_fuzz_ab.py tools/libbeam_r3.so tools/libbeam_r4.so 2>&1 | grep -v amdgpu.ids; } > $O/r04_beam_ab_vs_r03.txt 2>&1
fi
rm -rf $O/stats/*/*.db $O/fetch $O/write $O/fetch2 $O/write2 2>/dev/null
ls -la $O
for f in r04_bench_default r04_bench_cfg1 r04_bench_cfg3 r04_bench_cfg4 r04_bench_ref_yaml r04_bench_cfg2_f32; do python - "$O/$f.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "ms/step %.3f  value %.0f  fwd %.3f bwd %.3f  %s" % (d["ms_per_step"], d["value"], d["recurrence"]["fwd_us_per_timestep"], d["recurrence"]["bwd_us_per_timestep"], d["recurrence"]["bwd_kernel"]), "decode", (d.get("decode") or {}).get("value"), (d.get("decode") or {}).get("value_flat"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
