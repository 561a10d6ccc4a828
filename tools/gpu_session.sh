#!/bin/bash
# The command list of ONE GPU-box session (overwritten per session; outputs under gpurun_out/s<N>/, which is scratch -- what is kept is
# copied to profiles/ by hand).  usage: tools/gpu_session.sh <N>
set -u
cd "$(dirname "$0")/.."; R=$PWD; S=${1:-1}; O=$R/gpurun_out/s$S; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
case $S in
1)
  # full parity suite at HEAD (new: wide beams, reference-default BeamDecoder, cfg4 at B = 64, chunks into flat gradients)
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/summary.log
  tail -n 15 $O/pytest_gpu.log
  # fresh cycle accounts of the shipped recurrences
  ( timeout 200 ./tools/mb_step.bin > $O/mb_step.txt 2>&1 ); ( timeout 200 ./tools/mb_bwd2.bin 320 32 800 > $O/mb_bwd2_cfg2.txt 2>&1 ); ( timeout 100 ./tools/mb_bwd2.bin 512 64 1200 > $O/mb_bwd2_h512.txt 2>&1 )
  # cfg4: step timeline with the side stream listed, and the A/B of the direction split
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/cfg4prof -o cfg4 -- python $R/bench.py --workload cfg4 --steps 6 --warmup 2 --no-decode --no-cpu-baseline --no-others > $O/cfg4_under_rocprof.json 2> $O/cfg4prof.log )
  db=$(find $O/cfg4prof -name "*.db" | head -1)
  [ -n "$db" ] && python tools/prof_timeline.py $db -1 all > $O/cfg4_step_timeline_all.txt 2>&1 && python tools/prof_stats.py $db > $O/cfg4_kernel_stats.txt 2>&1
  rm -rf $O/cfg4prof
  for v in 1 0; do CTCN_SMALL_SPLIT=$v timeout 300 python bench.py --workload cfg4 --steps 15 --warmup 3 --no-decode --no-cpu-baseline --no-others > $O/cfg4_smallsplit$v.json 2> $O/cfg4_smallsplit$v.err; done
  # beam search: two workgroups per CU (option beam_occ2: 64-VGPR build, <= 80 KB LDS; 2 = LM in global memory, larger trie) x searches in flight
  for occ in 0 1 2; do for ns in 3 4; do
    CTCN_OPT_BEAM_OCC2=$occ CTCN_DECODE_STREAMS=$ns timeout 200 python bench.py --mode decode --steps 5 > $O/decode_occ${occ}_ns$ns.json 2> $O/decode_occ${occ}_ns$ns.err
  done; done
  { for r in peaky flat; do timeout 120 python tools/mb_beam.py run $r 2>&1 | grep -v amdgpu.ids; done; } > $O/mb_beam.txt 2>&1
  # the default driver command with the new other_workloads object
  timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/summary.log
  ;;
2)
  # the new two-rank test on the benchmarked kernels, alone (two processes share the GPU: bounded by its own timeout)
  timeout 400 python -m pytest tests -m gpu -q --timeout 350 -p no:cacheprovider -k "two_ranks_on_the_benchmarked" > $O/pytest_dp.log 2>&1; echo "pytest rc=$?" > $O/summary.log; tail -n 5 $O/pytest_dp.log
  # step timelines of every workload at HEAD (main stream + side stream)
  for wl in cfg3 ref_yaml cfg1 cfg2; do
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$wl -o p -- python $R/bench.py --workload $wl --steps 6 --warmup 2 --no-decode --no-cpu-baseline --no-others > $O/${wl}_under_rocprof.json 2> $O/prof_$wl.log )
    db=$(find $O/prof_$wl -name "*.db" | head -1)
    [ -n "$db" ] && python tools/prof_timeline.py $db -1 all > $O/${wl}_step_timeline_all.txt 2>&1 && python tools/prof_stats.py $db > $O/${wl}_kernel_stats.txt 2>&1
    rm -rf $O/prof_$wl
  done
  ;;
3)
  # bottom-layer weight GEMMs on XCD halves (CTCN_SPLIT_HALVES) x TN split count sized for the allowed XCDs (tn_splits_xcd)
  for wl in cfg2 cfg3; do for h in 0 1; do for x in 0 1; do
    CTCN_SPLIT_HALVES=$h CTCN_OPT_TN_SPLITS_XCD=$x timeout 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-decode --no-cpu-baseline --no-others > $O/${wl}_halves${h}_xcd$x.json 2> $O/${wl}_halves${h}_xcd$x.err
  done; done; done
  timeout 300 python bench.py --workload cfg4 --steps 15 --warmup 3 --no-decode --no-cpu-baseline --no-others > $O/cfg4.json 2> $O/cfg4.err
  timeout 600 python -m pytest tests -m gpu -q --maxfail=5 --timeout 600 -p no:cacheprovider -k "full_size or large_shape or model_ or batch_chunks or soak or persistent" > $O/pytest_sub.log 2>&1; echo "pytest rc=$?" > $O/summary.log; tail -n 4 $O/pytest_sub.log
  ;;
4)
  timeout 600 python -m pytest tests -m gpu -q --maxfail=5 --timeout 600 -p no:cacheprovider -k "bench_two_ranks or bench_under_torchrun or two_ranks" > $O/pytest_sub.log 2>&1; echo "pytest rc=$?" > $O/summary.log; tail -n 6 $O/pytest_sub.log
  timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/summary.log
  python - $O/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(json.dumps(d["roofline"], indent=1))
PY
  ;;
5)
  for i in 1 2 3; do timeout 200 python bench.py --mode decode --steps 5 > $O/decode_$i.json 2> $O/decode_$i.err; done
  python - $O <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/decode_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], {k: (round(v["value"]), round(v["ms_per_batch"], 3), round(v["kernel_us_per_batch"]), {a: round(b) for a, b in v["host_us_per_batch"].items()}) for k, v in d["regimes"].items()})
PY
  ;;
6)
  for i in 1 2; do for q in 4 8; do
    GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-cpu-baseline --no-others --no-pmc > $O/bench_q${q}_$i.json 2> $O/bench_q${q}_$i.err
  done; done
  ;;
7)
  for i in 1 2; do
    timeout 300 python bench.py --no-cpu-baseline > $O/bench_pmc_$i.json 2> $O/bench_pmc_$i.err
    timeout 300 python bench.py --no-cpu-baseline --no-pmc > $O/bench_nopmc_$i.json 2> $O/bench_nopmc_$i.err
  done
  ;;
8)
  timeout 300 python -m pytest tests -m gpu -q --maxfail=5 --timeout 300 -p no:cacheprovider -k "conv or cnn or pool" > $O/pytest_conv.log 2>&1; echo "pytest rc=$?" > $O/summary.log; tail -n 4 $O/pytest_conv.log
  for pf in 0 1; do CTCN_OPT_CONV_PREFETCH=$pf timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | grep '"mfma": 1' | sed "s/^/prefetch=$pf /"; done | tee $O/conv_bench.txt
  for pf in 0 1; do for wl in cfg3 ref_yaml; do
    CTCN_OPT_CONV_PREFETCH=$pf timeout 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-decode --no-cpu-baseline --no-others --no-pmc > $O/${wl}_prefetch$pf.json 2> $O/${wl}_prefetch$pf.err
  done; done
  ;;
9)
  timeout 300 python -m pytest tests -m gpu -q --maxfail=5 --timeout 300 -p no:cacheprovider -k "conv or cnn or pool" > $O/pytest_conv.log 2>&1; echo "pytest rc=$?" > $O/summary.log; tail -n 3 $O/pytest_conv.log
  timeout 200 python tools/conv_phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/conv_phase_probe.txt
  for wl in cfg3 ref_yaml; do timeout 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-decode --no-cpu-baseline --no-others --no-pmc > $O/${wl}.json 2> $O/${wl}.err; done
  ;;
10)
  timeout 300 python -m pytest tests -m gpu -q --maxfail=5 --timeout 300 -p no:cacheprovider -k "conv or cnn or pool" > $O/pytest_conv.log 2>&1; echo "pytest rc=$?" > $O/summary.log; tail -n 3 $O/pytest_conv.log
  timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | grep '"mfma": 1' | tee $O/conv_bench.txt
  for wl in cfg3 ref_yaml; do timeout 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-decode --no-cpu-baseline --no-others --no-pmc > $O/${wl}.json 2> $O/${wl}.err; done
  ;;
11)
  for shape in "1200 64 512 gru" "800 32 320 lstm" "1200 32 512 gru" "1200 64 384 gru"; do set -- $shape
    for c in FETCH_SIZE WRITE_SIZE; do
      ( cd /tmp && PMC_PROBE_SET=recurrence PMC_T=$1 PMC_B=$2 PMC_H=$3 PMC_CELL=$4 timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O/p_$c -o p -- python $R/tools/pmc_probe.py > $O/p.log 2>&1 )
      echo "== T=$1 B=$2 H=$3 $4 $c"; python tools/pmc_dump.py $(find $O/p_$c -name "*.db" | head -1) rnn_; rm -rf $O/p_$c
    done
  done | tee $O/rnn_traffic.txt
  ;;
12)
  for nt in 0 1; do
    for c in FETCH_SIZE WRITE_SIZE; do
      ( cd /tmp && CTCN_OPT_RNN_RSV_NT=$nt PMC_PROBE_SET=recurrence PMC_T=1200 PMC_B=64 PMC_H=512 PMC_CELL=gru timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O/p_$c -o p -- python $R/tools/pmc_probe.py > $O/p.log 2>&1 )
      echo "== rsv_nt=$nt $c"; python tools/pmc_dump.py $(find $O/p_$c -name "*.db" | head -1) rnn_bwd; rm -rf $O/p_$c
    done
    CTCN_OPT_RNN_RSV_NT=$nt timeout 300 python bench.py --workload cfg4 --steps 15 --warmup 3 --no-decode --no-cpu-baseline --no-others --no-pmc > $O/cfg4_nt$nt.json 2> $O/cfg4_nt$nt.err
  done | tee $O/rsv_nt.txt
  ;;
13)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" > $O/summary.log; tail -n 2 $O/smoke.log
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/summary.log; tail -n 6 $O/pytest_gpu.log
  timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/summary.log
  ;;
14)
  timeout 900 python -m pytest tests -m gpu -q --maxfail=10 --timeout 600 -p no:cacheprovider -k "beam or decoder or decode" > $O/pytest_beam.log 2>&1; echo "pytest rc=$?" > $O/summary.log; tail -n 12 $O/pytest_beam.log
  cat > /tmp/wide_time.py <<'PY'
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from ctc_pytorch_amd import ops
from ctc_pytorch_amd.utils.NgramLM import LanguageModel
from oracle import synth
dev = torch.device("cuda", 0)
V, T, B = 62, 800, 128
i2c = synth.int2char(V)
tab = torch.as_tensor(LanguageModel("tests/golden/lm_phone_bg.arpa").table([i2c[i] for i in range(V)]), dtype=torch.float64).to(dev)
for regime in ("peaky", "flat"):
    x = torch.from_numpy(synth.make_logprobs(seed=7, T=T, B=B, V=V, regime=regime)).to(dev)
    lens = torch.as_tensor(np.random.RandomState(2).randint(400, 801, size=B), dtype=torch.int32).to(dev)
    for W in (20, 60, 61, 128, 200, 256):
        for fast in ((1, 0) if W <= 60 else (1,)):
            ops.set_option("beam_fast", fast)
            ops.beam_decode_device(x, lens, tab, 0.01, W); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = ops.beam_decode_device(x, lens, tab, 0.01, W); e1.record(); torch.cuda.synchronize()
            print("cfg5 batch (128 x 800 x 62) %-5s W=%3d %s kernel: %9.2f ms  status ok %s" % (regime, W, "fast   " if (fast and W <= 60) else "generic", e0.elapsed_time(e1), bool((out[3] == 0).all())), flush=True)
ops.set_option("beam_fast", 1)
PY
  { echo "## this build"; python /tmp/wide_time.py 2>&1 | grep -v amdgpu.ids
    if [ -f tools/libctcn_r5base.so ]; then cp ctc_pytorch_amd/libctcn.so /tmp/libctcn_new.so; cp tools/libctcn_r5base.so ctc_pytorch_amd/libctcn.so
      echo "## the build before round 5's generic kernel (W block-wide arg-max rounds per frame, 256 threads)"; timeout 600 python /tmp/wide_time.py 2>&1 | grep -v amdgpu.ids; cp /tmp/libctcn_new.so ctc_pytorch_amd/libctcn.so; fi
    echo "## cycles per phase and frame of the generic kernel (tools/mb_beam.py generic <regime> <W>)"
    for r in peaky flat; do for W in 128 200 256; do timeout 200 python tools/mb_beam.py generic $r $W 2>&1 | grep -v amdgpu.ids; done; done; } | tee $O/wide_beam_time.txt
  ;;
15)
  timeout 600 python -m pytest tests -m gpu -q --maxfail=10 --timeout 600 -p no:cacheprovider -k "bn_relu_dropout or batchnorm or conv or cnn or model_ or dropout or shipped or end_to_end" > $O/pytest_sub.log 2>&1; echo "pytest rc=$?" > $O/summary.log; tail -n 6 $O/pytest_sub.log
  for f in 0 1; do for wl in cfg3 ref_yaml; do
    CTCN_FUSE_BN_DROPOUT=$f timeout 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-decode --no-cpu-baseline --no-others --no-pmc > $O/${wl}_fuse$f.json 2> $O/${wl}_fuse$f.err
  done; done
  ;;
17)
  # the round's last evidence session: full parity suite + smoke at HEAD, the driver's command (new: bf16_gemm_mode, comm.rccl_info under forced
  # collectives), the bf16 mode's tile probe, kernel stats / all-stream timeline of cfg4 in the bf16 mode
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/summary.log
  tail -n 6 $O/pytest_gpu.log
  timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.log
  timeout 900 python bench.py --steps 20 --warmup 3 > $O/r05_bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/summary.log
  { echo "# tools/gemm_single_probe.py: option gemm_bf16_single on the three 256-row GEMM tiles (us per product incl. its split passes; errors in units of mean |C|)"
    timeout 300 python tools/gemm_single_probe.py 2>&1 | grep -v amdgpu.ids; } > $O/r05_gemm_single_probe.txt
  for wl in cfg2 cfg3 cfg4 ref_yaml; do
    CTCN_OPT_GEMM_BF16_SINGLE=1 timeout 400 python bench.py --workload $wl --steps 20 --warmup 3 --no-decode --no-cpu-baseline --no-others --no-pmc > $O/r05_bench_${wl}_bf16_single.json 2> $O/bench_${wl}_bf16.err
  done
  ( cd /tmp && CTCN_OPT_GEMM_BF16_SINGLE=1 timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_cfg4 -o p -- python $R/bench.py --workload cfg4 --steps 8 --warmup 2 --no-decode --no-cpu-baseline --no-others --no-pmc > $O/cfg4_bf16_under_rocprof.json 2> $O/prof_cfg4.log )
  db=$(find $O/prof_cfg4 -name "*.db" | head -1)
  [ -n "$db" ] && python tools/prof_timeline.py $db -1 all > $O/r05_cfg4_bf16_single_step_timeline.txt 2>&1 && python tools/prof_stats.py $db > $O/r05_cfg4_bf16_single_kernel_stats.txt 2>&1
  rm -rf $O/prof_cfg4
  cat $O/summary.log
  python - $O <<'PY'
import json, sys, os
O = sys.argv[1]
d = json.loads(open(os.path.join(O, "r05_bench_default.json")).read().strip().splitlines()[-1])
print("default: %.3f ms/step  decode %s  others %s  bf16 mode %s  traffic %s" % (
    d["ms_per_step"], {k: round(v["value"]) for k, v in d["decode"]["regimes"].items()},
    {k: round(v.get("ms_per_step", -1), 2) for k, v in d["other_workloads"].items()},
    {k: round(v.get("ms_per_step", -1), 2) for k, v in d["bf16_gemm_mode"].items() if isinstance(v, dict)}, d["roofline"].get("traffic")))
for wl in ("cfg2", "cfg3", "cfg4", "ref_yaml"):
    e = json.loads(open(os.path.join(O, "r05_bench_%s_bf16_single.json" % wl)).read().strip().splitlines()[-1])
    print(wl, "bf16 single: %.3f ms/step" % e["ms_per_step"])
PY
  ;;
19)
  # after the XCD placement change (option xcd_interleave = 1 by default): full parity suite, smoke, soak, the driver's command
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/summary.log
  tail -n 4 $O/pytest_gpu.log
  timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.log
  timeout 900 python tools/soak.py --cfg2 1500 --cfg4 300 --ref-yaml 1500 --cfg1 1500 --cfg3 1000 --decode 300 --out $O/r05_soak.json > $O/soak.out 2> $O/soak.err; echo "soak rc=$?" >> $O/summary.log
  timeout 900 python bench.py --steps 20 --warmup 3 > $O/r05_bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/summary.log
  cat $O/summary.log; tail -n 3 $O/soak.out
  ;;
20)
  # the round's closing evidence at HEAD (bn_rows4 in, xcd_interleave at its default 0): the driver's command and the soak
  timeout 900 python bench.py --steps 20 --warmup 3 > $O/r05_bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" > $O/summary.log
  timeout 900 python tools/soak.py --cfg2 1500 --cfg4 300 --ref-yaml 1500 --cfg1 1500 --cfg3 1000 --decode 300 --out $O/r05_soak.json > $O/soak.out 2> $O/soak.err; echo "soak rc=$?" >> $O/summary.log
  cat $O/summary.log
  ;;
22)
  # XCD order 1 as the default where XCDs stay idle (cfg4's launch untouched): full parity suite, smoke, the driver's command
  timeout 1200 python -m pytest tests -m gpu -q --maxfail=8 --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/summary.log
  tail -n 3 $O/pytest_gpu.log
  timeout 200 python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.log
  timeout 600 python bench.py --steps 20 --warmup 3 > $O/r05_bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/summary.log
  cat $O/summary.log
  ;;
esac
ls -la $O; cat $O/summary.log
python - "$O" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        if "regimes" in d:
            print(os.path.basename(f), {k: (round(v["value"]), round(v["kernel_us_per_batch"]), v["strings_match_oracle"]) for k, v in d["regimes"].items()})
            continue
        print(os.path.basename(f), "ms/step %.3f" % d["ms_per_step"], "fwd %.3f bwd %.3f" % (d["recurrence"]["fwd_us_per_timestep"], d["recurrence"]["bwd_us_per_timestep"]),
              "decode", (d.get("decode") or {}).get("value"), (d.get("decode") or {}).get("value_flat"),
              {k: (round(v.get("ms_per_step", -1), 3) if isinstance(v, dict) else v) for k, v in (d.get("other_workloads") or {}).items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
