# the three overlap switches (default | projection pipeline off | side stream off) over a list of shape overrides of one bench workload, one box session:
#   tools/ab_shapes.sh cfg2 "CTCN_BENCH_H=256" "CTCN_BENCH_H=512 CTCN_BENCH_B=32" ...
cd "$GRAFT_REPO_ROOT"
wl="$1"; shift
run() { env "$@" python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-decode 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'])"; }
for cfg in "$@"; do
  a=$(run $cfg); b=$(run $cfg CTCN_FWD_OVERLAP=0); c=$(run $cfg CTCN_SIDE_STREAM=0)
  echo "[$wl $cfg] default $a | pipeline off $b | side stream off $c   (ms per step)"
done
