# A/B of the backward hand-off: CTCN_HANDOFF_TAGS=0 (drain + flag per block) vs 1 (tagged blocks polled directly)
for m in ${MODES:-0 1}; do
  export CTCN_HANDOFF_TAGS=$m
  echo "== tags $m"
  timeout 300 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "rnn or side or train" 2>&1 | tail -1
  for i in 1 2; do timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r[\"ms_per_step\"], r[\"recurrence\"][\"kernel_bwd_us\"], r[\"recurrence\"][\"kernel_fwd_us\"], r[\"final_loss\"])"; done
done
