"""The projection-order A/B inside the process that has just run the full parity suite (tests/conftest.py: CTCN_AFTER_SUITE).  The cfg4 divergence of
round 6 showed in 3 of 10 full suites and in none of ~200 fresh-process runs, so the order fix is A/B-ed where the event lives: `n` traced 12-step cfg4
runs per order, interleaved, every trace compared with the majority; option rnn_proj_order 0 = one product over ascending time (rounds 1-5),
1 = [T/2, T) then [0, T/2) (the fix).  Both orders are bit-identical by construction, so one reference trace serves both.
Writes one JSON line to CTCN_AFTER_SUITE_OUT (default gpurun_out/after_suite_ab.json)."""
import collections, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import squat_stress
from ctc_pytorch_amd import ops

n = int(os.environ.get("CTCN_AFTER_SUITE_N", "25"))
dev = torch.device("cuda", 0)
found = ops.get_option("rnn_proj_order")
runs = []
try:
    for i in range(n):
        for order in (0, 1):
            ops.set_option("rnn_proj_order", order)
            r = squat_stress.run("cfg4", 12, squat=(i % 2 == 1), seed=i + 1, dev=dev, trace=True)
            runs.append((order, json.dumps(r["trace"]), r["trace"]))
finally:
    ops.set_option("rnn_proj_order", found)
ref_j = collections.Counter(j for _, j, _ in runs).most_common(1)[0][0]
ref = json.loads(ref_j)
out = {"runs_per_order": n, "deviating": {0: [], 1: []}}
for k, (order, j, tr) in enumerate(runs):
    if j != ref_j:
        step, first = 0, None
        for (na, va), (nb, vb) in zip(tr, ref):
            if na == "end-of-step":
                step += 1
            elif va != vb:
                first = (step, na)
                break
        out["deviating"][order].append({"run": k, "first_difference": first})
out["summary"] = "old order (0): %d of %d runs deviate; new order (1): %d of %d" % (len(out["deviating"][0]), n, len(out["deviating"][1]), n)
print("\n[after_suite_ab] " + json.dumps(out), flush=True)
path = os.environ.get("CTCN_AFTER_SUITE_OUT", os.path.join(ROOT, "gpurun_out", "after_suite_ab.json"))
os.makedirs(os.path.dirname(path), exist_ok=True)
with open(path, "a") as f:
    f.write(json.dumps(out) + "\n")
