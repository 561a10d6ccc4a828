"""Experiments inside the process that has just run the full parity suite (tests/conftest.py: CTCN_AFTER_SUITE), where the cfg4 divergence of round 6
lives (8-10 % of traced 12-step runs there, 0 of ~200 in fresh processes).  Phases, each `n` traced cfg4 runs compared with the majority trace of ALL runs:
  order0 / order1   option rnn_proj_order 0 | 1 (one product over ascending time | [T/2, T) then [0, T/2)); bit-identical by construction
  nogc              the interpreter's cyclic collector off during the runs (a long-lived process pauses for tens of ms per generation-2 pass)
  emptied           gc.collect() + torch.cuda.empty_cache() first: a fresh allocator pool, new addresses
CTCN_AFTER_SUITE_PHASES selects them (default "order1,nogc,emptied,order1"), CTCN_AFTER_SUITE_N the runs per phase.  One JSON line to CTCN_AFTER_SUITE_OUT."""
import collections, gc, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import squat_stress
from ctc_pytorch_amd import ops

n = int(os.environ.get("CTCN_AFTER_SUITE_N", "16"))
phases = os.environ.get("CTCN_AFTER_SUITE_PHASES", "order1,nogc,emptied,order1").split(",")
dev = torch.device("cuda", 0)
found = ops.get_option("rnn_proj_order")
from ctc_pytorch_amd import _lib
print("\n[after_suite_ab] process state: side live %r pending %r deferred %r events %r streams %r; workspaces %r; allocator reserved %.1f GB" % (
    dict(ops._side["live"]), list(ops._side["pending"]), list(ops._side["deferred"]), list(ops._side["events"]), list(ops._side["streams"]),
    {k: (hex(v.data_ptr()), v.numel() >> 20) for k, v in _lib._WS.items()}, torch.cuda.memory_reserved() / 2 ** 30), flush=True)
runs = []
try:
    for pi, ph in enumerate(phases):
        ops.set_option("rnn_proj_order", 0 if ph == "order0" else 1)
        if ph == "nogc":
            gc.collect()
            gc.disable()
        if ph == "side_reset":
            ops._side["live"].clear(); ops._side["pending"].clear(); ops._side["deferred"].clear()
        if ph == "emptied":
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        try:
            for i in range(n):
                r = squat_stress.run("cfg4", 12, squat=(i % 2 == 1), seed=i + 1, dev=dev, trace=True)
                runs.append(("%d:%s" % (pi, ph), json.dumps(r["trace"]), r["trace"], r["addresses"][0]["gates"] if r.get("addresses") else 0, r.get("layer0_per_timestep")))
        finally:
            if ph == "nogc":
                gc.enable()
finally:
    ops.set_option("rnn_proj_order", found)
ref_j = collections.Counter(r_[1] for r_ in runs).most_common(1)[0][0]
import base64
import numpy as np
ref_td = next(r_[4] for r_ in runs if r_[1] == ref_j)


def td_pattern(td, step):
    """layer 0 per (timestep, direction): which timesteps of which direction differ from the reference at `step`"""
    pat = {}
    for kk in ("gates", "y"):
        a_ = np.frombuffer(base64.b64decode(td[kk]), dtype=np.int64).reshape(12, -1, 2)
        b_ = np.frombuffer(base64.b64decode(ref_td[kk]), dtype=np.int64).reshape(12, -1, 2)
        for dr in (0, 1):
            ts = np.nonzero(a_[step, :, dr] != b_[step, :, dr])[0]
            if len(ts):
                mag = np.abs(a_[step, ts, dr] - b_[step, ts, dr]).astype(float)
                pat["%s d%d" % (kk, dr)] = dict(count=int(len(ts)), first_t=int(ts.min()), last_t=int(ts.max()), largest_at_t=int(ts[int(np.argmax(mag))]))
    return pat

ref = json.loads(ref_j)
out = {"runs_per_phase": n, "phases": phases, "deviating": collections.OrderedDict(("%d:%s" % (pi, ph), []) for pi, ph in enumerate(phases)), "gates_addresses": {}}
for k, (ph, j, tr, addr, td) in enumerate(runs):
    out["gates_addresses"].setdefault(ph, sorted(set()))
    if j != ref_j:
        step, first = 0, None
        for (na, va), (nb, vb) in zip(tr, ref):
            if na == "end-of-step":
                step += 1
            elif va != vb:
                first = (step, na)
                break
        out["deviating"][ph].append({"run": k, "first_difference": first, "gates_at": hex(addr), "layer0_pattern": td_pattern(td, first[0]) if (first and td) else None})
out["gates_addresses"] = {ph: sorted({hex(r_[3]) for r_ in runs if r_[0] == ph}) for ph in out["deviating"]}
out["summary"] = "; ".join("%s: %d of %d deviate" % (ph, len(v), n) for ph, v in out["deviating"].items())
print("\n[after_suite_ab] " + json.dumps(out), flush=True)
path = os.environ.get("CTCN_AFTER_SUITE_OUT", os.path.join(ROOT, "gpurun_out", "after_suite_ab.json"))
os.makedirs(os.path.dirname(path), exist_ok=True)
with open(path, "a") as f:
    f.write(json.dumps(out) + "\n")
