"""Round-3 soak: long runs of the training step and of the beam search at precision 1 with every fast path on
(tagged-gather forward recurrence, chunk-counter projection pipeline, in-recurrence Philox dropout, scatter / item-gather backward
recurrence, weight-gradient side stream, gradient-slice hook), `check_health` EVERY step, loss finite, and the whole loss trajectory
BIT-REPRODUCIBLE across two runs from the same seed (everything in the step rests on L2-visibility timing; nothing may depend on it).

    python tools/soak.py [--cfg2 5000] [--cfg4 2000] [--ref-yaml 2000] [--cfg1 0] [--cfg3 0] [--decode 1000] [--out profiles/r04_soak.json]
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def train_run(workload, steps, dev):
    import bench
    from ctc_pytorch_amd import nn, ops, parallel
    from ctc_pytorch_amd.optim import FlatAdam
    from ctc_pytorch_amd.testing import synth
    c = dict(bench.WORKLOADS[workload])
    for key in ("T", "B", "H", "L"):                # the same shape overrides as bench.py (CTCN_BENCH_T / B / H / L)
        if os.environ.get("CTCN_BENCH_" + key):
            c[key] = int(os.environ["CTCN_BENCH_" + key])
    ops.set_precision(1)
    ops._fallback_shapes.clear()            # (batch chunks are learnt from a shape's first call: both runs of a workload must learn at the same step)
    parallel.enable_overlap(True)
    torch.manual_seed(1)
    ops._drop_counter[0] = 0
    model = bench.build(c, dev, drop_out=c.get("drop", 0.1)).train()
    opt = FlatAdam(model, lr=1e-3, weight_decay=5e-4)
    batch = synth.make_batch(seed=1, B=c["B"], T=c["T"], F=c.get("F", 40), V=c["V"], lab_lo=c["lab"][0], lab_hi=c["lab"][1], full_length=True)
    x = torch.from_numpy(batch["x"]).to(dev)
    tg, tl = torch.from_numpy(batch["targets"]).to(dev), torch.from_numpy(batch["tgt_len"]).to(dev)
    loss_fn = nn.CTCLoss(reduction="sum")
    status = ops._lib.status_word(dev)
    in_len = None
    losses = torch.zeros(steps, dtype=torch.float32, device=dev)
    health = torch.zeros(steps, dtype=torch.int32, device=dev)
    ring = [torch.cuda.Event() for _ in range(3)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = model(x)
        if in_len is None:
            in_len = torch.full((c["B"],), out.size(0), dtype=torch.int64, device=dev)
        loss = loss_fn(out, tg, in_len, tl) / c["B"]
        opt.zero_grad()
        loss.backward()
        ops.join_side_stream()
        parallel.allreduce_grads(opt.grad)
        opt.step()
        losses[i] = loss.detach()
        health[i] = status[0]                      # the sticky hand-off status word, sampled behind every step (stream order)
        ring[i % 3].record()
        if i >= 2:
            ring[(i - 2) % 3].synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.check_health()
    l = losses.cpu().numpy()
    h = health.cpu().numpy()
    return dict(steps=steps, ms_per_step=dt / steps * 1e3, finite=bool(np.isfinite(l).all()), health_nonzero_steps=int((h != 0).sum()),
                first_loss=float(l[0]), last_loss=float(l[-1]), sha256=hashlib.sha256(l.tobytes()).hexdigest(), kernels=list(ops.rnn_last_kernels()))


def decode_run(batches, dev):
    from ctc_pytorch_amd import ops
    from ctc_pytorch_amd.utils.NgramLM import LanguageModel
    from ctc_pytorch_amd.testing import synth
    V, T, B, W = 62, 800, 128, 20
    i2c = synth.int2char(V)
    tab = LanguageModel(os.path.join(ROOT, "tests", "golden", "lm_phone_bg.arpa")).table([i2c[i] for i in range(V)])
    tab_dev = torch.as_tensor(tab, dtype=torch.float64).to(dev)
    NS = 3                                              # searches in flight, as steps/test_ctc.decode_and_score runs them
    streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
    h = hashlib.sha256()
    bad = 0
    t0 = time.perf_counter()
    xs = []
    for k in range(4):                                  # four different batches, cycled (peaky / flat x two seeds)
        lp = synth.make_logprobs(seed=7 + k, T=T, B=B, V=V, regime="peaky" if k % 2 == 0 else "flat")
        lens = torch.as_tensor(np.random.RandomState(2 + k).randint(400, 801, size=B), dtype=torch.int32).to(dev)
        xs.append((torch.from_numpy(lp).to(dev), lens))
    pend = []
    first = {}
    for i in range(batches):
        x, lens = xs[i % 4]
        with torch.cuda.stream(streams[i % NS]):
            pend.append((i % 4, ops.beam_decode_async(x, lens, tab_dev, 0.1, W)))
        if len(pend) == 2 * NS:
            k, res = pend.pop(0)
            ids, score, st = res.result()
            key = (tuple(map(tuple, ids)), score.tobytes())
            if k in first:
                bad += first[k] != key
            else:
                first[k] = key
                h.update(repr(ids).encode()); h.update(score.tobytes())
            bad += int((st != 0).any())
    for k, res in pend:
        ids, score, st = res.result()
        bad += first.get(k, (tuple(map(tuple, ids)), score.tobytes())) != (tuple(map(tuple, ids)), score.tobytes())
    dt = time.perf_counter() - t0
    return dict(batches=batches, utterances=batches * B, utt_per_s=batches * B / dt, mismatching_batches=int(bad), sha256=h.hexdigest())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg2", type=int, default=5000)
    ap.add_argument("--cfg4", type=int, default=2000)
    ap.add_argument("--ref-yaml", type=int, default=2000)
    ap.add_argument("--decode", type=int, default=1000)
    ap.add_argument("--cfg1", type=int, default=0)
    ap.add_argument("--cfg3", type=int, default=0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    res = {"precision": 1, "note": "two runs per workload from the same seed; `bit_reproducible` compares the sha256 of the float32 loss trajectories"}
    ok = True
    for name, steps in (("cfg2", a.cfg2), ("cfg4", a.cfg4), ("ref_yaml", a.ref_yaml), ("cfg1", a.cfg1), ("cfg3", a.cfg3)):
        if steps <= 0:
            continue
        r1 = train_run(name, steps, dev)
        r2 = train_run(name, steps, dev)
        r1["bit_reproducible"] = r1["sha256"] == r2["sha256"]
        r1["ms_per_step_second_run"] = r2["ms_per_step"]
        res[name] = r1
        ok = ok and r1["finite"] and r1["bit_reproducible"] and r1["health_nonzero_steps"] == 0 and r2["health_nonzero_steps"] == 0
        print(name, json.dumps(r1), file=sys.stderr, flush=True)
    if a.decode > 0:
        d = decode_run(a.decode, dev)
        res["decode"] = d
        ok = ok and d["mismatching_batches"] == 0
        print("decode", json.dumps(d), file=sys.stderr, flush=True)
    res["ok"] = bool(ok)
    print(json.dumps(res))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)
    sys.exit(0 if ok else 1)
