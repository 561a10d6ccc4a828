"""Co-residency stress of the persistent recurrences against FOREIGN resident kernels (VERDICT r2 #1b).

An RCCL all-reduce that waits for a slow peer is, to the rest of the chip, a set of workgroups parked on some CUs of every XCD for an
unknown time.  The persistent recurrences need their whole grid co-resident on exact XCDs, the pipelined input projection needs its
side-stream GEMM workgroups to START on the recurrence's XCDs in order to leave them, and the weight-gradient side stream shares the
idle XCDs.  One GPU cannot run two RCCL ranks, so the parked kernel is simulated: `ctcn_diag_squat` launches k workgroups per XCD that
hold their slots for a given time, on a third stream, at random points of a run of training steps (precision 1, side stream, forward
overlap and the gradient-slice hook all on).  Requirements: no hand-off timeout (check_health), no hang, and the loss trajectory is
BIT-IDENTICAL to the undisturbed run (nothing in the step may depend on timing).

    python tools/squat_stress.py [--workload cfg2] [--steps 200] [--seed 1] [--max-wgs 12] [--max-us 4000] [--out profiles/r03_squat_stress.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run(workload="cfg2", steps=60, squat=True, seed=1, max_wgs=12, max_us=4000, squats_per_step=2.0, dev=None, log=None, trace=False, rows_capture=None):
    """`steps` training steps of `workload` (bench.py's model, batch and step); with `squat`, squatter launches at random host-side points
    of every step (before the forward pass, between forward and backward, inside the backward pass through a gradient hook on the
    output).  Returns dict(losses, ms_per_step, squats, squat_wg_us)."""
    import bench
    from ctc_pytorch_amd import nn, ops, parallel
    from ctc_pytorch_amd.optim import FlatAdam
    from ctc_pytorch_amd.testing import synth
    dev = dev or torch.device("cuda", 0)
    c = bench.WORKLOADS[workload]
    found = ops.state_snapshot()                      # (round 6) this function leaves the process-wide state as it found it, and says what it ran in
    if os.environ.get("CTCN_WS_FILL"):                # (round 6 experiment: the library workspaces hold garbage / zeros when the run starts)
        from ctc_pytorch_amd import _lib
        _lib.workspace(dev)
        _lib.workspace(dev, tag="side")
        torch.cuda.synchronize()
        for buf in _lib._WS.values():
            f32 = buf.view(torch.float32)
            if os.environ["CTCN_WS_FILL"] == "zero":
                f32.zero_()
            else:
                f32.copy_(torch.randn(f32.shape, device=f32.device) * 3.0)
        torch.cuda.synchronize()
    ops.set_precision(1)
    parallel.enable_overlap(True)
    torch.manual_seed(1)
    ops._drop_counter[0] = 0                          # the dropout stream restarts: two runs of this function draw the same masks
    model = bench.build(c, dev, drop_out=0.1).train()
    opt = FlatAdam(model, lr=1e-3, weight_decay=5e-4)
    batch = synth.make_batch(seed=1, B=c["B"], T=c["T"], F=40, V=c["V"], lab_lo=c["lab"][0], lab_hi=c["lab"][1], full_length=True)
    x = torch.from_numpy(batch["x"]).to(dev)
    tg, tl = torch.from_numpy(batch["targets"]).to(dev), torch.from_numpy(batch["tgt_len"]).to(dev)
    loss_fn = nn.CTCLoss(reduction="sum")
    rs = np.random.RandomState(seed)
    third = torch.cuda.Stream(device=dev)
    stats = dict(n=0, wg_us=0.0)

    def maybe_squat(p):
        if not squat or rs.rand() >= p:
            return
        k = int(rs.choice([1, 2, 4, 8, max_wgs]))
        us = int(rs.randint(50, max_us + 1))
        thr = int(rs.choice([256, 512]))
        lds = int(rs.choice([0, 16 << 10, 64 << 10]))
        ops.diag_squat(k, us, threads=thr, lds_bytes=lds, stream=third)
        stats["n"] += 1
        stats["wg_us"] += k * us

    # trace (round 6, tools/traj_bisect.py): per step, a 64-bit sum of the bit patterns of every recurrent layer's output, of the head's
    # output, of the flat gradient before and of the flat parameters after the optimiser step -- computed ON the device, read after the last
    # step (no synchronisation inside the run: the launches keep the timing of the untraced loop)
    tr_names, tr_vals = [], []

    def note(name, t):
        tr_names.append(name)
        tr_vals.append(t.detach().reshape(-1).view(torch.int32).sum(dtype=torch.int64))

    addrs, per_td = [], {"gates": [], "y": []}
    if trace:
        layer_no = [0]

        def rnn_note(tag, addresses, tensors):          # per recurrent-layer call: buffer addresses + checksums of what it read and wrote
            l = layer_no[0] % c["L"]
            layer_no[0] += 1
            addrs.append(dict(addresses, layer=l))
            for k in ("x", "gates", "aux", "y", "y_drop"):
                if tensors.get(k) is not None:
                    note("layer %d %s" % (l, k), tensors[k])
            if l == 0 and rows_capture is not None:      # (tools/first_rows_probe.py) the bottom layer's reserve and output at chosen timesteps, kept per step
                ridx = torch.tensor(rows_capture[0], device=dev)
                xr = tensors["x"][ridx]                                    # (rows, B, I); the pre-activations the projection should have produced (f32 matmul, ~1e-6)
                pre = torch.stack([xr @ w_.t() for w_ in tensors["w_ih"]], dim=2) if tensors.get("w_ih") and tensors["w_ih"][1] is not None else None
                rows_capture[1].append(dict(gates=tensors["gates"][ridx].clone(), y=tensors["y"][ridx].clone(), pre=pre, aux=tensors["aux"][ridx].clone() if tensors.get("aux") is not None else None))
            if l == 0:                                   # the bottom layer per (timestep, direction): where along the sequence does a run leave the others?
                for k in ("gates", "y"):
                    t_ = tensors[k]
                    per_td[k].append(t_.detach().reshape(t_.shape[0], t_.shape[1], 2, -1).view(torch.int32).sum(dim=(1, 3), dtype=torch.int64))

        ops._debug_note[0] = rnn_note
        note("model-input", x)
        for name, mod in list(model.rnns.named_children()) + [("fc", model.fc)]:
            mod.register_forward_hook(lambda m, i, o, name=name: note("fwd %s" % name, o))
            mod.register_full_backward_pre_hook(lambda m, g, name=name: note("grad-of-output %s" % name, g[0]))
    p = squats_per_step / 4.0
    in_len = None
    losses = []
    ring = [torch.cuda.Event() for _ in range(3)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lag = float(os.environ.get("CTCN_STEP_LAG", "0"))          # (round 6 experiment: a host that falls behind -- the GPU drains between steps)
    lag_rs = np.random.RandomState(seed + 1000)
    for i in range(steps):
        if lag > 0 and lag_rs.rand() < 0.4:
            time.sleep(lag)
        maybe_squat(p)
        if trace:
            note("model-input", x)
        out = model(x)
        if in_len is None:
            in_len = torch.full((c["B"],), out.size(0), dtype=torch.int64, device=dev)
        maybe_squat(p)
        loss = loss_fn(out, tg, in_len, tl) / c["B"]
        opt.zero_grad()
        if squat:
            out.register_hook(lambda g: (maybe_squat(p), g)[1])       # fires at the start of the backward pass of the model body
        loss.backward()
        maybe_squat(p)
        ops.join_side_stream()
        parallel.allreduce_grads(opt.grad)
        if trace:
            note("loss", loss)
            note("flat-gradient", opt.grad)
        opt.step()
        if trace:
            note("flat-parameters", opt.flat)
            tr_names.append("end-of-step")
            tr_vals.append(torch.zeros((), dtype=torch.int64, device=dev))
        losses.append(loss.detach())
        ring[i % 3].record()
        if i >= 2:
            ring[(i - 2) % 3].synchronize()
        if log and (i + 1) % 50 == 0:
            log("step %d" % (i + 1))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.check_health()
    res = dict(losses=[float(l) for l in losses], ms_per_step=dt / steps * 1e3, squats=stats["n"], squat_wg_us=stats["wg_us"],
               kernels=ops.rnn_last_kernels())
    if trace:
        ops._debug_note[0] = None
        res["trace"] = list(zip(tr_names, [int(v) for v in torch.stack(tr_vals).cpu().tolist()]))
        res["addresses"] = addrs
        import base64
        res["layer0_per_timestep"] = {k: base64.b64encode(torch.stack(v).cpu().numpy().tobytes()).decode() for k, v in per_td.items() if v}
    ran_in = ops.state_snapshot()
    ops.restore_state(dict(found, fallback_shapes=ran_in["fallback_shapes"], drop_counter=ran_in["drop_counter"]))
    if os.environ.get("CTCN_TRAJ_LOG"):               # one line per run: the trajectory next to the state it was computed in
        with open(os.environ["CTCN_TRAJ_LOG"], "a") as f:
            f.write(json.dumps(dict(workload=workload, steps=steps, squat=bool(squat), seed=seed, losses=res["losses"], kernels=res["kernels"], trace=res.get("trace"), addresses=res.get("addresses"), layer0_per_timestep=res.get("layer0_per_timestep"),
                                    state_found=found, state_ran_in=ran_in)) + "\n")
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-wgs", type=int, default=12)
    ap.add_argument("--max-us", type=int, default=4000)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    base = run(a.workload, a.steps, squat=False)
    hit = run(a.workload, a.steps, squat=True, seed=a.seed, max_wgs=a.max_wgs, max_us=a.max_us, log=lambda s: print(s, file=sys.stderr, flush=True))
    same = base["losses"] == hit["losses"]
    res = dict(workload=a.workload, steps=a.steps, kernels=hit["kernels"], ms_per_step_undisturbed=base["ms_per_step"], ms_per_step_with_squatters=hit["ms_per_step"],
               squatter_launches=hit["squats"], squatter_workgroup_us_per_xcd=hit["squat_wg_us"], loss_trajectory_bit_identical=same,
               finite=bool(np.isfinite(hit["losses"]).all()), final_loss=hit["losses"][-1], health="ok")
    print(json.dumps(res))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)
    sys.exit(0 if same and res["finite"] else 1)
