#!/usr/bin/env python3
"""bench.py -- training-step throughput of the CTC hot path on N x MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload cfg2|cfg3|cfg4|cfg1] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the hot path over one synthetic minibatch that is already resident in HBM:
forward (BN -> BiLSTM x4 -> BN -> Linear -> log_softmax) -> CTC loss (sum)/B -> backward -> one RCCL all-reduce of
the flat gradient -> fused Adam.  Default workload = BASELINE.json configs[1] ("cfg2": 4x320 BiLSTM + DNN, B=32
per GPU, T=800, F=40, V=62, dropout 0.1, batch_norm, Adam lr 1e-3 wd 5e-4), weak scaling (fixed per-GPU batch).
Rank 0 prints ONE JSON line; `value` = acoustic frames/s of the whole job.  `--mode decode` times cfg5 instead
(beam W=20 + bigram LM over 128 utterances x 800 frames; utterances/s).
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {   # per-GPU shapes
    "cfg1": dict(B=8, T=300, V=62, H=128, L=2, rnn="LSTM", cnn=False, lab=(10, 35)),
    "cfg2": dict(B=32, T=800, V=62, H=320, L=4, rnn="LSTM", cnn=False, lab=(30, 60)),
    "cfg3": dict(B=32, T=800, V=62, H=320, L=4, rnn="LSTM", cnn=True, lab=(30, 60)),
    "cfg4": dict(B=64, T=1200, V=200, H=512, L=5, rnn="GRU", cnn=False, lab=(60, 100)),
    # the reference's SHIPPED configuration (timit/conf/ctc_config.yaml:11-40): 81-d fbank spliced with right_ctx 2 -> 243-d, n_skip_frame 2
    # (400 kept frames = an 8 s utterance), 2-layer CNN -> 1 952-wide RNN input, 4 x 384 BiLSTM, batch 8, drop_out 0.2, 39 phones + blank + UNK
    "ref_yaml": dict(B=8, T=400, V=41, H=384, L=4, rnn="LSTM", cnn=True, lab=(20, 40), F=243, drop=0.2),
}
CNN_LAYERS = [[(1, 32), (3, 3), (1, 2), (1, 1), None], [(32, 32), (3, 3), (2, 2), (1, 1), None]]
PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0             # HBM3E spec; ~6300 GB/s measured achievable
PREWARM = 40                      # untimed steps before the --warmup steps (start-up transient of a fresh process, see run_train)


def gemm_weights(c, feat_in):
    G = 4 if c["rnn"] == "LSTM" else 3
    H, L = c["H"], c["L"]
    w = 0
    for l in range(L):
        I = feat_in if l == 0 else 2 * H
        w += 2 * (G * H * I + G * H * H)
    return w + 2 * H * c["V"]


def algorithmic_bytes_per_step(c):
    """SURVEY.md section 8(d)'s explicit fp32 "store-once / load-once" HBM model of one training step (every tensor saved for backward is
    written once in the forward pass and read once in the backward pass; weights stay on chip): per output frame, RNN reserve (G*H gates +
    c for the LSTM + h) x 2 directions x 4 B x 2 per layer, L BatchNorm and L dropout layers x 2H x 4 B x (read + write) x (fwd + bwd),
    logits V x 4 B x 4, input F x 4 B; CNN front-end: its two activations (conv out, BN out) stored once and loaded once.  cfg2: 205 952 B per
    frame = 5.27 GB per 25 600-frame step; cfg4 ~331 KB per frame."""
    G = 4 if c["rnn"] == "LSTM" else 3
    H, L, Fd = c["H"], c["L"], c.get("F", 40)
    per_out = L * (G * H + (H if c["rnn"] == "LSTM" else 0) + H) * 2 * 4 * 2 + 2 * L * (2 * H * 16) + c["V"] * 16
    t_out = c["T"] // 2 if c["cnn"] else c["T"]
    total = per_out * t_out * c["B"] + Fd * 4 * c["T"] * c["B"]
    if c["cnn"]:
        f1 = (Fd + 2 - 3) // 2 + 1
        f2 = (f1 + 2 - 3) // 2 + 1
        total += (32 * f1 * c["T"] + 32 * f2 * (c["T"] // 2)) * 4 * 4 * c["B"]
    return int(total)


def build(c, dev, drop_out):
    from ctc_pytorch_amd import nn
    from ctc_pytorch_amd.models.model_ctc import CTC_Model
    rp = {"rnn_input_size": c.get("F", 40), "rnn_hidden_size": c["H"], "rnn_layers": c["L"], "rnn_type": getattr(nn, c["rnn"]),
          "bidirectional": True, "batch_norm": True}
    if c["cnn"]:
        cp = {"batch_norm": True, "activate_function": nn.ReLU, "layer": CNN_LAYERS}
        m = CTC_Model(add_cnn=True, cnn_param=cp, rnn_param=rp, num_class=c["V"], drop_out=drop_out)
    else:
        m = CTC_Model(rnn_param=rp, num_class=c["V"], drop_out=drop_out)
    return m.to(dev)


def cpu_baseline_train(c, batch, steps=1, sweep=(8, 16, 32, 64)):
    """The reference's torch calls re-issued on the host cores (oracle/torch_cpu.py; kind = 'port').  oneDNN's LSTM does not
    scale to every core of the GPU box's host (round 1: 24 s per step on 128 threads, 4-6x slower than 8 threads), so the
    thread count is swept and the BEST setting is the baseline."""
    import torch.nn as tnn
    from oracle import torch_cpu
    rp = {"rnn_input_size": c.get("F", 40), "rnn_hidden_size": c["H"], "rnn_layers": c["L"], "rnn_type": getattr(tnn, c["rnn"]),
          "bidirectional": True, "batch_norm": True}
    cp = {"batch_norm": True, "activate_function": tnn.ReLU, "layer": CNN_LAYERS} if c["cnn"] else None
    torch.manual_seed(1)
    m = torch_cpu.TorchCpuCTCModel(add_cnn=c["cnn"], cnn_param=cp, rnn_param=rp, num_class=c["V"], drop_out=c.get("drop", 0.1))
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-4)
    x, frac = torch.from_numpy(batch["x"]), torch.from_numpy(batch["frac"])
    tg, tl = torch.from_numpy(batch["targets"]), torch.from_numpy(batch["tgt_len"])
    ncpu = os.cpu_count() or 8
    before = torch.get_num_threads()
    tried = {}
    try:
        for nt in [n for n in sweep if n <= ncpu] or [min(8, ncpu)]:
            torch.set_num_threads(nt)
            if not tried:
                torch_cpu.train_step(m, opt, x, frac, tg, tl)             # warm-up (oneDNN primitive creation)
            t0 = time.time()
            for _ in range(steps):
                torch_cpu.train_step(m, opt, x, frac, tg, tl)
            tried[nt] = (time.time() - t0) / steps
            if sum(tried.values()) > 60.0:                                  # keep the default bench run within a few minutes
                break
    finally:
        torch.set_num_threads(before)
    best = min(tried, key=tried.get)
    dt = tried[best]
    return dict(value=c["B"] * c["T"] / dt, unit="frames/s", cores=best, kind="port",
                sample="%d train step(s) per setting of the same %dx%dx%d batch through oracle/torch_cpu.py (torch %s CPU); thread sweep "
                       "%s of %d host cores, best = %d threads" % (steps, c["B"], c["T"], c.get("F", 40), torch.__version__,
                                                                  {k: round(v, 2) for k, v in tried.items()}, ncpu, best),
                seconds_per_step=dt, seconds_per_step_by_threads=tried)


def gemm_roofline(dev, c):
    """HIP-event timing of the dominant MFMA kernel of the step: the time-parallel input projection
    X[T*B, 2H] * W_ih^T[2H, 4H] (f32 in / f32 accumulate, v_mfma_f32_32x32x2_f32)."""
    from ctc_pytorch_amd import ops
    G = 4 if c["rnn"] == "LSTM" else 3
    M, K, N = c["T"] * c["B"], 2 * c["H"], G * c["H"]
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev)
    C = torch.empty(M, N, device=dev)
    for _ in range(3):
        ops.gemm(0, 1, M, N, K, A, K, W, K, C, N)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        ops.gemm(0, 1, M, N, K, A, K, W, K, C, N)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * M * N * K
    tf = flops / (ms * 1e-3) / 1e12
    prec = ops.get_precision()
    # HBM bytes per launch from the committed rocprofv3 PMC passes (25 600 x 1 280 x 640 probe shape)
    traffic = pmc_traffic("gemm_planes_nt256pp_af32_kernel<2>") if (prec == 1 and (M, N, K) == (25600, 1280, 640)) else None
    # `frac` = algorithmic flops against the guide's dense peak of the arithmetic used; bf16x3 issues 3 bf16 MFMAs per algorithmic product,
    # so its own ceiling for ALGORITHMIC flops is 2500/3 TFLOP/s (`frac_of_bf16x3_ceiling`)
    peak = PEAK_F32_MFMA_TFLOPS if prec == 0 else 2500.0
    return dict(kernel=("gemm_f32_kernel<NT>" if prec == 0 else "gemm_planes_nt256pp_af32 (256-row ping-pong tile, A split while staged) + the operand-split pass of B,") + " %dx%dx%d" % (M, N, K), bound="mfma",
                achieved=tf, peak=peak, unit="TFLOP/s", frac=tf / peak, frac_of_bf16x3_ceiling=None if prec == 0 else tf / (peak / 3.0),
                traffic=traffic, traffic_source=None if traffic is None else "committed rocprofv3 PMC passes (profiles/%s)" % PMC_FILE,
                us_per_launch=ms * 1e3, algorithmic_flops_per_launch=flops,
                peak_note="f32 MFMA 157.3 TFLOP/s (dense)" if prec == 0 else "dense bf16 MFMA 2500 TFLOP/s; 3 bf16 MFMAs per algorithmic product (bf16x3)")


PMC_FILE = next((f for f in ("r06_pmc_hbm_traffic.json", "r05_pmc_hbm_traffic.json", "r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", f))), "r02_pmc_hbm_traffic.json")


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/rNN_pmc_hbm_traffic.json, collected
    by tools/run_profiles_r*.sh: counters cannot be read from inside bench.py).  None when the kernel is not in the table."""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
        for k, v in pm.items():
            if isinstance(v, dict) and ("[" + kernel) in k:
                return v["hbm_bytes"]
    except Exception:
        pass
    return None


def measure_recurrence_traffic(c, kernel, precision, timeout=150):
    """HBM bytes per launch of the dominant recurrence MEASURED IN THIS RUN: two child processes -- rocprofv3
    --kernel-trace --pmc FETCH_SIZE, then --pmc WRITE_SIZE, separate passes as /opt/skills/guides/MI355X_MICROARCH.md prescribes (nothing but
    the kernel trace next to the counters) -- over tools/pmc_probe.py's single-layer set at this workload's shape; FETCH_SIZE doubled (the
    guide's gfx950 correction: 128-B requests of wide reads are tallied as 64 B), both counters in KiB.  Returns (bytes, note) or (None, why)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None or os.environ.get("CTCN_BENCH_PMC", "1") == "0":
        return None, "rocprofv3 not on PATH" if exe is None else "CTCN_BENCH_PMC=0"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):            # this process is itself being profiled: no profiler inside a profiler
        return None, "bench.py is running under a profiler"
    T = c["T"] // 2 if c["cnn"] else c["T"]
    env = dict(os.environ, PMC_PROBE_SET="recurrence", PMC_T=str(T), PMC_B=str(c["B"]), PMC_H=str(c["H"]), PMC_CELL={"LSTM": "lstm", "GRU": "gru"}[c["rnn"]],
               CTCN_PRECISION=str(precision), PYTHONDONTWRITEBYTECODE="1", TMPDIR="/tmp")
    vals = {}
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                out = os.path.join(tmp, counter)
                r = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "-d", out, "-o", "p", "--", sys.executable, os.path.join(ROOT, "tools", "pmc_probe.py")],
                                   cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
                dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith(".db")]
                if r.returncode != 0 or not dbs:
                    return None, "rocprofv3 --pmc %s failed (rc %d)" % (counter, r.returncode)
                cur = sqlite3.connect(dbs[0]).cursor()
                rows = list(cur.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name", (counter,)))
                hit = [(n, a) for name, n, a in rows if (kernel + "<") in name or (kernel + "(") in name]
                if not hit:
                    return None, "kernel %s not in the %s pass" % (kernel, counter)
                vals[counter] = hit[0]
    except Exception as e:      # noqa: BLE001
        return None, repr(e)
    nbytes = int(2 * vals["FETCH_SIZE"][1] * 1024 + vals["WRITE_SIZE"][1] * 1024)
    return nbytes, ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two separate child passes over tools/pmc_probe.py, one layer of this "
                    "workload, %d launches averaged); FETCH_SIZE x 2 (gfx950 correction of MI355X_MICROARCH.md), KiB units" % vals["FETCH_SIZE"][0])


def recurrence_traffic(workload, kernel="rnn_bwd_scatter"):
    """HBM bytes per launch of the dominant recurrence from the committed PMC passes, for the two shapes tools/pmc_probe.py runs (a cfg2
    layer; a ref_yaml layer for rnn_bwd_scatter2)."""
    if workload == "cfg2" or (workload == "ref_yaml" and kernel == "rnn_bwd_scatter2"):
        return pmc_traffic(kernel + "<")
    return None


def recurrence_probe(dev, c):
    """HIP-event timing of one BiLSTM layer of the workload.  Two passes: the full layer (input projection +
    recurrence, recurrence + deferred gradient GEMMs), and -- with the library's `rnn_recurrence_only` measurement option
    -- the persistent recurrent kernels alone (one launch = T dependent timesteps of both directions)."""
    from ctc_pytorch_amd import ops
    G = 4 if c["rnn"] == "LSTM" else 3
    cell = {"LSTM": "lstm", "GRU": "gru"}[c["rnn"]]
    T, B, H = (c["T"] // 2 if c["cnn"] else c["T"]), c["B"], c["H"]          # (the CNN front-end halves the frame rate)
    x = torch.randn(T, B, 2 * H, device=dev, requires_grad=True)
    w = [(torch.randn(G * H, 2 * H, device=dev) * 0.05).requires_grad_(True), (torch.randn(G * H, H, device=dev) * 0.05).requires_grad_(True),
         (torch.randn(G * H, 2 * H, device=dev) * 0.05).requires_grad_(True), (torch.randn(G * H, H, device=dev) * 0.05).requires_grad_(True)]
    gy = torch.ones(T, B, 2 * H, device=dev)

    def timed(reps):
        # median of the passes behind a warm-up pass, the interpreter's collector off (as in the timed steps: a collection of the heap the
        # training loop left behind is tens of milliseconds of host time in the middle of an enqueue -- the device then waits for its kernels)
        fs, bs = [], []
        gc_was = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            for i in range(reps + 1):
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record()
                y = ops.rnn_layer(x, w[0], w[1], w[2], w[3], cell)
                ev[1].record()
                y.backward(gy)
                ev[2].record()
                torch.cuda.synchronize()
                if i:
                    fs.append(ev[0].elapsed_time(ev[1]))
                    bs.append(ev[1].elapsed_time(ev[2]))
        finally:
            if gc_was:
                gc.enable()
        fs.sort(); bs.sort()
        return fs[len(fs) // 2] * 1e3, bs[len(bs) // 2] * 1e3          # us per layer pass

    lf, lb = timed(3)
    ops.set_option("rnn_recurrence_only", 1)
    try:
        kf, kb = timed(5)
    finally:
        ops.set_option("rnn_recurrence_only", 0)
    ops.check_health()
    flops = 2.0 * T * 2 * B * (G * H) * H              # recurrent matmul of one launch (both directions)
    # HBM bytes a recurrent launch must move (both directions): forward = pre-activations in + saved gates out (2 x G*H) + c / n
    # out (LSTM / GRU: H) + y out (H); backward = saved gates in (G*H) + c in (H) + dy in (H) + d(pre-act) out (G*H) -- both (2G + 2) * H floats
    # per (frame, row, direction)
    abytes = T * B * 2 * (2 * G + 2) * H * 4
    names = ops.rnn_last_kernels()                     # what the library really launched for this shape (not what the host expects)
    return dict(layer_fwd_us=lf, layer_bwd_us=lb, kernel_fwd_us=kf, kernel_bwd_us=kb, fwd_us_per_timestep=kf / T, bwd_us_per_timestep=kb / T,
                fwd_kernel=names[0], bwd_kernel=names[1], T=T,
                algorithmic_flops_per_launch=flops, algorithmic_bytes_per_launch=abytes,
                note="kernel_*: persistent recurrent launch alone (T dependent timesteps, both directions; includes its <10 us of memsets / "
                     "W_hh transposes); layer_*: with the input-projection (fwd) / deferred gradient (bwd) GEMMs")


def make_training_step(c, dev, rank, world):
    """Model, optimiser, resident synthetic batch and the step closure of one workload: forward -> CTC (sum) / B_global -> backward ->
    gradient all-reduce -> fused Adam.  The SAME closure is what the headline number and the `other_workloads` entries time."""
    from ctc_pytorch_amd import nn, parallel, ops as _ops
    from ctc_pytorch_amd.optim import FlatAdam
    from ctc_pytorch_amd.testing import synth                      # synthetic inputs only (no arithmetic)
    torch.manual_seed(1)
    model = build(c, dev, drop_out=c.get("drop", 0.1)).train()
    opt = FlatAdam(model, lr=1e-3, weight_decay=5e-4)
    parallel.broadcast_params(opt.flat)
    batch = synth.make_batch(seed=1 + rank, B=c["B"], T=c["T"], F=c.get("F", 40), V=c["V"], lab_lo=c["lab"][0], lab_hi=c["lab"][1],
                             full_length=True)
    x = torch.from_numpy(batch["x"]).to(dev)
    tg = torch.from_numpy(batch["targets"]).to(dev)
    tl = torch.from_numpy(batch["tgt_len"]).to(dev)
    loss_fn = nn.CTCLoss(reduction="sum")
    global_b = c["B"] * world
    state = {"in_len": None}
    overlapped, early_bytes = [0], [0]

    def step(mark=None):
        out = model(x)
        if state["in_len"] is None:
            state["in_len"] = torch.full((c["B"],), out.size(0), dtype=torch.int64, device=dev)
        loss = loss_fn(out, tg, state["in_len"], tl) / global_b
        opt.zero_grad()
        loss.backward()
        _ops.join_side_stream()
        overlapped[0] = len(parallel._overlap["done"])         # gradient slices already all-reduced behind their weight GEMMs
        early_bytes[0] = sum(hi - lo for lo, hi in parallel._overlap["done"])
        if mark is not None:
            mark[0].record()
        parallel.allreduce_grads(opt.grad)
        if mark is not None:
            mark[1].record()
        opt.step()
        return loss

    # the host enqueues a step ~7x faster than the GPU runs it; like a training loop that reads its statistics one step behind
    # (steps/train_ctc.run_epoch) it stays at most two steps ahead
    ring = [torch.cuda.Event() for _ in range(3)]

    def paced(i):
        out = step()
        ring[i % 3].record()
        if i >= 2:
            ring[(i - 2) % 3].synchronize()
        return out

    return dict(step=step, paced=paced, model=model, opt=opt, batch=batch, loss_fn=loss_fn, global_b=global_b, overlapped=overlapped,
                early_bytes=early_bytes)


def ragged_epoch(dev, c, model, opt, loss_fn, global_b, batch_sizes=(8, None), steps=200, seed=11):
    """The loop a user runs, on batches drawn like the reference's loader produces them (timit/utils/data_loader.py:119-151: a minibatch of
    utterances padded to ITS longest one, lengths as float32 fractions of that maximum): utterance lengths T ~ U{T/4 .. T} frames, label
    lengths U{15 .. 60} (capped so that CTC stays feasible), `steps` minibatches per batch size -- the reference's shipped batch_size 8 and
    the workload's own B -- collated by the product's create_input, staged by DevicePrefetcher, trained by steps/train_ctc.run_epoch.  Every
    step sees another (T_max, L_max): the per-call planners, the learnt batch-chunk marks, allocator growth and workspace reuse that the
    replayed full-length batch of the headline never exercises.  Reports frames/s on REAL (unpadded) frames, the padded
    rate beside it, the distinct (T_max, B) shapes seen, how often each recurrence kernel was launched (a shape that falls off the persistent
    path shows as rnn_fwd_step / rnn_bwd_step) and the allocator's peak."""
    from ctc_pytorch_amd import ops as _ops
    from ctc_pytorch_amd.steps.train_ctc import run_epoch
    from ctc_pytorch_amd.utils.data_loader import DevicePrefetcher, create_input
    import gc
    out = {}
    Fd, V, Tmax = c.get("F", 40), c["V"], c["T"]
    for B in batch_sizes:
        B = B or c["B"]
        rs = np.random.RandomState(seed + B)
        batches, real_frames, padded_frames, shapes = [], 0, 0, set()
        for _ in range(steps):
            utts = []
            for u in range(B):
                t = int(rs.randint(Tmax // 4, Tmax + 1))
                t_out = t // 2 if c["cnn"] else t
                l = int(min(rs.randint(15, 61), max(1, t_out // 3)))
                utts.append((torch.from_numpy(rs.standard_normal((t, Fd)).astype(np.float32)), torch.from_numpy(rs.randint(2, V, size=l).astype(np.int64)), "u%d" % u))
                real_frames += t
            b = create_input(utts)
            batches.append(b)
            padded_frames += int(b[0].shape[0]) * int(b[0].shape[1])
            shapes.add((int(b[0].shape[1]), int(b[0].shape[0])))
        pf = DevicePrefetcher(batches[:3], dev)
        run_epoch(0, model, pf, loss_fn, dev, optimizer=opt, print_every=10 ** 9, is_training=True, global_batch=None, log=lambda *_: None)   # (pinned slots, first shapes)
        torch.cuda.synchronize()
        _ops.kernel_counts(reset=True)
        torch.cuda.reset_peak_memory_stats(dev)
        gc.collect()
        gc.freeze()
        pf.loader = batches
        t0 = time.perf_counter()
        acc, avg = run_epoch(0, model, pf, loss_fn, dev, optimizer=opt, print_every=10 ** 9, is_training=True, global_batch=None, log=lambda *_: None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gc.unfreeze()
        _ops.check_health()
        counts = _ops.kernel_counts(reset=True)
        out["B%d" % B] = dict(batch_size=B, steps=steps, ms_per_step=dt / steps * 1e3, real_frames_per_s=real_frames / dt, padded_frames_per_s=padded_frames / dt,
                              padding_fraction=1.0 - real_frames / float(padded_frames), distinct_shapes_T_B=len(shapes),
                              T_max_range=[min(t for t, _ in shapes), max(t for t, _ in shapes)], recurrence_kernel_launches=counts,
                              per_timestep_fallback_launches=sum(n for k, n in counts.items() if k.endswith("_step")),
                              learnt_chunk_shapes=[list(k) for k in sorted(_ops._fallback_shapes)],
                              peak_allocated_bytes=int(torch.cuda.max_memory_allocated(dev)), peak_reserved_bytes=int(torch.cuda.max_memory_reserved(dev)),
                              final_avg_loss=float(avg))
        del pf, batches
    out["note"] = ("steps/train_ctc.run_epoch + DevicePrefetcher over minibatches collated like the reference's loader (utterances of T/4..T frames padded to the "
                   "batch maximum, fractions as lengths): every step another (T_max, L_max); value = real (unpadded) frames per second; not the headline")
    return out


def sync_bn_cost(args, steps=20):
    """What the EXACT data-parallel mode costs (BatchNorm statistics over the global batch: two all-reduces of 2*C doubles per BatchNorm layer and
    pass), as a number instead of an estimate: two child runs of this file on the one GPU with CTCN_FORCE_COLLECTIVES=1 -- the process group,
    RCCL's communicator and every collective of an N-rank step are really issued, over one rank -- without and with --sync-bn."""
    import subprocess
    out = {}
    for name, extra in (("per_shard_bn", []), ("sync_bn", ["--sync-bn"])):
        env = dict(os.environ, CTCN_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + (os.getpid() + len(out)) % 200), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--steps", str(steps), "--warmup", "3", "--precision", str(args.precision),
               "--no-decode", "--no-cpu-baseline", "--no-others", "--no-pmc", "--no-ragged", "--no-sync-bn-cost", "--train-only"] + extra
        try:
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240, cwd=ROOT)
            d = json.loads(p.stdout.strip().splitlines()[-1])
            out[name] = dict(ms_per_step=d["ms_per_step"], exposed_allreduce_us_median=d.get("comm", {}).get("exposed_allreduce_us_median"),
                             backend=d.get("comm", {}).get("backend"), early_slices=d.get("comm", {}).get("early_slices"))
        except Exception as e:      # noqa: BLE001
            out[name] = {"error": repr(e)}
    try:
        out["sync_bn_extra_ms_per_step"] = out["sync_bn"]["ms_per_step"] - out["per_shard_bn"]["ms_per_step"]
    except Exception:               # noqa: BLE001
        pass
    out["note"] = "one rank with forced collectives (RCCL self all-reduce): the launch + synchronisation cost of the extra collectives, not xGMI time"
    return out


def other_workloads(dev, precision, names=("cfg1", "cfg3", "cfg4", "ref_yaml"), steps=12, warmup=2, prewarm=25):
    """The other single-GPU BASELINE workloads (and the reference's shipped YAML shape) through the SAME step closure as the headline,
    in the same process, so that a driver-run line carries them (VERDICT r4 #1 iv): ms per step (barrier-free single rank: wall clock over
    `steps` steps between two synchronisations, after `prewarm` + `warmup` untimed ones), frames/s, and the per-timestep time of both
    persistent recurrences of one layer.  No CPU leg, no decode leg; a few seconds per workload."""
    from ctc_pytorch_amd import ops as _ops
    out = {}
    for name in names:
        c = dict(WORKLOADS[name])
        try:
            ts = make_training_step(c, dev, 0, 1)
            for i in range(prewarm + warmup):
                ts["paced"](i)
            torch.cuda.synchronize()
            gc_was = gc.isenabled()
            gc.collect()
            gc.disable()
            try:
                t0 = time.perf_counter()
                for i in range(steps):
                    loss = ts["paced"](i)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / steps
            finally:
                if gc_was:
                    gc.enable()
            _ops.check_health()
            rec = recurrence_probe(dev, c)
            out[name] = dict(ms_per_step=dt * 1e3, value=c["B"] * c["T"] / dt, unit="frames/s", steps=steps, warmup=warmup, prewarm_steps=prewarm,
                             final_loss=float(loss.detach()),
                             fwd_us_per_timestep=rec["fwd_us_per_timestep"], bwd_us_per_timestep=rec["bwd_us_per_timestep"],
                             fwd_kernel=rec["fwd_kernel"], bwd_kernel=rec["bwd_kernel"], recurrent_steps_per_layer=rec["T"],
                             workload="%s: %dx%d Bi%s, B=%d, T=%d, F=%d, V=%d%s, dropout %.1f" % (name, c["L"], c["H"], c["rnn"], c["B"], c["T"], c.get("F", 40),
                                                                                              c["V"], ", 2-layer CNN front-end" if c["cnn"] else "", c.get("drop", 0.1)))
            del ts
        except Exception as e:      # noqa: BLE001
            out[name] = {"error": repr(e)}
        gc.collect()
        torch.cuda.empty_cache()
    return out


def summarize_rccl_log(path, limit=48):
    """Rank 0's view of the communicator out of an NCCL_DEBUG=INFO file: topology, channels, algorithm / protocol tuning.  RCCL prints one
    'Tree N' / 'Ring N' / 'Channel N' line per channel (64-128 of them): the first two of a kind are kept and the count is appended."""
    import re
    keep = ("Channel", "Ring", "Tree", "nChannels", "Connected all", "Algo", "algo", "proto", "NET/", "P2P", "xgmi", "XGMI", "comm 0x", "NCCL_")
    per_channel, lines = {}, []
    with open(path, errors="replace") as fh:
        for ln in fh:
            if not any(k in ln for k in keep):
                continue
            m = re.search(r"NCCL INFO (Tree|Ring|Channel) \d+", ln)
            if m:
                per_channel[m.group(1)] = per_channel.get(m.group(1), 0) + 1
                if per_channel[m.group(1)] > 2:
                    continue
            if len(lines) < limit:
                lines.append(ln.strip()[-220:])
    return lines + ["(%d '%s N' lines in all)" % (n, k) for k, n in sorted(per_channel.items())]


def trouble_line(args, report, rccl_log):
    """The ONE JSON line of a run that did not get to its measurement (a rank failed, or the deadline passed): same keys as the normal line
    with `value` null, plus `error`, what every rank last reported (phase, failure text, hand-off status word, recurrence kernels) and RCCL's
    own topology lines when it got as far as writing them."""
    lines = None
    try:
        lines = summarize_rccl_log(rccl_log) if rccl_log and os.path.exists(rccl_log) else None
    except Exception:       # noqa: BLE001
        pass
    return {"metric": "acoustic frames/sec/GPU (train)", "value": None, "unit": "frames/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": None, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "data": "synthetic",
            "config": {"workload": args.workload}, "error": report.get("error"), "ranks": report.get("ranks"),
            "comm": {"ranks": args.gpus, "rccl_info": lines, "dp_safe": os.environ.get("CTCN_DP_SAFE", "0") == "1"},
            "hint": "re-run with CTCN_DP_SAFE=1 (no early slice all-reduce, no side stream, no pipelined projection) to separate RCCL / rendezvous "
                    "trouble from the co-residency contract of DESIGN.md section 6; CTCN_RNN_PERSISTENT=0 takes the persistent kernels out as well"}


def run_train(args):
    """Rank set-up and the safety net around the measurement: with more than one rank (or forced collectives) a RankMonitor watches every
    rank through the rendezvous store; a failing rank or a passed deadline (CTCN_BENCH_DEADLINE_S, default 900 s) still yields rank 0's JSON
    line -- with `error` and the per-rank records -- instead of a hang or a bare traceback."""
    from ctc_pytorch_amd import parallel
    # N > 1: have RCCL say what it built (rings / trees, channels, protocol) into a per-rank file, so that the first real SCALE record can be
    # interpreted -- must be in the environment before the communicator exists; the summary goes into the line's `comm` object
    rccl_log = None
    # (the GPU image exports NCCL_DEBUG=VERSION; only a caller who asked for INFO / TRACE or named a file of their own is left alone)
    theirs = os.environ.get("NCCL_DEBUG", "").upper() in ("INFO", "TRACE") or "NCCL_DEBUG_FILE" in os.environ
    multi = int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("CTCN_FORCE_COLLECTIVES", "0") == "1"
    if multi and not theirs:
        rccl_log = "/tmp/ctcn_rccl_%d_rank%s.log" % (os.getppid(), os.environ.get("RANK", "0"))
        os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH,TUNING,ENV", NCCL_DEBUG_FILE=rccl_log)
    monitor = None
    try:
        rank, world, local = parallel.init_from_env()
    except Exception as e:          # noqa: BLE001 -- the rendezvous itself failed (a peer never arrived within CTCN_DIST_TIMEOUT_S): no store to report through
        if multi and int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps(trouble_line(args, {"error": "rendezvous failed: %r" % (e,), "ranks": None}, rccl_log)), flush=True)
        raise
    if multi and torch.distributed.is_initialized():
        monitor = parallel.RankMonitor(rank, world, deadline_s=float(os.environ.get("CTCN_BENCH_DEADLINE_S", "900")),
                                       on_trouble=lambda rep: print(json.dumps(trouble_line(args, rep, rccl_log)), flush=True))
        monitor.progress("initialised", backend=torch.distributed.get_backend())
    try:
        _run_train(args, rank, world, local, monitor, rccl_log)
    except BaseException as e:      # noqa: BLE001
        if monitor is None or isinstance(e, SystemExit):
            raise
        record = {}
        try:        # (no device synchronisation here: the stream may be stuck behind the very thing that failed)
            from ctc_pytorch_amd import ops as _ops
            record["kernels"] = list(_ops.rnn_last_kernels())
        except Exception:   # noqa: BLE001
            pass
        monitor.fail(e, **record)
        if rank == 0:               # the watcher thread prints the line and ends the process; this is the fallback if it did not
            time.sleep(5.0)
            print(json.dumps(trouble_line(args, {"error": "rank 0 failed: %r" % (e,), "ranks": monitor.collect()}, rccl_log)), flush=True)
            monitor.close()
        sys.exit(3)
    if monitor is not None:
        monitor.close()


def _run_train(args, rank, world, local, monitor, rccl_log):
    from ctc_pytorch_amd import nn, parallel
    parallel.enable_sync_bn(bool(getattr(args, "sync_bn", False)))
    parallel.enable_overlap(True)       # per-layer gradient slices are all-reduced behind their weight GEMMs (no-op without collectives)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch `python bench.py --gpus N` or torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (the product has no CPU path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    c = dict(WORKLOADS[args.workload])
    for key in ("T", "B", "H", "L"):                                           # experiment aid (tools/ab_env.sh): the named workload at another T / B / H / L --
        if os.environ.get("CTCN_BENCH_" + key):                                # the config string of the line reports them; never a headline number
            c[key] = int(os.environ["CTCN_BENCH_" + key])
    if os.environ.get("CTCN_BENCH_RNN"):
        c["rnn"] = os.environ["CTCN_BENCH_RNN"]
    if args.scaling == "strong":           # SURVEY 8d "additionally global-batch-32 strong scaling for information": the workload's batch is GLOBAL
        if c["B"] % world:
            raise SystemExit("bench.py --scaling strong: the workload's batch of %d does not split over %d ranks" % (c["B"], world))
        c["B"] //= world
    from ctc_pytorch_amd import ops as _ops
    _ops.set_precision(args.precision)
    ts = make_training_step(c, dev, rank, world)
    step, paced, model, opt, batch, loss_fn, global_b = ts["step"], ts["paced"], ts["model"], ts["opt"], ts["batch"], ts["loss_fn"], ts["global_b"]
    overlapped, early_bytes = ts["overlapped"], ts["early_bytes"]
    losses = []

    # start-up transient: the first tens of steps of a fresh process carry allocator growth and the interpreter's first cyclic-GC passes
    # (five runs with 5 warm-up steps: 13.7-14.9 ms, with 40: 13.75-13.77), so PREWARM untimed steps run before the W warm-up steps the
    # command line asks for; the timed region is unchanged (exactly K steps between two barriers)
    # (tests, VERDICT r5 next 5: CTCN_BENCH_FAIL_RANK / CTCN_BENCH_FAIL_STEP make one rank raise at a step of the prewarm loop)
    fail_rank, fail_step = int(os.environ.get("CTCN_BENCH_FAIL_RANK", "-1")), int(os.environ.get("CTCN_BENCH_FAIL_STEP", "3"))
    if monitor is not None:
        monitor.progress("prewarm", steps=PREWARM)
    for i in range(PREWARM):
        if rank == fail_rank and i == fail_step:
            raise RuntimeError("injected failure of rank %d at step %d (CTCN_BENCH_FAIL_RANK)" % (rank, i))
        paced(i)
    torch.cuda.synchronize()
    if monitor is not None:
        monitor.progress("warmup", steps=args.warmup)
    for i in range(args.warmup):
        paced(i)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    torch.cuda.synchronize()
    # (as timeit does: no cyclic-GC pass of the interpreter inside the timed region -- a generation-2 collection of a process that holds
    # the torch module tree takes ~50 ms, i.e. 25 cfg1 steps; it was the 13.7 / 14.9 ms bimodality of short cfg2 runs as well)
    import gc
    gc_was = gc.isenabled()
    gc.collect()
    gc.disable()
    ticks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]      # per-step GPU time stamps (for the median only)
    if monitor is not None:
        monitor.progress("timed", steps=args.steps)
    t0 = time.perf_counter()
    ticks[0].record()
    for i in range(args.steps):
        losses.append(paced(i))
        ticks[i + 1].record()
    torch.cuda.synchronize()
    if gc_was:
        gc.enable()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    _ops.check_health()                # a hand-off timeout in a persistent kernel poisons the step: never report such a run
    per_rank_s = parallel.gather_over_ranks(dt, dev)     # every rank's own wall clock over the K steps (the headline uses the MAX)
    dt = parallel.max_over_ranks(dt, dev)
    last_loss = float(losses[-1].detach()) * world if losses else float("nan")
    # Everything below that issues a collective must run on EVERY rank (the other ranks are gone after the return): the exchange-step
    # measurement here, on all ranks; the rank-0-only probes further down contain no collective, and the two that do (host enqueue time,
    # epoch loop) are single-rank measurements that N > 1 skips.
    comm_marks = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    comm_err = None
    try:
        for mk in comm_marks:
            step(mk)
        torch.cuda.synchronize()
    except Exception as e:          # noqa: BLE001
        comm_err = repr(e)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    if monitor is not None:         # this rank's own record: its clock, its hand-off status word (0 = healthy), the recurrence kernels it ran
        monitor.finish(ms_per_step=per_rank_s[rank] / args.steps * 1e3 if rank < len(per_rank_s) else None, status=0,
                       kernels=list(_ops.rnn_last_kernels()), kernel_counts=_ops.kernel_counts(), exchange_error=comm_err)
    if rank != 0:
        return
    frames = c["B"] * c["T"] * world * args.steps
    value = frames / dt
    Fd = c.get("F", 40)
    feat_in = 32 * (((Fd + 2 - 3) // 2 + 1 + 2 - 3) // 2 + 1) if c["cnn"] else Fd          # two 3x3 convs, frequency stride 2 each, 32 channels
    wts = gemm_weights(c, feat_in)
    t_frames = c["T"] // 2 if c["cnn"] else c["T"]
    train_flops_per_step = 3 * 2 * wts * c["B"] * t_frames
    res = {
        "metric": "acoustic frames/sec/GPU (train) + utterances/sec beam-decode (`decode` object), TIMIT 4x320 BiLSTM" if args.workload == "cfg2" else "acoustic frames/sec (train), " + args.workload,
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": PREWARM,
        "ms_per_step": dt / args.steps * 1e3, "ms_per_step_median": float(np.median([ticks[i].elapsed_time(ticks[i + 1]) for i in range(args.steps)])) if args.steps else None,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32" if args.precision == 0 else "f32 via bf16x3 split-operand MFMA (f32 accumulate)", "data": "synthetic",
        "config": {"workload": "%s: %dx%d Bi%s + BN + Linear(%d) + CTC, B=%d/GPU, T=%d, F=%d%s, dropout %.1f, Adam" % (
            args.workload, c["L"], c["H"], c["rnn"], c["V"], c["B"], c["T"], Fd, ", 2-layer CNN front-end" if c["cnn"] else "", c.get("drop", 0.1)),
            "global_batch": global_b, "seq_len": c["T"], "parallelism": "dp%d" % world, "full_length_utterances": True,
            "sync_bn": bool(getattr(args, "sync_bn", False))},
        "final_loss": last_loss, "overlapped_allreduce_slices_last_step": overlapped[0],
        "model_tflops_per_s": train_flops_per_step * world / (dt / args.steps) / 1e12,
        "per_gpu_frames_per_s": value / world,
    }
    try:    # the exchange step, so that a SCALE record can be checked: who carried it, how many bytes, how much of it the main stream saw
        import torch.distributed as tdist
        if comm_err is not None:
            raise RuntimeError(comm_err)
        exposed = sorted(a.elapsed_time(b) * 1e3 for a, b in comm_marks)
        on = parallel._collectives_on()
        rccl_lines = summarize_rccl_log(rccl_log) if rccl_log and os.path.exists(rccl_log) else None
        res["per_rank_ms_per_step"] = [t / args.steps * 1e3 for t in per_rank_s]
        res["comm"] = dict(ranks=world, collectives_issued=bool(on), rccl_info=rccl_lines, dp_safe=parallel.dp_safe(),
                           per_rank=(monitor.collect() if monitor is not None else None),
                           backend=("none (single rank: allreduce_grads returns at once)" if not on else
                                    ("ctcn_comm_* (RCCL behind the C ABI)" if os.environ.get("CTCN_COMM", "0") == "1" else "torch.distributed/" + tdist.get_backend() + " (= RCCL on ROCm)")),
                           allreduce_bytes_per_step=int(opt.grad.numel() * 4), early_slices=overlapped[0], early_slice_bytes=int(early_bytes[0]),
                           exposed_allreduce_us_median=exposed[len(exposed) // 2],
                           note="exposed = HIP-event time of parallel.allreduce_grads on the main stream (waits for the early slices + reduces the remainder); "
                                "early slices are all-reduced on the side stream behind their layer's weight-gradient GEMMs, next to the recurrence of the layer below")
    except Exception as e:
        res["comm"] = {"error": repr(e)}
    if args.train_only:
        print(json.dumps(res))
        return
    try:
        # dominant kernels by GPU time (profiles/): the persistent recurrences.  Each is a chain of T dependent
        # [B x H] x [H x 4H] products, so its ceiling is the MFMA peak of the arithmetic it uses -- which B = 32 rows and an
        # 800-step dependence chain cannot approach: the per-timestep cost is two in-XCD L2 hand-offs, not flops.
        rec = recurrence_probe(dev, c)
        # `peak` is the guide's DENSE figure for the arithmetic the kernel issues (bf16 MFMA 2 500 TFLOP/s; f32 MFMA 157.3): `frac` is the
        # algorithmic fraction of that.  bf16x3 issues three bf16 MFMAs per algorithmic product, so the matrix pipes themselves are
        # 3x busier than `frac` says: `frac_of_bf16x3_ceiling` (peak / 3) is reported next to it, never instead of it.
        peak = PEAK_F32_MFMA_TFLOPS if args.precision == 0 else 2500.0
        # the roofline object describes the DOMINANT kernel: whichever recurrence (forward / backward) is the longer launch
        fwd_dom = rec["kernel_fwd_us"] >= rec["kernel_bwd_us"]
        kname, kus, kstep = ((rec["fwd_kernel"], rec["kernel_fwd_us"], rec["fwd_us_per_timestep"]) if fwd_dom else
                             (rec["bwd_kernel"], rec["kernel_bwd_us"], rec["bwd_us_per_timestep"]))
        tf = rec["algorithmic_flops_per_launch"] / (kus * 1e-6) / 1e12
        traffic, traffic_note = ((None, "--no-pmc" if args.no_pmc else "N > 1: single-rank measurement") if (args.no_pmc or world > 1)
                                 else measure_recurrence_traffic(c, kname, args.precision))
        if traffic is None:             # fall back to the committed passes (same kernel instantiation, same shape), and say so
            traffic = recurrence_traffic(args.workload, kname)
            traffic_note = None if traffic is None else ("HBM bytes per launch of this kernel instantiation from the committed rocprofv3 PMC passes (profiles/%s), not a "
                                                         "counter read of this run (%s)" % (PMC_FILE, traffic_note))
        res["roofline"] = dict(kernel="%s (%s recurrence of one Bi%s layer, T=%d dependent steps, both directions)" % (
                                   kname, "forward" if fwd_dom else "backward", c["rnn"], rec["T"]),
                               bound="mfma", achieved=tf, peak=peak, unit="TFLOP/s", frac=tf / peak, traffic=traffic,
                               traffic_source=traffic_note,
                               algorithmic_bytes_per_launch=rec["algorithmic_bytes_per_launch"],
                               frac_of_bf16x3_ceiling=None if args.precision == 0 else tf / (peak / 3.0),
                               us_per_launch=kus, us_per_dependent_step=kstep, dependent_steps_per_launch=rec["T"],
                               algorithmic_flops_per_launch=rec["algorithmic_flops_per_launch"],
                               peak_note=("f32 MFMA 157.3 TFLOP/s (dense)" if args.precision == 0 else "dense bf16 MFMA 2500 TFLOP/s; the kernel issues 3 bf16 MFMAs per algorithmic product (bf16x3)")
                               + "; latency-bound: see DESIGN.md section 5 for the per-step critical path")
        res["recurrence"] = rec
        res["roofline_gemm"] = gemm_roofline(dev, c)
        # the WHOLE step against both peaks (SURVEY 8d "mixed; state both"): algorithmic flops (3 x 2 x GEMM weights x frames) and the
        # store-once / load-once HBM bytes of algorithmic_bytes_per_step, over the measured step time
        step_s, nbytes = dt / args.steps, algorithmic_bytes_per_step(c)
        res["roofline_step"] = dict(algorithmic_flops_per_step=train_flops_per_step, algorithmic_bytes_per_step=nbytes,
                                    mfma=dict(achieved=train_flops_per_step / step_s / 1e12, peak=peak, unit="TFLOP/s", frac=train_flops_per_step / step_s / 1e12 / peak),
                                    hbm=dict(achieved=nbytes / step_s / 1e9, peak=PEAK_HBM_GBS, unit="GB/s", frac=nbytes / step_s / 1e9 / PEAK_HBM_GBS),
                                    note="per GPU; neither peak binds: %d dependent recurrence steps per training step at %.2f / %.2f us (forward / backward) are %.0f %% of it"
                                         % (2 * c["L"] * rec["T"], rec["fwd_us_per_timestep"], rec["bwd_us_per_timestep"],
                                            100.0 * c["L"] * rec["T"] * (rec["fwd_us_per_timestep"] + rec["bwd_us_per_timestep"]) * 1e-6 / step_s))
    except Exception as e:      # keep the headline line even if a probe fails
        res["roofline"] = {"error": repr(e)}
    try:        # host side of one step: time to ENQUEUE it (python + autograd + ~300 launches) with the device idle at the start
        if world > 1:
            raise RuntimeError("single-rank measurement (its steps contain collectives the other ranks no longer answer)")
        torch.cuda.synchronize()
        th = []
        for _ in range(3):
            t0 = time.perf_counter()
            step()
            th.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
        res["host_enqueue_ms_per_step"] = 1e3 * min(th)
    except Exception as e:
        res["host_enqueue_ms_per_step"] = None if world > 1 else repr(e)
    try:
        # the real training loop: steps/train_ctc.run_epoch over host batches staged by DevicePrefetcher -- on top
        # of the timed step above it runs the greedy error count (arg-max, collapse, edit distance) and ONE small D2H per step
        if world > 1:
            raise RuntimeError("single-rank measurement (run_epoch all-reduces its statistics)")
        from ctc_pytorch_amd.steps.train_ctc import run_epoch
        from ctc_pytorch_amd.utils.data_loader import DevicePrefetcher
        nloop = max(60, min(3 * args.steps, 120))  # (long enough to amortise the fill and drain of the prefetch pipeline: with 20 steps they were 3 % of the figure)
        hb = (torch.from_numpy(batch["x"]), torch.ones(c["B"], dtype=torch.float32), torch.from_numpy(batch["targets"]),
              torch.from_numpy(batch["tgt_len"]), ["u%d" % i for i in range(c["B"])])
        pf = DevicePrefetcher([hb] * 3, dev)        # one prefetcher for the run, as in steps/train_ctc.main: its pinned slots persist
        run_epoch(0, model, pf, loss_fn, dev, optimizer=opt, print_every=10 ** 9, is_training=True,
                  global_batch=global_b, log=lambda *_: None)
        torch.cuda.synchronize()
        import gc
        gc.collect()
        gc.freeze()                                  # as steps/train_ctc.main does before its first epoch
        t0 = time.perf_counter()
        pf.loader = [hb] * nloop
        run_epoch(0, model, pf, loss_fn, dev, optimizer=opt, print_every=10 ** 9, is_training=True,
                  global_batch=global_b, log=lambda *_: None)
        torch.cuda.synchronize()
        dte = (time.perf_counter() - t0) / nloop
        gc.unfreeze()
        res["epoch_loop"] = {"ms_per_step": dte * 1e3, "frames_per_s": c["B"] * c["T"] * world / dte, "steps": nloop,
                             "note": "steps/train_ctc.run_epoch with DevicePrefetcher (pinned host batch -> async H2D each step), on-device greedy "
                                     "error count, step statistics read one step behind through pinned memory; not the headline `value`"}
    except Exception as e:
        res["epoch_loop"] = {"skipped": str(e)} if world > 1 else {"error": repr(e)}
    if world == 1 and not args.no_ragged and not args.train_only:
        try:
            res["epoch_loop_ragged"] = ragged_epoch(dev, c, model, opt, loss_fn, global_b)
        except Exception as e:      # noqa: BLE001
            res["epoch_loop_ragged"] = {"error": repr(e)}
    if world == 1 and not args.no_sync_bn_cost and not args.train_only and os.environ.get("CTCN_FORCE_COLLECTIVES", "0") != "1":
        res["sync_bn_cost"] = sync_bn_cost(args)
    if world == 1 and not args.no_decode:
        try:        # the utterances/sec beam-decode half of BASELINE.json's metric (cfg5), with its own roofline / cpu_baseline
            res["decode"] = decode_leg(dev)
        except Exception as e:
            res["decode"] = {"error": repr(e)}
    if world == 1 and args.workload == "cfg2" and not args.no_others:
        res["other_workloads"] = other_workloads(dev, args.precision)
        if args.precision == 1:
            # the library's opt-in bf16 mode (option "gemm_bf16_single": the 256-row GEMM tiles multiply the bf16 roundings of their operands once
            # instead of the three bf16x3 products -- north_star's "within 1e-3 bf16 tolerance"; tests: loss within 3e-5 of the reference's, log-probs
            # 2-3e-3 mean next to the default mode's).  Reported next to the headline, never as it: `value` above is the f32-equivalent default.
            from ctc_pytorch_amd import ops as _ops
            try:
                _ops.set_option("gemm_bf16_single", 1)
                res["bf16_gemm_mode"] = dict(other_workloads(dev, args.precision, names=("cfg2", "cfg4"), steps=12, warmup=2, prewarm=10),
                                             option="gemm_bf16_single = 1 (default 0)",
                                             tolerance="OUTSIDE north_star's activation tolerance: loss and averages within 1e-3 of the reference's, single log-probs "
                                                       "move by up to 2e-2 (tests gate mean 5e-3 / max 5e-2) -- opt-in, not a headline, not a parity claim",
                                             note="same step closure and clock as `other_workloads`; time-parallel GEMMs (input projections, dx, weight "
                                                  "gradients) as ONE bf16 product with f32 accumulation, recurrent matmul still bf16x3; parity at this "
                                                  "mode's tolerance: tests/test_gpu_kernels.py::test_bf16_single_mode_against_reference_checksums")
            except Exception as e:      # noqa: BLE001
                res["bf16_gemm_mode"] = {"error": repr(e)}
            finally:
                _ops.set_option("gemm_bf16_single", 0)
    if not args.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = cpu_baseline_train(c, batch, steps=args.cpu_steps)
    print(json.dumps(res))


def decode_leg(dev, steps=5):
    """cfg5 -- the second half of BASELINE.json's metric: BeamDecoder W=20 + bigram LM (alpha 0.1) over 128 x (T=800, V=62)
    log-probs resident in HBM, both synthetic regimes of SURVEY 8d; utterances/s, checked against the C restatement of
    BeamSearch.py (oracle/beam_ref.c) on a bounded sample, which is also the CPU baseline (1 host core)."""
    from ctc_pytorch_amd import ops
    from ctc_pytorch_amd.utils.NgramLM import LanguageModel
    from oracle import beam_ref
    from ctc_pytorch_amd.testing import synth
    V, T, B, W = 62, 800, 128, 20
    i2c = synth.int2char(V)
    arpa = os.path.join(ROOT, "tests", "golden", "lm_phone_bg.arpa")
    tab = LanguageModel(arpa).table([i2c[i] for i in range(V)])
    NS = int(os.environ.get("CTCN_DECODE_STREAMS", "3"))              # searches in flight, one stream each, as steps/test_ctc.decode_and_score runs them
    out = {"metric": "utterances/sec beam-decode (W=20, bigram LM alpha=0.1, 128 x 800 x 62 log-probs in HBM; %d searches in flight, results handed to the host and assembled into phone strings)" % NS, "unit": "utt/s", "n_gpus": 1,
           "config": {"workload": "cfg5: BeamDecoder W=20 + phone bigram LM over 128 utterances x 800 frames x 62 classes, lens U{400..800}"},
           "regimes": {}}
    tab_dev = torch.as_tensor(tab, dtype=torch.float64).to(dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
    for regime in ("peaky", "flat"):
        lp = synth.make_logprobs(seed=7, T=T, B=B, V=V, regime=regime)
        lens = list(np.random.RandomState(2).randint(400, 801, size=B))
        x = torch.from_numpy(lp).to(dev)
        lens_dev = torch.as_tensor(lens, dtype=torch.int32).to(dev)         # resident like the log-probs
        ids, _, st = ops.beam_decode(x, lens, tab, 0.1, W)                  # warm-up + the strings that are checked
        torch.cuda.synchronize()
        # one batch at a time on one stream: the kernel time of a batch (HIP events) ...
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            dev_out = ops.beam_decode_device(x, lens_dev, tab_dev, 0.1, W)
        e1.record()
        ids_c, len_c = dev_out[0].cpu(), dev_out[1].cpu()                   # host hand-over of the last batch (synchronises)
        dt1 = (time.perf_counter() - t0) / steps
        kernel_us = e0.elapsed_time(e1) * 1e3 / steps
        # ... and the way steps/test_ctc.decode_and_score runs it: NS batches in flight on NS streams (a batch is one workgroup per
        # utterance = half of the CUs), every batch handed to the host through pinned memory (ops.beam_decode_async)
        nfl = 16 * max(steps, 4)
        warm = []
        for k in range(2 * NS):                                             # (the pinned hand-over buffers of the batches in flight exist after this)
            with torch.cuda.stream(streams[k % NS]):
                warm.append(ops.beam_decode_async(x, lens_dev, tab_dev, 0.1, W))
        for h in warm:
            h.result()
        del warm
        torch.cuda.synchronize()
        # (the id -> phone-string assembly BeamDecoder.decode performs on the host is part of a decoded batch: inside the timed loop)
        # (as BeamDecoder.decode_async does it: one pass of native host code over the pinned result buffer -- ctcn_join_tokens -- instead of
        # str.join per utterance, which made the flat regime's loop host-bound: 72 k tokens per batch, 2.3 ms of interpreter against 0.8 ms of
        # device time per batch with three searches in flight)
        phones = [i2c[i] for i in range(V)]
        finish = lambda h: h.strings(phones, " ")[0]
        t0 = time.perf_counter()
        pend = []
        for k in range(nfl):
            with torch.cuda.stream(streams[k % NS]):
                pend.append(ops.beam_decode_async(x, lens_dev, tab_dev, 0.1, W))
            if len(pend) == 2 * NS:                                         # NS searches running, NS queued behind them
                strings = finish(pend.pop(0))
        for h in pend:
            strings = finish(h)
        dt = (time.perf_counter() - t0) / nfl
        assert strings == [" ".join(map(phones.__getitem__, seq)) for seq in ids] and len(strings) == B
        # host side of one batch, apart: the time to ENQUEUE a search (workspace, launches, pinned copy, event) with the queue kept short, and
        # to FINISH one whose results are already on the host (string assembly) -- what the loop above needs per batch next to the device
        torch.cuda.synchronize()
        te, tf = [], []
        for k in range(12):
            t1 = time.perf_counter()
            with torch.cuda.stream(streams[k % NS]):
                hh = ops.beam_decode_async(x, lens_dev, tab_dev, 0.1, W)
            te.append(time.perf_counter() - t1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            finish(hh)
            tf.append(time.perf_counter() - t1)
        te.sort(); tf.sort()
        # frames the search really processes: the reference skips a frame when 1 - p(blank) < 0.1 (BeamSearch.py:93-94)
        pb = np.exp(lp[:, :, 0])
        processed = int(sum(int(((1.0 - pb[:lens[b], b]) >= 0.1).sum()) for b in range(B)))
        longest = max(int(((1.0 - pb[:lens[b], b]) >= 0.1).sum()) for b in range(B))
        r = {"value": B / dt, "ms_per_batch": dt * 1e3, "batches_in_flight": "%d on %d streams + %d queued behind them" % (NS, NS, NS), "value_one_batch_at_a_time": B / dt1, "kernel_us_per_batch": kernel_us,
             "processed_frames": processed, "host_us_per_batch": {"enqueue": te[len(te) // 2] * 1e6, "finish": tf[len(tf) // 2] * 1e6},
             "us_per_processed_frame_on_the_longest_utterance": kernel_us / max(longest, 1)}
        nref = 4 if regime == "flat" else 16
        probs = np.exp(lp[:, :nref, :]).transpose(1, 0, 2)
        t0 = time.time()
        want, _, _ = beam_ref.decode_ids(probs, lens[:nref], tab, 0.1, W)
        cdt = time.time() - t0
        r["cpu_baseline"] = {"value": nref / cdt, "unit": "utt/s", "cores": 1, "kind": "port",
                             "sample": "%d utterances of the same batch through oracle/beam_ref.c (C restatement of BeamSearch.py, 1 core, %.2f s)" % (nref, cdt)}
        r["strings_match_oracle"] = bool([list(map(int, s)) for s in want] == ids[:nref]) and bool((st == 0).all())
        r["roofline"] = dict(kernel="beam_prep_kernel + beam_fast_kernel (one launch each per 128-utterance batch; latency / fp64-ALU bound -- the HBM floor is reported, "
                                    "utt/s against the CPU is the figure of merit, SURVEY 8d)", bound="hbm",
                             achieved=lp.nbytes / (kernel_us * 1e-6) / 1e9, peak=PEAK_HBM_GBS, unit="GB/s",
                             frac=lp.nbytes / (kernel_us * 1e-6) / 1e9 / PEAK_HBM_GBS,
                             traffic=(pmc_traffic("beam_fast_kernel") or 0) + (pmc_traffic("beam_prep_kernel") or 0) or None if regime == "peaky" else None,
                             algorithmic_bytes_per_launch=int(lp.nbytes), us_per_launch=kernel_us)
        out["regimes"][regime] = r
    # the reference's class-default beam (ctcDecoder.py:170: beam_width = 200, lm_alpha = 0.01) on the same batches: the generic kernel (round 5:
    # pruning bound + rank count instead of W arg-max rounds per frame); one batch at a time, HIP events; one utterance per regime checked
    # against the C oracle (bounded: the oracle needs seconds per utterance at this width)
    try:
        wide = {"beam_width": 200, "lm_alpha": 0.01, "kernel": "beam_kernel<1024> (generic)"}
        for regime in ("peaky", "flat"):
            lp = synth.make_logprobs(seed=7, T=T, B=B, V=V, regime=regime)
            lens = list(np.random.RandomState(2).randint(400, 801, size=B))
            x = torch.from_numpy(lp).to(dev)
            lens_dev = torch.as_tensor(lens, dtype=torch.int32).to(dev)
            ops.beam_decode_device(x, lens_dev, tab_dev, 0.01, 200)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dev_out = ops.beam_decode_device(x, lens_dev, tab_dev, 0.01, 200)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            t0 = time.time()
            want, _, _ = beam_ref.decode_ids(np.exp(lp[:, :1, :]).transpose(1, 0, 2), lens[:1], tab, 0.01, 200)
            cdt = time.time() - t0
            got = dev_out[0][0, : int(dev_out[1][0])].cpu().tolist()
            wide[regime] = {"ms_per_batch": ms, "utt_per_s_one_batch_at_a_time": B / (ms * 1e-3), "first_utterance_matches_oracle": bool(list(map(int, want[0])) == got) and bool((dev_out[3] == 0).all()),
                            "cpu_oracle_utt_per_s": 1.0 / cdt}
            # ... and as steps/test_ctc.decode_and_score runs any width: NS searches in flight on NS streams, every batch handed to the host and
            # assembled into phone strings.  One batch is 128 workgroups of one CU each (99 KB of LDS per search) -- HALF the device -- and a
            # workgroup whose utterance is short (lens U{400..800}) leaves its CU idle until the batch's longest search ends: batches in flight fill both
            ids_w = [dev_out[0][k, : int(dev_out[1][k])].cpu().tolist() for k in range(B)]
            nfl_w = 8 * NS
            warm = []
            for k in range(NS):
                with torch.cuda.stream(streams[k % NS]):
                    warm.append(ops.beam_decode_async(x, lens_dev, tab_dev, 0.01, 200))
            for h in warm:
                h.result()
            del warm
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pend, strings = [], None
            for k in range(nfl_w):
                with torch.cuda.stream(streams[k % NS]):
                    pend.append(ops.beam_decode_async(x, lens_dev, tab_dev, 0.01, 200))
                if len(pend) == 2 * NS:
                    strings = finish(pend.pop(0))
            for h in pend:
                strings = finish(h)
            dtw = (time.perf_counter() - t0) / nfl_w
            wide[regime].update(utt_per_s=B / dtw, ms_per_batch_in_flight=dtw * 1e3, batches_in_flight="%d on %d streams + %d queued behind them" % (NS, NS, NS),
                                strings_match_the_one_at_a_time_run=bool(strings == [" ".join(map(phones.__getitem__, seq)) for seq in ids_w]))
        out["wide_beam"] = wide
    except Exception as e:      # noqa: BLE001
        out["wide_beam"] = {"error": repr(e)}
    out["value"] = out["regimes"]["peaky"]["value"]
    out["value_flat"] = out["regimes"]["flat"]["value"]
    out["strings_match_oracle"] = all(r["strings_match_oracle"] for r in out["regimes"].values())
    return out


def run_decode(args):
    print(json.dumps(decode_leg(torch.device("cuda", 0), steps=args.steps)))


def self_launch_argv(args, argv, environ):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): the command line and environment of
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>`
    -- one rank per GPU over RCCL, the form the driver uses for N > 1 (which keeps working: WORLD_SIZE is set there).  None when no
    re-launch is needed."""
    if args.mode != "train" or args.gpus <= 1 or "WORLD_SIZE" in environ:
        return None
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(environ, HSA_ENABLE_IPC_MODE_LEGACY=environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
             "--master-port", str(port), os.path.abspath(__file__)] + list(argv), env)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default, the driver's SCALE run): the workload's batch PER GPU; strong (SURVEY 8d, for information): the workload's batch is "
                         "the GLOBAL batch, sharded B/N utterances per rank")
    ap.add_argument("--sync-bn", action="store_true", help="BatchNorm statistics over the global batch (N-GPU == 1-GPU math); default: per shard")
    ap.add_argument("--mode", default="train", choices=["train", "decode"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the cfg5 beam-decode leg of the default (N=1, train) run")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic with rocprofv3 --pmc child passes (the committed profile is quoted instead)")
    ap.add_argument("--no-others", action="store_true", help="skip the `other_workloads` object (cfg1 / cfg3 / cfg4 / ref_yaml) of the default cfg2 run")
    ap.add_argument("--no-ragged", action="store_true", help="skip `epoch_loop_ragged` (run_epoch over minibatches of ragged utterances)")
    ap.add_argument("--no-sync-bn-cost", action="store_true", help="skip `sync_bn_cost` (two forced-collectives child runs, without / with --sync-bn)")
    ap.add_argument("--train-only", action="store_true", help="the timed step loop and the comm object only (what the sync_bn_cost children run)")
    ap.add_argument("--cpu-steps", type=int, default=1, help="timed CPU train steps per thread setting of the sweep")
    ap.add_argument("--precision", type=int, default=int(os.environ.get("CTCN_PRECISION", "1")), choices=[0, 1],
                    help="0: exact f32 MFMA GEMMs; 1: bf16x3 split-operand MFMA GEMMs (f32-class accuracy)")
    a = ap.parse_args()
    relaunch = self_launch_argv(a, sys.argv[1:], os.environ)
    if relaunch is not None:
        os.execvpe(relaunch[0][0], relaunch[0], relaunch[1])
    (run_train if a.mode == "train" else run_decode)(a)
