"""Round 6, the cfg4 divergence: WHERE in the bottom layer does a deviating run leave the others?  Runs IN the process of a pytest subset
(CTCN_AFTER_SUITE, tests/conftest.py) -- the divergence needs such a process.  Parts (CTCN_PROBE_PARTS, default "gemm,rec,layer,runs"):
  gemm    the bottom layer's input projection alone (cfg4: 76 800 x 3 072 x 40 as the two row blocks), n times, outputs compared bit for bit
  rec     the recurrence alone (option rnn_recurrence_only) on fixed pre-activations copied into the reserve before every launch
  layer   projection + recurrence through the C ABI (forward only), n times
  runs    traced 12-step cfg4 training runs with the first / last eight timesteps of the bottom layer's reserve and output KEPT per step, arms
          interleaved (CTCN_PROBE_ARMS, "name:option=value+option=value,..."); a deviating run is compared element by element with the arm's first run
One JSON line per finding on stdout and in CTCN_PROBE_OUT."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from ctc_pytorch_amd import _lib, ops

dev = torch.device("cuda", 0)
OUT = os.environ.get("CTCN_PROBE_OUT", os.path.join(ROOT, "gpurun_out", "first_rows_probe.jsonl"))
os.makedirs(os.path.dirname(OUT), exist_ok=True)
parts = os.environ.get("CTCN_PROBE_PARTS", "gemm,rec,layer,runs").split(",")
N = int(os.environ.get("CTCN_PROBE_N", "300"))
RUNS = int(os.environ.get("CTCN_PROBE_RUNS", "30"))


def emit(**kw):
    line = json.dumps(kw)
    print("[probe] " + line, flush=True)
    with open(OUT, "a") as f:
        f.write(line + "\n")


def describe(ref, got, T, B, D, GH, rows=None):
    """element pattern of the difference of two (T', B, D, GH) tensors: per (row, direction) counts, batch rows, column structure"""
    neq = (ref.view(torch.int32) != got.view(torch.int32))
    out = []
    for ti in range(neq.shape[0]):
        for d in range(D):
            m = neq[ti, :, d, :]
            n = int(m.sum())
            if n == 0:
                continue
            bs = torch.nonzero(m.any(dim=1)).flatten().tolist()
            cs = torch.nonzero(m.any(dim=0)).flatten().tolist()
            diff = (ref[ti, :, d, :].double() - got[ti, :, d, :].double()).abs()
            out.append(dict(t=int(rows[ti]) if rows is not None else ti, d=d, n=n, of=int(m.numel()), batch_rows=bs if len(bs) <= 20 else [bs[0], "..", bs[-1], len(bs)],
                            cols=cs if len(cs) <= 24 else [cs[0], "..", cs[-1], len(cs)], slices16=sorted({c // 16 for c in cs})[:40], maxabs=float(diff.max()),
                            sample=[(int(b_), int(c_), float(ref[ti, b_, d, c_]), float(got[ti, b_, d, c_])) for b_, c_ in torch.nonzero(m)[:6].tolist()]))
    return out


T, B, I, H, D, G = 1200, 64, 40, 512, 2, 3
GH = G * H
L = _lib.lib()
ops.set_precision(1)
torch.manual_seed(21)
x = torch.randn(T, B, I, device=dev)
w_ih = torch.randn(D * GH, I, device=dev) * 0.2
w_hh = torch.randn(D, GH, H, device=dev) * (1.0 / H ** 0.5)
ws_t, wp, wn = ops._ws(x)
st = _lib.stream_ptr()


def project(gates):
    for t0, t1 in ((T // 2, T), (0, T // 2)):
        _lib.check(L.ctcn_gemm(0, 1, (t1 - t0) * B, D * GH, I, ctypes.c_void_p(x[t0].data_ptr()), I, ops._ptr(w_ih), I, ctypes.c_void_p(gates[t0].data_ptr()), D * GH, ctypes.c_float(0.0), 1, wp, wn, st), "gemm")


def rnn_fwd(gates, y, aux):
    call = _lib.RnnCall()
    call.status = _lib.status_word(dev).data_ptr()
    _lib.check(L.ctcn_rnn_fwd_ex(1, T, B, I, H, D, ops._ptr(x), ctypes.c_void_p(w_ih.data_ptr()), ctypes.c_void_p(w_hh[0].data_ptr()), ctypes.c_void_p(w_ih[GH].data_ptr()), ctypes.c_void_p(w_hh[1].data_ptr()),
                                 ops._ptr(y), ops._ptr(gates), ops._ptr(aux), 1, wp, wn, st, ctypes.byref(call)), "rnn_fwd_ex")


if "gemm" in parts:
    ref = torch.empty(T, B, D, GH, device=dev); got = torch.empty_like(ref)
    project(ref)
    bad = []
    t0_ = time.time()
    for i in range(N):
        got.fill_(float("nan")) if i % 2 else None
        project(got)
        if not torch.equal(ref.view(torch.int32), got.view(torch.int32)):
            bad.append(i)
            if len(bad) <= 3:
                rows = torch.nonzero((ref.view(torch.int32) != got.view(torch.int32)).reshape(T, -1).any(dim=1)).flatten()
                emit(part="gemm", iteration=i, rows_differ=[int(rows[0]), int(rows[-1]), int(len(rows))], detail=describe(ref[rows[:6]], got[rows[:6]], T, B, D, GH, rows[:6].tolist())[:12])
    emit(part="gemm", n=N, deviating=len(bad), which=bad[:20], seconds=round(time.time() - t0_, 1))
    del ref, got

if "rec" in parts or "layer" in parts:
    pre = torch.empty(T, B, D, GH, device=dev)
    project(pre)
    gates = torch.empty_like(pre)
    y_ref = torch.empty(T, B, D * H, device=dev); g_ref = torch.empty_like(pre); aux = torch.empty(T, B, D, H, device=dev)
    y = torch.empty_like(y_ref)
    for part in ("rec", "layer"):
        if part not in parts:
            continue
        ops.set_option("rnn_recurrence_only", 1 if part == "rec" else 0)
        try:
            gates.copy_(pre)
            rnn_fwd(gates, y_ref, aux)
            g_ref.copy_(gates)
            bad = []
            t0_ = time.time()
            for i in range(N):
                if part == "rec":
                    gates.copy_(pre)
                elif i % 2:
                    gates.copy_(g_ref)                  # what a training step finds there: the previous step's activations
                rnn_fwd(gates, y, aux)
                if not torch.equal(y.view(torch.int32), y_ref.view(torch.int32)):
                    bad.append(i)
                    if len(bad) <= 4:
                        yy, rr = y.view(T, B, D, H), y_ref.view(T, B, D, H)
                        per = (yy.view(torch.int32) != rr.view(torch.int32)).reshape(T, B, D, H).any(dim=3).any(dim=1)          # (T, D)
                        info = {}
                        for d in range(D):
                            ts = torch.nonzero(per[:, d]).flatten()
                            if len(ts):
                                first = int(ts[0]) if d == 0 else int(ts[-1])
                                near = [first + k * (1 if d == 0 else -1) for k in range(4) if 0 <= first + k * (1 if d == 0 else -1) < T]
                                info["d%d" % d] = dict(timesteps=int(len(ts)), first_in_direction_order=first,
                                                       y=describe(rr[near], yy[near], T, B, D, H, near)[:6], gates=describe(g_ref[near], gates[near], T, B, D, GH, near)[:6])
                        emit(part=part, iteration=i, detail=info)
            ops.check_health(dev)
            emit(part=part, n=N, deviating=len(bad), which=bad[:20], seconds=round(time.time() - t0_, 1), kernel=ops.rnn_last_kernels()[0])
        finally:
            ops.set_option("rnn_recurrence_only", 0)
    del pre, gates, y_ref, g_ref, aux, y

if "rec2" in parts or "layer2" in parts:
    # two inputs A / B alternate: a stale read shows WHICH older content it returned (the other input's pre-activations, the activations the
    # previous launch left in the reserve, or something else)
    xs = [x, x * 1.7 + 0.3]
    pres = [torch.empty(T, B, D, GH, device=dev) for _ in xs]
    x_keep = x
    for k_ in range(2):
        x = xs[k_]; project(pres[k_])
    gates = torch.empty_like(pres[0]); aux = torch.empty(T, B, D, H, device=dev); y = torch.empty(T, B, D * H, device=dev)
    for part in ("rec2", "layer2"):
        if part not in parts:
            continue
        ops.set_option("rnn_recurrence_only", 1 if part == "rec2" else 0)
        try:
            refs = []
            for k_ in range(2):
                x = xs[k_]
                gates.copy_(pres[k_]); rnn_fwd(gates, y, aux); refs.append((y.clone(), gates.clone()))
            bad = 0
            t0_ = time.time()
            for i in range(N):
                k_ = i % 2
                x = xs[k_]
                if part == "rec2":
                    gates.copy_(pres[k_])
                rnn_fwd(gates, y, aux)
                if not torch.equal(y.view(torch.int32), refs[k_][0].view(torch.int32)):
                    bad += 1
                    if bad <= 6:
                        info = []
                        for d, t in ((0, 0), (1, T - 1), (0, 1), (1, T - 2)):
                            m = gates[t, :, d].view(torch.int32) != refs[k_][1][t, :, d].view(torch.int32)
                            if not bool(m.any()):
                                continue
                            idx = torch.nonzero(m)
                            bs = sorted(set(idx[:, 0].tolist())); cs = sorted(set((idx[:, 1] // 16).tolist()))
                            # candidates for what a wrong element was computed from (step 0: r, z = sigmoid(pre), n = tanh(pre))
                            def act(p_):
                                a_ = p_[t, :, d].clone(); a_[:, :2 * H] = torch.sigmoid(a_[:, :2 * H]); a_[:, 2 * H:] = torch.tanh(a_[:, 2 * H:]); return a_
                            got = gates[t, :, d][m]
                            cand = dict(this_input=act(pres[k_])[m], other_input=act(pres[1 - k_])[m], act_of_prev_reserve=act(refs[1 - k_][1])[m], prev_reserve=refs[1 - k_][1][t, :, d][m],
                                        this_next_t=act(pres[k_].roll(-1 if d == 0 else 1, 0))[m], other_next_t=act(pres[1 - k_].roll(-1 if d == 0 else 1, 0))[m])
                            info.append(dict(t=t, d=d, n=int(m.sum()), batch_rows=bs, slices16=cs[:12], close_to={k2: float((v2 - got).abs().max()) for k2, v2 in cand.items()},
                                             sample=[float(v_) for v_ in got[:3]], expected=[float(v_) for v_ in refs[k_][1][t, :, d][m][:3]]))
                        emit(part=part, iteration=i, input=k_, detail=info)
            ops.check_health(dev)
            emit(part=part, n=N, deviating=bad, seconds=round(time.time() - t0_, 1), kernel=ops.rnn_last_kernels()[0])
        finally:
            ops.set_option("rnn_recurrence_only", 0)
    x = x_keep
    del pres, gates, aux, y

if "runs" in parts:
    import squat_stress
    arms = []
    for spec in os.environ.get("CTCN_PROBE_ARMS", "base:,rsv0:fwd_rsv_lds=0,clear:rnn_dbg=1,sync:rnn_dbg=2,acq:rnn_dbg=4,syncproj:rnn_dbg=10").split(","):
        name, _, opts = spec.partition(":")
        arms.append((name, [(o.split("=")[0], int(o.split("=")[1])) for o in opts.split("+") if o]))
    TT = 1200
    rows = list(range(8)) + list(range(TT - 8, TT))
    ref, stats = {}, {a[0]: dict(runs=0, deviating=0) for a in arms}
    saved = 0
    for i in range(RUNS):
        for name, opts in arms:
            found = [(k, ops.get_option(k)) for k, _ in opts]
            for k, v in opts:
                ops.set_option(k, v)
            cap = []
            try:
                r = squat_stress.run("cfg4", 12, squat=(os.environ.get("CTCN_PROBE_SQUAT") == "alt" and i % 2 == 1), seed=i + 1, dev=dev, trace=True, rows_capture=(rows, cap))
            finally:
                for k, v in found:
                    ops.set_option(k, v)
            stats[name]["runs"] += 1
            if name not in ref:
                ref[name] = cap
                continue
            rc = ref[name]
            for step, (a_, b_) in enumerate(zip(rc, cap)):
                if not (torch.equal(a_["gates"].view(torch.int32), b_["gates"].view(torch.int32)) and torch.equal(a_["y"].view(torch.int32), b_["y"].view(torch.int32))):
                    stats[name]["deviating"] += 1
                    gd = describe(a_["gates"], b_["gates"], TT, 64, 2, 1536, rows)
                    yd = describe(a_["y"].view(len(rows), 64, 2, 512), b_["y"].view(len(rows), 64, 2, 512), TT, 64, 2, 512, rows)
                    emit(part="runs", arm=name, run=i, step=step, gates=gd[:16], y=yd[:16], losses=r["losses"][:step + 2])
                    # which pre-activation was a wrong first-step element computed from?  r, z = sigmoid(a): a_got = logit(got), compared with the pre-activations
                    # (f32 matmul in the hook) of every kept timestep, both directions, this step and the previous one
                    for d_, ti in ((0, 0), (1, len(rows) - 1)):
                        m = (a_["gates"][ti, :, d_, :1024].view(torch.int32) != b_["gates"][ti, :, d_, :1024].view(torch.int32))
                        if not bool(m.any()) or int(m.sum()) > 4096 or b_.get("pre") is None:
                            continue
                        got = b_["gates"][ti, :, d_, :1024][m].double()
                        a_got = torch.log(got / (1 - got))
                        cands = {}
                        for nm, capk in (("this", cap[step]), ("prev", cap[step - 1] if step > 0 else rc[-1]), ("next", cap[step + 1] if step + 1 < len(cap) else None)):
                            if capk is None or capk.get("pre") is None:
                                continue
                            for tj in range(len(rows)):
                                for dj in (0, 1):
                                    cands["%s t=%d d=%d" % (nm, rows[tj], dj)] = float((capk["pre"][tj, :, dj, :1024][m].double() - a_got).abs().max())
                        best = sorted(cands.items(), key=lambda kv: kv[1])[:5]
                        # GRU: aux = hn = the recurrent product of the n gate, exactly 0 at a direction's first step
                        auxrow = b_["aux"][ti, :, d_] if b_.get("aux") is not None else None
                        aux_info = dict(nonzero=int((auxrow != 0).sum()), maxabs=float(auxrow.abs().max()), rows=sorted(set(torch.nonzero(auxrow)[:, 0].tolist())),
                                        slices16=sorted(set((torch.nonzero(auxrow)[:, 1] // 16).tolist())), sample=[float(v_) for v_ in auxrow[auxrow != 0][:6]]) if auxrow is not None else None
                        emit(part="runs-source", arm=name, run=i, step=step, d=d_, t=rows[ti], n=int(m.sum()), best=best, aux_first_step=aux_info, expected=cands.get("this t=%d d=%d" % (rows[ti], d_)),
                             sample_a_got=[float(v_) for v_ in a_got[:4]], sample_pre_this=[float(v_) for v_ in cap[step]["pre"][ti, :, d_, :1024][m][:4]])
                    if saved < 4:                     # the differing elements themselves (sparse)
                        m = torch.nonzero(a_["gates"].view(torch.int32) != b_["gates"].view(torch.int32))[:400000]
                        np.savez_compressed(os.path.join(os.path.dirname(OUT), "rows_%s_run%d_step%d.npz" % (name, i, step)), rows=np.array(rows), index=m.cpu().numpy(),
                                            ref=a_["gates"][tuple(m.t())].cpu().numpy(), got=b_["gates"][tuple(m.t())].cpu().numpy())
                        saved += 1
                    break
            del cap
    emit(part="runs", arms=stats)
