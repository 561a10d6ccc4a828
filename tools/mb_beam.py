"""Per-phase cycle budget of beam_fast_kernel (development aid): builds decode.hip with -DCTCN_BEAM_STATS into
tools/libctcn_beamstats.so, decodes the cfg5 batch and prints the cycles workgroup 0 spent per phase and frame.
    python tools/mb_beam.py build     (no GPU needed)        python tools/mb_beam.py run [peaky|flat]
    python tools/mb_beam.py build_plain <decode.hip> <out.so>;  python tools/mb_beam.py time <a.so> <b.so> ...   (A/B of two revisions on one box)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "libctcn_beamstats.so")
CS = os.path.join(ROOT, "ctc_pytorch_amd", "csrc")


def build():
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DCTCN_BEAM_STATS", "-o", SO,
                           os.path.join(CS, "decode.hip"), os.path.join(CS, "core.hip")])


def run(regime):
    import numpy as np
    import torch
    from ctc_pytorch_amd.utils.NgramLM import LanguageModel
    from ctc_pytorch_amd.testing import synth
    L = ctypes.CDLL(SO)
    V, T, B, W = 62, 800, 128, 20
    i2c = synth.int2char(V)
    tab = LanguageModel(os.path.join(ROOT, "tests", "golden", "lm_phone_bg.arpa")).table([i2c[i] for i in range(V)])
    lp = synth.make_logprobs(seed=7, T=T, B=B, V=V, regime=regime)
    lens = np.random.RandomState(2).randint(400, 801, size=B).astype(np.int32)
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(lp).to(dev)
    lens_t = torch.from_numpy(lens).to(dev)
    lm = torch.from_numpy(np.asarray(tab, dtype=np.float64)).to(dev)
    L.ctcn_beam_ws_bytes.restype = ctypes.c_size_t
    nb = L.ctcn_beam_ws_bytes(T, B, V, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    out_ids = torch.zeros((B, T), dtype=torch.int32, device=dev)
    out_len = torch.zeros(B, dtype=torch.int32, device=dev)
    score = torch.zeros(B, dtype=torch.float64, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    L.ctcn_beam_decode.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    for _ in range(2):
        rc = L.ctcn_beam_decode(P(x), 0, P(lens_t), P(lm), 0.1, W, 0, P(out_ids), P(out_len), P(score), P(status), T, B, V, P(ws), nb, None)
        assert rc == 0
        torch.cuda.synchronize()
    st = (ctypes.c_longlong * 64)()
    assert L.ctcn_beam_stats(st) == 0
    names = ["(loop top)", "-", "until the frame's last barrier (wave 0: publish + waiting; wave 3: waiting)", "wave 0: idle at the selection's barriers 1-3 | wave 3: pruning bound + prune / compact",
             "barrier + rank count + barrier", "wave 0: new beam (P4a) + publish", "wave 0: parent slots + gathers + pr | wave 3: waiting for the beam flag", "wave 0: waiting for the stay totals + log-adds",
             "wave 3: candidate load + row max", "wave 3: extension scores of the next frame", "(scratch stamp 10)", "(scratch stamp 11)"]
    for base, who in ((0, "wave 0"), (16, "wave 3")):
        nfl = max(st[base + 14], 1)
        print("%s  %s: frames %d, total %.0f cycles/frame, rounds/frame %.2f, merge iterations/frame %.2f" % (regime, who, nfl, st[base + 15] / nfl, st[base + 12] / nfl, st[base + 13] / nfl))
        for i, n in enumerate(names):
            print("    %-36s %8.0f cycles/frame" % (n, st[base + i] / nfl))


def run_generic(regime, W):
    """cycles per phase and frame of the GENERIC beam_kernel (workgroup 0, thread 0) on the cfg5 batch at beam width W (> 60: the generic kernel)"""
    import numpy as np
    import torch
    from ctc_pytorch_amd.utils.NgramLM import LanguageModel
    from ctc_pytorch_amd.testing import synth
    L = ctypes.CDLL(SO)
    V, T, B = 62, 800, 128
    i2c = synth.int2char(V)
    tab = LanguageModel(os.path.join(ROOT, "tests", "golden", "lm_phone_bg.arpa")).table([i2c[i] for i in range(V)])
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(synth.make_logprobs(seed=7, T=T, B=B, V=V, regime=regime)).to(dev)
    lens_t = torch.from_numpy(np.random.RandomState(2).randint(400, 801, size=B).astype(np.int32)).to(dev)
    lm = torch.from_numpy(np.asarray(tab, dtype=np.float64)).to(dev)
    L.ctcn_beam_ws_bytes.restype = ctypes.c_size_t
    nb = L.ctcn_beam_ws_bytes(T, B, V, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    out_ids = torch.zeros((B, T), dtype=torch.int32, device=dev)
    out_len = torch.zeros(B, dtype=torch.int32, device=dev)
    score = torch.zeros(B, dtype=torch.float64, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    L.ctcn_beam_decode.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    for _ in range(2):
        assert L.ctcn_beam_decode(P(x), 0, P(lens_t), P(lm), 0.1, W, 0, P(out_ids), P(out_len), P(score), P(status), T, B, V, P(ws), nb, None) == 0
        torch.cuda.synchronize()
    st = (ctypes.c_longlong * 64)()
    assert L.ctcn_beam_stats(st) == 0
    nfl = max(st[48], 1)
    names = ["loop top (skip test, pointers)", "ln p of the frame + parent slots (table lookup) + barrier", "extension scores + barrier", "stay entries / merges + barrier",
             "selection (maxima, bound, compaction, rank count | arg-max rounds)", "new beam (trie lookups / inserts) + barrier"]
    print("%s W=%d generic kernel, workgroup 0: %d processed frames, %.0f cycles per frame" % (regime, W, nfl, (sum(st[32 + i] for i in range(6)) + sum(st[40 + i] for i in range(7))) / nfl))
    for i, n in enumerate(names):
        print("    %-70s %9.0f cycles/frame" % (n, st[32 + i] / nfl))
    for i, n in enumerate(["selection: scan for the maxima per thread", "selection: bound, rest (second barrier, block-wide counts, theta; the three parts below come on top)", "selection: count + compact the survivors (two scans, + barrier)",
                           "selection: survivors sorted (bitonic, <= 256) | pair counts (+ barrier)"]):
        print("    %-70s %9.0f cycles/frame" % (n, st[40 + i] / nfl))
    for i, n in enumerate(["  bound: wave sum of the counts + the wave's ballot search", "  bound: waiting at the first barrier (the slowest wave's scan + search)", "  bound: the waves' bounds against this wave's maxima (16 ballots)"]):
        print("    %-70s %9.0f cycles/frame" % (n, st[44 + i] / nfl))
    print("    survivors of the pruning bound per frame %.0f; frames whose survivors overflowed into the arg-max rounds %d" % (st[38] / nfl, st[39]))


def build_plain(src, out):
    """decode.hip `src` (e.g. an older revision written to a temporary file) -> un-instrumented library `out`, for A/B timing on one box"""
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", CS, "-o", out, src,
                           os.path.join(CS, "core.hip")])


def time_lib(so, iters=20):
    """kernel time of one cfg5 batch (HIP events around ctcn_beam_decode, both regimes) with the library `so`"""
    import numpy as np
    import torch
    from ctc_pytorch_amd.utils.NgramLM import LanguageModel
    from ctc_pytorch_amd.testing import synth
    L = ctypes.CDLL(so)
    V, T, B, W = 62, 800, 128, 20
    i2c = synth.int2char(V)
    tab = LanguageModel(os.path.join(ROOT, "tests", "golden", "lm_phone_bg.arpa")).table([i2c[i] for i in range(V)])
    dev = torch.device("cuda", 0)
    L.ctcn_beam_ws_bytes.restype = ctypes.c_size_t
    nb = L.ctcn_beam_ws_bytes(T, B, V, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    L.ctcn_beam_decode.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lens = np.random.RandomState(2).randint(400, 801, size=B).astype(np.int32)
    lens_t = torch.from_numpy(lens).to(dev)
    lm = torch.from_numpy(np.asarray(tab, dtype=np.float64)).to(dev)
    res = {}
    for regime in ("peaky", "flat"):
        x = torch.from_numpy(synth.make_logprobs(seed=7, T=T, B=B, V=V, regime=regime)).to(dev)
        out_ids = torch.zeros((B, T), dtype=torch.int32, device=dev)
        out_len = torch.zeros(B, dtype=torch.int32, device=dev)
        score = torch.zeros(B, dtype=torch.float64, device=dev)
        status = torch.zeros(B, dtype=torch.int32, device=dev)
        ts = []
        for it in range(iters + 3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = L.ctcn_beam_decode(P(x), 0, P(lens_t), P(lm), 0.1, W, 0, P(out_ids), P(out_len), P(score), P(status), T, B, V, P(ws), nb, None)
            e1.record()
            assert rc == 0
            torch.cuda.synchronize()
            if it >= 3:
                ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        res[regime] = (ts[len(ts) // 2], int(out_len.sum().item()), float(score.sum().item()), int(status.abs().sum().item()))
        print("%s  %s: median %.1f us per batch (min %.1f)   [sum len %d, sum score %.9f, status %d]" % (os.path.basename(so), regime, ts[len(ts) // 2], ts[0], res[regime][1], res[regime][2], res[regime][3]))
    return res


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "generic":
        run_generic(sys.argv[2], int(sys.argv[3]))
    elif sys.argv[1] == "build_plain":
        build_plain(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "time":
        for so in sys.argv[2:]:
            time_lib(os.path.abspath(so))
    else:
        run(sys.argv[2] if len(sys.argv) > 2 else "flat")
