"""A/B of two builds of decode.hip (tools/mb_beam.py build_plain) on the fuzz batches of test_beam_fuzz_small_alphabets_vs_c_oracle: labellings and
status against the C oracle, ulp distance of the float64 scores, and whether the two builds are bit-equal to each other.
    python tools/beam_fuzz_ab.py tools/libbeam_r3.so tools/libbeam_new.so      (paths relative to the repository root)"""
import ctypes, os, sys, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from oracle import beam_ref
from ctc_pytorch_amd.testing import synth
def run(so, probs_tbv, lens, tab, alpha, W):
    L = ctypes.CDLL(so)
    T, B, V = probs_tbv.shape
    dev = torch.device("cuda", 0)
    L.ctcn_beam_ws_bytes.restype = ctypes.c_size_t
    nb = L.ctcn_beam_ws_bytes(T, B, V, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    L.ctcn_beam_decode.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    x = torch.from_numpy(probs_tbv).to(dev)
    lens_t = torch.from_numpy(np.asarray(lens, dtype=np.int32)).to(dev)
    lm = torch.from_numpy(np.asarray(tab, dtype=np.float64)).to(dev)
    out_ids = torch.zeros((B, T), dtype=torch.int32, device=dev); out_len = torch.zeros(B, dtype=torch.int32, device=dev)
    score = torch.zeros(B, dtype=torch.float64, device=dev); status = torch.zeros(B, dtype=torch.int32, device=dev)
    rc = L.ctcn_beam_decode(P(x), 1, P(lens_t), P(lm), alpha, W, 0, P(out_ids), P(out_len), P(score), P(status), T, B, V, P(ws), nb, None)
    assert rc == 0
    torch.cuda.synchronize()
    return out_ids.cpu().numpy(), out_len.cpu().numpy(), score.cpu().numpy(), status.cpu().numpy()
cfgs = [(3, 2, "flat", 0.0), (4, 5, "flat", 0.5), (4, 20, "flat", 0.1), (8, 33, "flat", 0.3), (8, 52, "peaky", 0.1), (5, 52, "flat", 1.0), (30, 10, "flat", 0.1), (200, 10, "peaky", 0.1), (200, 16, "flat", 0.2), (62, 1, "flat", 0.1)]
for V, W, regime, alpha in cfgs:
    T, B = 70, 14
    rs = np.random.RandomState(1000 * V + W)
    lp = synth.make_logprobs(seed=V * 7 + W, T=T, B=B, V=V, regime=regime)
    lens = list(rs.randint(T // 3, T + 1, size=B)); lens[0], lens[1], lens[2] = 0, 1, T
    tab = -3.0 * rs.random_sample((V + 1, V + 1))
    probs = np.exp(lp).astype(np.float32)
    want, wscore, wst = beam_ref.decode_ids(probs.transpose(1, 0, 2), lens, tab, alpha, W)
    res = {}
    for name in sys.argv[1:]:
        ids, ln, sc, st = run(os.path.join(ROOT, name), probs, lens, tab, alpha, W)
        got = [list(ids[b, :ln[b]]) for b in range(B)]
        ok_ids = got == [list(map(int, s)) for s in want]
        ulp = np.max(np.abs(sc - wscore) / np.maximum(np.spacing(np.abs(wscore)), 1e-300))
        res[name] = sc
        print(V, W, regime, alpha, name, "ids", ok_ids, "status", list(st) == list(wst), "max ulp vs oracle %.1f" % ulp)
    if len(res) == 2:
        a, b_ = list(res.values())
        print("      the two libraries bit-equal:", np.array_equal(a, b_))
