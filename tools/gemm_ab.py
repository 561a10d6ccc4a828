import ctypes, torch, sys
sys.path.insert(0, '.')
from ctc_pytorch_amd import _lib, ops
dev = torch.device('cuda:0')
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr())
def run(M, N, K, tA, tB, use_ws, reps=20):
    A = torch.randn((K, M) if tA else (M, K), device=dev)
    Bm = torch.randn((N, K) if tB else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    ws = _lib.workspace(dev)
    lda = A.shape[1]; ldb = Bm.shape[1]
    st = _lib.stream_ptr()
    def call():
        _lib.check(L.ctcn_gemm(tA, tB, M, N, K, P(A), lda, P(Bm), ldb, P(C), N, 0.0, 1, P(ws) if use_ws else None, ws.numel() if use_ws else 0, st), "gemm")
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    return us, C
for (M, N, K, tA, tB, name) in [(25600, 640, 2560, 0, 0, "dx"), (25600, 2560, 640, 0, 1, "fwd proj")]:
    u1, c1 = run(M, N, K, tA, tB, True)
    u0, c0 = run(M, N, K, tA, tB, False)
    print("%s %dx%dx%d: planes path %.1f us, in-kernel split %.1f us" % (name, M, N, K, u1, u0))
