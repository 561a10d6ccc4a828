"""Do power-of-two row pitches cost the bf16x3 GEMMs anything?  cfg4's products (H = 512: rows of 1 024 / 3 072 floats = 4 KB / 12 KB) next to the
same products with the offending dimension moved off the power of two.  TFLOP/s per (kind, M, N, K); python tools/pitch_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_pytorch_amd import ops
ops.set_precision(1)
dev = torch.device("cuda", 0)
def timed(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
T = 76800
for kind, shapes in (("NT projection  C = x W^T ", [(T, 3072, 1024), (T, 3072, 1040), (T, 3072, 992)]),
                     ("NT dx          C = da W  ", [(T, 1024, 3072), (T, 1040, 3072), (T, 1024, 3088)]),
                     ("TN dW_ih       C = da^T x", [(1536, 1024, 76800), (1536, 1040, 76800), (1552, 1024, 76800), (1552, 1040, 76800)]),
                     ("TN dW_hh       C = da^T h", [(1536, 512, 76800), (1536, 528, 76800)])):
    for (M, N, K) in shapes:
        if kind.startswith("NT proj"):
            A, B, C = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
            us = timed(lambda: ops.gemm(0, 1, M, N, K, A, K, B, K, C, N))
        elif kind.startswith("NT dx"):
            A, B, C = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev), torch.empty(M, N, device=dev)
            us = timed(lambda: ops.gemm(0, 0, M, N, K, A, K, B, N, C, N))
        else:
            A, B, C = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev), torch.zeros(M, N, device=dev)
            us = timed(lambda: ops.gemm(1, 0, M, N, K, A, M, B, N, C, N, beta=1.0))
        print("%s %6d x %5d x %6d: %8.1f us  %6.1f TFLOP/s" % (kind, M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
