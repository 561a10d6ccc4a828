"""The TN weight-gradient tile on cfg4's / cfg2's products with the split-K count forced (development option "tn_splits_force"): is the
rule of gemm.hip:tn_splits (one round of items on the device) where the time is shortest?  (development aid; run on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctc_pytorch_amd import ops
ops.set_precision(1)
dev = torch.device("cuda", 0)
shapes = [(1536, 1024, 76800, 3072, 1024), (1536, 512, 76736, 3072, 1024), (1280, 640, 25600, 2560, 640), (1280, 320, 25568, 2560, 640)]
for M, N, K, lda, ldb in shapes:
    A = torch.randn(K, lda, device=dev)
    B = torch.randn(K, ldb, device=dev)
    C = torch.empty(M, N, device=dev)
    line = []
    for f in (0, 2, 4, 5, 8, 10, 12, 16, 21, 24, 32):
        ops.set_option("tn_splits_force", f)
        for _ in range(2):
            ops.gemm(1, 0, M, N, K, A, lda, B, ldb, C, N)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm(1, 0, M, N, K, A, lda, B, ldb, C, N)
        e1.record()
        torch.cuda.synchronize()
        line.append("%s %.0f" % ("auto" if f == 0 else "s=%d" % f, e0.elapsed_time(e1) / 10 * 1e3))
    ops.set_option("tn_splits_force", 0)
    print("%5d x %5d x %6d (lda %d, ldb %d) us:  %s" % (M, N, K, lda, ldb, "  ".join(line)), flush=True)
