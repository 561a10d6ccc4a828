"""Time the weight-gradient products C = A^T B (both operands contraction-major) on the TN tile and on the plane path (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctc_pytorch_amd import ops
ops.set_precision(1)
dev = torch.device("cuda", 0)
shapes = [(1280, 640, 25600), (1280, 320, 25568), (1536, 1024, 76800), (1536, 512, 76736), (1280, 40, 25600), (62, 640, 25600)]
for M, N, K in shapes:
    A = torch.randn(K, M, device=dev)
    B = torch.randn(K, N, device=dev)
    C = torch.empty(M, N, device=dev)
    for tn in (0, 1):
        ops.set_option("gemm_tn", tn)
        for _ in range(3):
            ops.gemm(1, 0, M, N, K, A, M, B, N, C, N)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(1, 0, M, N, K, A, M, B, N, C, N)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print("%5d x %5d x %6d  gemm_tn=%d  %8.1f us  %7.1f TFLOP/s (all passes)" % (M, N, K, tn, us, 2.0 * M * N * K / us / 1e6), flush=True)
ops.set_option("gemm_tn", 1)
