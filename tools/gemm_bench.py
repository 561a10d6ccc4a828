"""Time the bf16x3 plane GEMM on the shapes of a cfg2 step with both workgroup tilings (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctc_pytorch_amd import ops
ops.set_precision(1)
dev = torch.device("cuda", 0)
shapes = [(25600, 1280, 640, 0, 1), (25600, 2560, 640, 0, 1), (25600, 640, 2560, 0, 0), (25600, 2560, 40, 0, 1), (76800, 3072, 1024, 0, 1)]
for M, N, K, ta, tb in shapes:
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    for t256, dbg in ((1, 1), (1, 4), (1, 5)):
        ops.set_option("gemm_tile256", t256)
        ops.set_option("gemm_pingpong", 1 if dbg in (1, 5) else 0)
        ops.set_option("gemm_a_inline", 1 if dbg in (4, 5) else 0)
        for _ in range(3):
            ops.gemm(ta, tb, M, N, K, A, A.shape[1], B, B.shape[1], C, N)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(ta, tb, M, N, K, A, A.shape[1], B, B.shape[1], C, N)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print("%6d x %5d x %5d  tile256=%d variant=%d (1: ping-pong planes, 4: inline-A, 5: inline-A ping-pong)  %8.1f us  %7.1f TFLOP/s (incl. split passes)" % (M, N, K, t256, dbg, us, 2.0 * M * N * K / us / 1e6))
ops.set_option("gemm_tile256", 1)
ops.set_option("gemm_pingpong", 1)
ops.set_option("gemm_a_inline", 1)
