"""Main-loop anatomy of the 256x256 plane tile (development aid): gemm_dbg bit 0 = no C stores, bit 1 = no DMA issue (stale LDS), bit 2 = nontemporal C stores."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctc_pytorch_amd import ops
ops.set_precision(1)
dev = torch.device("cuda", 0)
ops.set_option("gemm_a_inline", 0)
for M, N, K in ((76800, 3072, 1024), (25600, 2560, 640)):
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
    for pp, dbg in ((0, 0), (1, 0), (1, 1), (1, 8), (1, 9)):
        ops.set_option("gemm_pingpong", pp)
        ops.set_option("gemm_dbg", dbg)
        for _ in range(3): ops.gemm(0, 1, M, N, K, A, K, B, K, C, N)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gemm(0, 1, M, N, K, A, K, B, K, C, N)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        print("%6d x %5d x %5d pp=%d dbg=%d  %8.1f us  %7.1f TFLOP/s eff (incl. split passes)" % (M, N, K, pp, dbg, us, 2.0 * M * N * K / us / 1e6), flush=True)
ops.set_option("gemm_dbg", 0)
