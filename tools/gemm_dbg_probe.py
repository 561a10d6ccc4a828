"""Where does the 256-row plane tile spend its time?  gemm_dbg bit 0 = no C stores, bit 3 = no vmcnt wait in front of the stage barrier (results invalid).  Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctc_pytorch_amd import ops
ops.set_precision(1)
dev = torch.device("cuda", 0)
for M, N, K in [(25600, 2560, 640), (76800, 3072, 1024), (25600, 1280, 640)]:
    A = torch.randn(M, K, device=dev)
    B = torch.randn(N, K, device=dev)
    C = torch.empty(M, N, device=dev)
    ref = None
    for dbg in (0, 1, 8, 9):
        ops.set_option("gemm_dbg", dbg)
        for _ in range(3):
            ops.gemm(0, 1, M, N, K, A, K, B, K, C, N)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(0, 1, M, N, K, A, K, B, K, C, N)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        if dbg == 0: ref = C.clone()
        print("%6d x %5d x %5d  gemm_dbg=%d (%s)  %8.1f us  %7.1f TFLOP/s algorithmic (incl. split passes)" % (
            M, N, K, dbg, {0: "as shipped", 1: "no C stores", 8: "no vmcnt wait before the barrier (wrong results)", 9: "8 + no C stores"}[dbg] + ("" if dbg & 1 else (" equal=%s" % bool(torch.equal(C, ref)) if ref is not None else "")), us, 2.0 * M * N * K / us / 1e6))
    ops.set_option("gemm_dbg", 0)
