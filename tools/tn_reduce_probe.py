"""TN weight-gradient product + its split-K reduce for a few (M, N, K): run under rocprofv3 --kernel-trace (development aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_pytorch_amd import ops
ops.set_precision(1)
dev = torch.device("cuda", 0)
for (M, N, K) in [(1536, 1024, 76800), (1536, 1000, 76800), (1280, 640, 25600), (1536, 512, 76800), (512, 512, 76800), (1536, 1024, 25600)]:
    A = torch.randn(K, M, device=dev)
    B = torch.randn(K, N, device=dev)
    C = torch.zeros(M, N, device=dev)
    for beta in (0.0, 1.0):
        for _ in range(3):
            ops.gemm(1, 0, M, N, K, A, M, B, N, C, N, beta=beta)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm(1, 0, M, N, K, A, M, B, N, C, N, beta=1.0)
    e1.record(); torch.cuda.synchronize()
    print("TN %5d x %5d x %6d: %8.1f us per product (kernel + reduce), %6.1f TFLOP/s" % (M, N, K, e0.elapsed_time(e1) * 100, 2.0 * M * N * K / (e0.elapsed_time(e1) * 100) / 1e6), flush=True)
