"""TN weight-gradient product with padded leading dimensions: do power-of-two row strides cost anything?  (development aid)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_pytorch_amd import ops
ops.set_precision(1)
dev = torch.device("cuda", 0)
for (M, N, K, pa, pb) in [(3072, 1024, 76800, 0, 0), (3072, 1024, 76800, 0, 0), (3072, 1024, 76800, 0, 32), (3072, 1024, 76800, 32, 32), (3072, 1024, 76800, 32, 0),
                          (1536, 512, 76800, 0, 0), (1536, 512, 76800, 32, 32), (3072, 1024, 76800, 0, 0)]:
    A = torch.randn(K, M + pa, device=dev)
    B = torch.randn(K, N + pb, device=dev)
    C = torch.zeros(M, N, device=dev)
    for _ in range(3):
        ops.gemm(1, 0, M, N, K, A, M + pa, B, N + pb, C, N, beta=1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm(1, 0, M, N, K, A, M + pa, B, N + pb, C, N, beta=1.0)
    e1.record(); torch.cuda.synchronize()
    print("TN %5d x %5d x %6d lda %5d ldb %5d: %8.1f us, %6.1f TFLOP/s" % (M, N, K, M + pa, N + pb, e0.elapsed_time(e1) * 100, 2.0 * M * N * K / (e0.elapsed_time(e1) * 100) / 1e6), flush=True)
