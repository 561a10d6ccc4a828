"""A few launches of the TN tile for a PMC pass (development aid): rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/gemm_tn_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctc_pytorch_amd import ops
ops.set_precision(1)
dev = torch.device("cuda", 0)
for M, N, K in [(1280, 640, 25600), (1536, 1024, 76800)]:
    A = torch.randn(K, M, device=dev)
    B = torch.randn(K, N, device=dev)
    C = torch.empty(M, N, device=dev)
    for _ in range(4):
        ops.gemm(1, 0, M, N, K, A, M, B, N, C, N)
    torch.cuda.synchronize()
