"""A few launches of the three bf16x3 GEMM tiles of a cfg2 step for an SQ counter pass (development aid):
   rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -- python tools/gemm_pmc_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctc_pytorch_amd import ops
ops.set_precision(1)
dev = torch.device("cuda", 0)
# TN tile (weight gradients), both widths
for M, N, K in [(1280, 640, 25600), (1536, 1024, 76800)]:
    A, B, C = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev), torch.empty(M, N, device=dev)
    for _ in range(3):
        ops.gemm(1, 0, M, N, K, A, M, B, N, C, N)
# float32-A ping-pong tile: dx of a layer (NN) and the input projection (NT)
M, N, K = 25600, 640, 2560
A, B, C = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev), torch.empty(M, N, device=dev)
for _ in range(3):
    ops.gemm(0, 0, M, N, K, A, K, B, N, C, N)
M, N, K = 25600, 2560, 640
A, B, C = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
for _ in range(3):
    ops.gemm(0, 1, M, N, K, A, K, B, K, C, N)
torch.cuda.synchronize()
