#!/usr/bin/env python3
"""The roofline_gemm probe shape of bench.py (25 600 x 1 280 x 640, precision 1) alone, for a rocprofv3 --pmc pass:
   cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <out> -o f -- python tools/pmc_gemm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_pytorch_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
ops.set_precision(1)
M, N, K = 25600, 1280, 640
A = torch.randn(M, K, device=dev)
W = torch.randn(N, K, device=dev)
C = torch.empty(M, N, device=dev)
for _ in range(6):
    ops.gemm(0, 1, M, N, K, A, K, W, K, C, N)
torch.cuda.synchronize()
