"""Runs the kernels whose roofline is quoted in DESIGN.md a few times each, for rocprofv3 --pmc passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_pytorch_amd import ops
dev = torch.device("cuda:0")
ops.set_precision(int(os.environ.get("CTCN_PRECISION", "1")))
M, K, N = 25600, 640, 1280
A, W, C = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
for _ in range(4):
    ops.gemm(0, 1, M, N, K, A, K, W, K, C, N)
x = torch.randn(M, K, device=dev, requires_grad=True)
g, b = torch.ones(K, device=dev, requires_grad=True), torch.zeros(K, device=dev, requires_grad=True)
rm, rv = torch.zeros(K, device=dev), torch.ones(K, device=dev)
for _ in range(3):
    y = ops.batch_norm(x, g, b, rm, rv, M, K, 1, True)
    y.backward(torch.ones_like(y))
    z = ops.dropout(x, 0.1, True)
# one cfg2 BiLSTM layer (T=800, B=32, I=640, H=320), forward + backward: the persistent recurrent kernels
T, B, H = 800, 32, 320
xr = torch.randn(T, B, 2 * H, device=dev, requires_grad=True)
wr = [(torch.randn(4 * H, 2 * H, device=dev) * 0.05).requires_grad_(True), (torch.randn(4 * H, H, device=dev) * 0.05).requires_grad_(True),
      (torch.randn(4 * H, 2 * H, device=dev) * 0.05).requires_grad_(True), (torch.randn(4 * H, H, device=dev) * 0.05).requires_grad_(True)]
for _ in range(3):
    yr = ops.rnn_layer(xr, wr[0], wr[1], wr[2], wr[3], "lstm")
    yr.backward(torch.ones_like(yr))
torch.cuda.synchronize()
