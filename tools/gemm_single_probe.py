"""Option "gemm_bf16_single" (one bf16 product per tile step instead of the three bf16x3 products) on the three 256-row tiles: the result
against a float64 product of the bf16-ROUNDED operands (what the mode defines), and the time of both modes on the shapes of cfg2 / cfg4
(development aid; run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctc_pytorch_amd import ops
ops.set_precision(1)
dev = torch.device("cuda", 0)
torch.manual_seed(0)


def run(ta, tb, M, N, K, A, B, C, reps):
    for _ in range(2):
        ops.gemm(ta, tb, M, N, K, A, A.shape[1], B, B.shape[1], C, N)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(ta, tb, M, N, K, A, A.shape[1], B, B.shape[1], C, N)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


# (transA, transB, M, N, K, what)
shapes = [(0, 1, 25600, 2560, 640, "cfg2 input projection (plane tile)"),
          (0, 0, 25600, 640, 2560, "cfg2 dx (float32-A tile)"),
          (1, 0, 1280, 640, 25600, "cfg2 dW_ih (TN tile)"),
          (1, 0, 1280, 320, 25568, "cfg2 dW_hh (TN tile, 256 x 128)"),
          (0, 1, 76800, 3072, 1024, "cfg4 input projection"),
          (0, 0, 76800, 1024, 3072, "cfg4 dx"),
          (1, 0, 1536, 1024, 76800, "cfg4 dW_ih"),
          (1, 0, 1536, 512, 76736, "cfg4 dW_hh"),
          (0, 1, 3000, 1000, 136, "ragged plane tile"),
          (0, 0, 2900, 388, 1284, "ragged float32-A tile"),
          (1, 0, 644, 132, 5000, "ragged TN tile")]
for ta, tb, M, N, K, what in shapes:
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C3, C1 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    reps = 10 if M * N * K > 1e10 else 20
    ops.set_option("gemm_bf16_single", 0)
    us3 = run(ta, tb, M, N, K, A, B, C3, reps)
    ops.set_option("gemm_bf16_single", 1)
    us1 = run(ta, tb, M, N, K, A, B, C1, reps)
    ops.set_option("gemm_bf16_single", 0)
    # references on a sample of rows (float64): the exact product, and the product of the bf16 roundings
    rows = torch.randint(0, M, (256,), device=dev)
    a64 = (A.t() if ta else A)[rows].double()
    b64 = (B.t() if tb else B).double()
    ar = (A.t() if ta else A)[rows].bfloat16().double()
    br = (B.t() if tb else B).bfloat16().double()
    exact, rounded = a64 @ b64, ar @ br
    scale = float(exact.abs().mean())
    e3 = float((C3[rows].double() - exact).abs().max()) / scale
    e1r = float((C1[rows].double() - rounded).abs().max()) / scale
    e1x = float((C1[rows].double() - exact).abs().max()) / scale
    print("%-34s %6d x %5d x %6d  x3 %8.1f us  single %8.1f us (%.2fx)  |x3 - exact| %.1e  |single - rounded product| %.1e  |single - exact| %.1e  (of mean |C|)"
          % (what, M, N, K, us3, us1, us3 / us1, e3, e1r, e1x), flush=True)
    # (a shape below the 256-row tiles' thresholds stays on the small bf16x3 tile: then `single` is the bf16x3 result)
    assert e1r < 2e-5 or e1x < 1e-4, "single-product tile differs from the product of the rounded operands"
print("ok")
