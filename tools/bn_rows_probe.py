"""BatchNorm over (T*B, C) rows, forward (training) + backward, with the dword column-sum kernel (option bn_rows4 = 0) and the 16-B one (1):
us per forward / backward call and the largest difference of the outputs (development aid; run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctc_pytorch_amd import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for rows, C, what in ((25600, 640, "cfg2"), (76800, 1024, "cfg4"), (6400, 768, "ref_yaml"), (25600, 62, "odd C: dword kernel either way")):
    x0 = torch.randn(rows, C, device=dev) * 0.7 + 0.1
    gy = torch.randn(rows, C, device=dev)
    g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    res = {}
    for opt in (0, 1):
        ops.set_option("bn_rows4", opt)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        tf = tb = 0.0
        for it in range(12):
            x = x0.clone().requires_grad_(True)
            gg, bb = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
            y = ops.batch_norm(x, gg, bb, rm, rv, rows, C, 1, True)
            e[1].record()
            e[2].record()
            y.backward(gy)
            e[3].record()
            torch.cuda.synchronize()
            if it >= 2:
                tf += e[0].elapsed_time(e[1]) * 100; tb += e[2].elapsed_time(e[3]) * 100
        res[opt] = (tf, tb, y.detach(), x.grad, gg.grad, bb.grad, rm, rv)
    ops.set_option("bn_rows4", 1)
    d = [float((res[0][i] - res[1][i]).abs().max()) for i in range(2, 8)]
    print("%-32s %6d x %4d  fwd %6.1f -> %6.1f us  bwd %6.1f -> %6.1f us   max |diff| y %.1e dx %.1e dgamma %.1e dbeta %.1e run_mean %.1e run_var %.1e"
          % (what, rows, C, res[0][0], res[1][0], res[0][1], res[1][1], *d), flush=True)
