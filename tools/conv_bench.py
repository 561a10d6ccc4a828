"""Conv2d kernels of the cfg3 front-end (B=32, T=800, F=40; layers 1->32 s(1,2) and 32->32 s(2,2)), MFMA implicit GEMM vs the
direct kernels: HIP-event time per call of forward and backward.  python tools/conv_bench.py"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_pytorch_amd import ops, _lib

dev = torch.device("cuda", 0)
shapes = [(32, 1, 800, 40, 32, 3, 3, 1, 2, 1, 1), (32, 32, 800, 20, 32, 3, 3, 2, 2, 1, 1)]
res = []
for (B, Ci, Hi, Wi, Co, kh, kw, sh, sw, ph, pw) in shapes:
    x = torch.randn(B, Ci, Hi, Wi, device=dev, requires_grad=True)
    w = (torch.randn(Co, Ci, kh, kw, device=dev) / (Ci * kh * kw) ** 0.5).requires_grad_()
    b = torch.randn(Co, device=dev, requires_grad=True)
    for mfma in (0, 1):
        _lib.lib().ctcn_set_option(b"conv_mfma", mfma)
        y = ops.conv2d(x, w, b, (sh, sw), (ph, pw))
        dy = torch.randn_like(y)
        for _ in range(3):
            y = ops.conv2d(x, w, b, (sh, sw), (ph, pw)); y.backward(dy)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        n = 20
        tf = tb = 0.0
        for _ in range(n):
            e[0].record(); y = ops.conv2d(x, w, b, (sh, sw), (ph, pw)); e[1].record(); y.backward(dy); e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
        flops = 2.0 * y.numel() * Ci * kh * kw
        res.append(dict(shape=[B, Ci, Hi, Wi, Co], mfma=mfma, fwd_us=1e3 * tf / n, bwd_us=1e3 * tb / n, fwd_gflop=flops / 1e9,
                        fwd_bytes_mb=(x.numel() + y.numel()) * 4 / 1e6))
        print(json.dumps(res[-1]), flush=True)
_lib.lib().ctcn_set_option(b"conv_mfma", 1)
