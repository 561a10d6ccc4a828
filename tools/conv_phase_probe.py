"""Where a conv_mfma_kernel launch spends its time (development aid): the forward convolutions of cfg3 / the shipped YAML with phases
switched off (option conv_dbg: 1 = no window load, 2 = no MFMA loop, 4 = no output phase; results invalid).  HIP-event time per launch of ctcn_conv2d_fwd, no autograd.  python tools/conv_phase_probe.py"""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_pytorch_amd import ops, _lib
dev = torch.device("cuda", 0)
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr())
for (B, Ci, Hi, Wi, Co, sh, sw) in [(32, 1, 800, 40, 32, 1, 2), (32, 32, 800, 20, 32, 2, 2), (8, 1, 400, 243, 32, 1, 2), (8, 32, 400, 122, 32, 2, 2)]:
    x = torch.randn(B, Ci, Hi, Wi, device=dev)
    w = torch.randn(Co, Ci, 3, 3, device=dev) / (Ci * 9) ** 0.5
    b = torch.randn(Co, device=dev)
    Ho, Wo = (Hi + 2 - 3) // sh + 1, (Wi + 2 - 3) // sw + 1
    y = torch.empty(B, Co, Ho, Wo, device=dev)
    for pf in (0,):
        row = []
        for dbg in (0, 1, 2, 4, 3, 5, 6, 7):
            L.ctcn_set_option(b"conv_dbg", dbg)
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(3):
                L.ctcn_conv2d_fwd(P(x), P(w), P(b), P(y), B, Ci, Hi, Wi, Co, 3, 3, sh, sw, 1, 1, st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                L.ctcn_conv2d_fwd(P(x), P(w), P(b), P(y), B, Ci, Hi, Wi, Co, 3, 3, sh, sw, 1, 1, st)
            e1.record(); torch.cuda.synchronize()
            row.append("dbg%d %6.1f" % (dbg, e0.elapsed_time(e1) * 50))
        L.ctcn_set_option(b"conv_dbg", 0)
        print("x (%d,%d,%d,%d) stride (%d,%d) us per launch: %s" % (B, Ci, Hi, Wi, sh, sw, " | ".join(row)), flush=True)
