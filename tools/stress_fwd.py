"""Determinism stress of the forward path (development aid): the same model / batch forward N times; every run must reproduce the
first one bit for bit.  python tools/stress_fwd.py [cfg4|cfg2] [runs]"""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_kernels as TG
from ctc_pytorch_amd import ops
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
ops.set_precision(1)
m, b, c = TG._full_size_model(name, dev)
x = TG.gpu(b["x"], dev)
ref = None
bad = 0
for i in range(runs):
    with torch.no_grad():
        lp = m(x)
    torch.cuda.synchronize()
    ops.check_health()
    if ref is None:
        ref = lp.clone()
    else:
        d = float((lp - ref).abs().max())
        nan = bool(torch.isnan(lp).any())
        if d != 0.0 or nan:
            bad += 1
            print("run %d differs: max abs %.3e nan=%s" % (i, d, nan), flush=True)
print("%s: %d runs, %d differing (CTCN_FWD_OVERLAP=%s)" % (name, runs, bad, os.environ.get("CTCN_FWD_OVERLAP", "1")))
