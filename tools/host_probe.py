import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import bench
from ctc_pytorch_amd import nn, ops, parallel
from ctc_pytorch_amd.optim import FlatAdam
from oracle import synth
c = bench.WORKLOADS["cfg1"]
dev = torch.device("cuda", 0)
ops.set_precision(1)
torch.manual_seed(1)
model = bench.build(c, dev, drop_out=0.1).train()
opt = FlatAdam(model, lr=1e-3, weight_decay=5e-4)
batch = synth.make_batch(seed=1, B=c["B"], T=c["T"], F=40, V=c["V"], lab_lo=c["lab"][0], lab_hi=c["lab"][1], full_length=True)
x = torch.from_numpy(batch["x"]).to(dev); tg = torch.from_numpy(batch["targets"]).to(dev); tl = torch.from_numpy(batch["tgt_len"]).to(dev)
loss_fn = nn.CTCLoss(reduction="sum")
in_len = None
def step():
    global in_len
    out = model(x)
    if in_len is None:
        in_len = torch.full((c["B"],), out.size(0), dtype=torch.int64, device=dev)
    loss = loss_fn(out, tg, in_len, tl) / c["B"]
    opt.zero_grad(); loss.backward(); ops.join_side_stream(); parallel.allreduce_grads(opt.grad); opt.step()
    return loss
for blk in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = 0.0
    for _ in range(10):
        h0 = time.perf_counter(); step(); th += time.perf_counter() - h0
    torch.cuda.synchronize()
    print("steps %3d-%3d: %.2f ms/step (host enqueue %.2f ms/step)" % (blk * 10, blk * 10 + 9, (time.perf_counter() - t0) * 100, th * 100), flush=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
