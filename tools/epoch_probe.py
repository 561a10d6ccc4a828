"""Where does steps/train_ctc.run_epoch spend its time beyond the bare training step?  HIP-event timing of the pieces."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ctc_pytorch_amd import ops, nn, parallel, _lib
from ctc_pytorch_amd.optim import FlatAdam
from ctc_pytorch_amd.utils.data_loader import DevicePrefetcher
from ctc_pytorch_amd.steps.train_ctc import run_epoch
from ctc_pytorch_amd.testing import synth

dev = torch.device("cuda", 0)
c = bench.WORKLOADS["cfg2"]
model = bench.build(c, dev, 0.1)
opt = FlatAdam(model, lr=1e-3)
loss_fn = nn.CTCLoss(reduction="sum")
batch = synth.make_batch(seed=0, B=c["B"], T=c["T"], F=40, V=c["V"], lab_lo=c["lab"][0], lab_hi=c["lab"][1])
x = torch.from_numpy(batch["x"]).to(dev); tg = torch.from_numpy(batch["targets"]).to(dev); tl = torch.from_numpy(batch["tgt_len"]).to(dev)
in_len = torch.full((c["B"],), c["T"], dtype=torch.int64, device=dev)

def ev_time(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n, 1e3 * (time.perf_counter() - t0) / n

def step():
    out = model(x)
    loss = loss_fn(out, tg, in_len, tl) / c["B"]
    opt.zero_grad(); loss.backward(); ops.join_side_stream(); opt.step()
    return out, loss
out, loss = step()
def greedy():
    idx = ops.argmax_last(out); ids, l = ops.greedy_collapse(idx, in_len, blank=0); return ops.edit_distance(ids, l, tg, tl)
def argmax(): return ops.argmax_last(out)
idx = ops.argmax_last(out); ids, idl = ops.greedy_collapse(idx, in_len, blank=0)
def collapse(): return ops.greedy_collapse(idx, in_len, blank=0)
def edit(): return ops.edit_distance(ids, idl, tg, tl)
def stats():
    d = greedy()
    h = _lib.status_word(dev).reshape(1).double()
    return torch.cat([torch.stack([loss.detach().double(), d.sum().double(), tl.sum().double()]), h])
hx = torch.from_numpy(batch["x"])
def pin(): return hx.pin_memory().to(dev, non_blocking=True)
hb = (hx, torch.ones(c["B"], dtype=torch.float32), torch.from_numpy(batch["targets"]), torch.from_numpy(batch["tgt_len"]), ["u%d" % i for i in range(c["B"])])
pf = DevicePrefetcher([hb] * 6, dev)
def epoch(): run_epoch(0, model, pf, loss_fn, dev, optimizer=opt, print_every=10 ** 9, is_training=True, global_batch=c["B"], log=lambda *_: None)
def epoch_nopf(): run_epoch(0, model, [(x, torch.ones(c["B"], device=dev), tg, tl, None)] * 6, loss_fn, dev, optimizer=opt, print_every=10 ** 9, is_training=True, global_batch=c["B"], log=lambda *_: None)
for name, fn, div in (("step", step, 1), ("argmax", argmax, 1), ("collapse", collapse, 1), ("edit_distance", edit, 1), ("greedy_all", greedy, 1), ("stats", stats, 1),
                      ("pin+h2d 4MB", pin, 1), ("run_epoch/6 device batches", epoch_nopf, 6), ("run_epoch/6 prefetcher", epoch, 6)):
    g, h = ev_time(fn, 3)
    print("%-30s gpu %.3f ms  wall %.3f ms" % (name, g / div, h / div), flush=True)


# ---- the same loop with Tensor.copy_ (all host cores) as the staging copy: the 8-10 ms per step this tool found ---------------------
orig = DevicePrefetcher._pinned
def torch_copy(slot, key, t):
    buf = slot["bufs"].get(key)
    if buf is None or buf.numel() < t.numel():
        return orig(slot, key, t)
    v = buf[:t.numel()].view(t.shape); v.copy_(t); return v
DevicePrefetcher._pinned = staticmethod(torch_copy)
g, h = ev_time(epoch, 3)
print("%-30s gpu %.3f ms  wall %.3f ms" % ("staging with Tensor.copy_ (%d threads)" % torch.get_num_threads(), g / 6, h / 6), flush=True)
