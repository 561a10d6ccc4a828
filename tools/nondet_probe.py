"""Which tensor of a training step is not a pure function of the step's inputs?  (Round 6: the cfg4 loss trajectory of the squatter test differed
between three runs of ONE process -- gpurun_out/s1 -- with identical process-wide state.)

One model, one batch, NO optimiser step: `reps` forward + backward passes from identical weights, inputs and dropout stream.  Every module
output (forward hooks), every gradient that reaches a module output (tensor hooks) and every parameter gradient is reduced on the device to a
64-bit checksum of its bit pattern; a repetition is compared with repetition 0 and the FIRST tensors that differ are printed in execution
order (forward tensors in module order, then gradients in the order they were produced) -- the kernel that wrote the first one is the suspect.
Between repetitions the probe can dirty what a kernel must not depend on:

    --poison-ws  V    fill the library's persistent workspaces (ops._ws buffers: operand planes, split-K partials, hand-off tiles) with V
    --poison-pool V   fill the caching allocator's free pool with V (as tools/uninit_probe.py)
    --squat           squatter kernels on a third stream at random points (as tools/squat_stress.py)
    V: nan | rand | <float>

usage: nondet_probe.py [--workload cfg4] [--reps 40] [--T n] [--B n] [--poison-ws V] [--poison-pool V] [--squat] [--precision 1] [--drop 0.1]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def checksum(t):
    """64-bit sum of the 32-bit patterns (+ a position-weighted one: a permutation of equal values must not cancel)."""
    v = t.detach().contiguous().view(-1)
    if v.dtype != torch.float32:
        v = v.float()
    bits = v.view(torch.int32).to(torch.int64)
    n = bits.numel()
    w = (torch.arange(n, device=bits.device, dtype=torch.int64) % 8191) + 1
    return (int(bits.sum().item()), int((bits * w).sum().item()))


def fill(buf, how, gen):
    f = buf.view(torch.float32) if buf.dtype == torch.uint8 else buf
    if how == "nan":
        f.fill_(float("nan"))
    elif how == "rand":
        f.copy_(torch.randn(f.shape, device=f.device, generator=gen) * 3.0)
    else:
        f.fill_(float(how))


def poison_pool(dev, how, gen):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    blocks = []
    for sz in (1 << 30, 256 << 20, 64 << 20, 16 << 20, 2 << 20, 1 << 20, 256 << 10, 32 << 10, 4 << 10):
        for _ in range(16 if sz >= (256 << 20) else 32):
            if sum(b.numel() * 4 for b in blocks) + sz > 0.35 * free:
                break
            b = torch.empty(sz // 4, dtype=torch.float32, device=dev)
            fill(b, how, gen)
            blocks.append(b)
    del blocks
    torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg4")
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--T", type=int, default=0)
    ap.add_argument("--B", type=int, default=0)
    ap.add_argument("--precision", type=int, default=1)
    ap.add_argument("--drop", type=float, default=0.1)
    ap.add_argument("--poison-ws", default=None)
    ap.add_argument("--poison-pool", default=None)
    ap.add_argument("--squat", action="store_true")
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import bench
    from ctc_pytorch_amd import _lib, nn, ops, parallel
    from ctc_pytorch_amd.optim import FlatAdam
    from ctc_pytorch_amd.testing import synth
    dev = torch.device("cuda", 0)
    c = dict(bench.WORKLOADS[a.workload])
    if a.T:
        c["T"] = a.T
    if a.B:
        c["B"] = a.B
    ops.set_precision(a.precision)
    parallel.enable_overlap(True)
    torch.manual_seed(1)
    model = bench.build(c, dev, drop_out=a.drop).train()
    opt = FlatAdam(model, lr=1e-3, weight_decay=5e-4)
    batch = synth.make_batch(seed=1, B=c["B"], T=c["T"], F=c.get("F", 40), V=c["V"], lab_lo=c["lab"][0], lab_hi=c["lab"][1], full_length=True)
    x = torch.from_numpy(batch["x"]).to(dev)
    tg, tl = torch.from_numpy(batch["targets"]).to(dev), torch.from_numpy(batch["tgt_len"]).to(dev)
    loss_fn = nn.CTCLoss(reduction="sum")
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    rs = np.random.RandomState(a.seed)
    third = torch.cuda.Stream(device=dev)
    record = []

    def fwd_hook(name):
        def hook(mod, inp, out):
            if torch.is_tensor(out):
                record.append(("fwd " + name, checksum(out)))
                if out.requires_grad:
                    out.register_hook(lambda g, name=name: record.append(("grad-of-output " + name, checksum(g))))
        return hook

    for name, mod in model.named_modules():
        if name:
            mod.register_forward_hook(fwd_hook(name))

    def squat(p=0.5):
        if a.squat and rs.rand() < p:
            ops.diag_squat(int(rs.choice([1, 2, 4, 8, 12])), int(rs.randint(50, 4001)), threads=int(rs.choice([256, 512])),
                           lds_bytes=int(rs.choice([0, 16 << 10, 64 << 10])), stream=third)

    def one_pass():
        del record[:]
        ops._drop_counter[0] = 0
        squat()
        out = model(x)
        in_len = torch.full((c["B"],), out.size(0), dtype=torch.int64, device=dev)
        squat()
        loss = loss_fn(out, tg, in_len, tl) / c["B"]
        opt.zero_grad()
        if a.squat:
            out.register_hook(lambda g: (squat(), g)[1])
        loss.backward()
        squat()
        ops.join_side_stream()
        record.append(("loss", checksum(loss)))
        for n, p in model.named_parameters():
            record.append(("param-grad " + n, checksum(p.grad)))
        torch.cuda.synchronize()
        ops.check_health()
        return list(record), float(loss)

    ref, loss0 = one_pass()
    print("%s T=%d B=%d precision %d dropout %.2f: %d tensors per pass, loss %r, kernels %r" % (a.workload, c["T"], c["B"], a.precision, a.drop, len(ref), loss0,
                                                                                               ops.rnn_last_kernels()), flush=True)
    bad = 0
    for rep in range(1, a.reps + 1):
        if a.poison_ws:
            torch.cuda.synchronize()
            for key, buf in list(_lib._WS.items()):
                fill(buf, a.poison_ws, gen)
        if a.poison_pool:
            poison_pool(dev, a.poison_pool, gen)
        got, loss = one_pass()
        assert [n for n, _ in got] == [n for n, _ in ref], "the passes did not produce the same tensor list"
        diff = [n for (n, v), (_, w) in zip(got, ref) if v != w]
        if diff:
            bad += 1
            print("rep %d: %d of %d tensors differ from rep 0 (loss %r vs %r); first in execution order: %s" % (rep, len(diff), len(ref), loss, loss0, diff[:6]), flush=True)
    print("%s: %d of %d repetitions differ from repetition 0  [poison-ws %s, poison-pool %s, squat %s]" % (a.workload, bad, a.reps, a.poison_ws, a.poison_pool, a.squat), flush=True)


if __name__ == "__main__":
    main()
