"""Can a kernel read, right behind a kernel boundary, rows that the kernel in front of it wrote LAST -- and see something else?  (Round 6, DESIGN.md
section 8 item 1: the reverse direction of cfg4's bottom layer started from other pre-activations than its projection GEMM had produced, in 3 of 10
full parity suites and never in a fresh process.)

The pattern, without the recurrence: the bottom layer's projection of cfg4 (76 800 x 3 072 x 40, one product over rows in ascending time: 943 MB in
~340 us) writes `gates`; a small fill follows (the hand-off tiles' memset); then a copy kernel reads the LAST 64 rows (frame T - 1, what the reverse
direction starts on) and the FIRST 64 rows (frame 0) at once -- `early` -- and again after a device synchronisation -- `late`.  Every iteration scales
the input, so stale rows hold other values than fresh ones.  Any element where early != late is a read that did not see the preceding kernel's write.
`--lag S` sleeps S seconds on the host before every iteration (a host that falls behind lets the GPU go idle between steps, which is what the parity
suite's process does and a benchmark loop does not); `--order 1` issues the projection as [T/2, T) then [0, T/2) (the fix).

usage: stale_probe.py [--iters 400] [--lag 0.0] [--order 0] [--dirty GB]   (--dirty: allocate and free that much first, in odd sizes)
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--lag", type=float, default=0.0)
    ap.add_argument("--order", type=int, default=0)
    ap.add_argument("--dirty", type=float, default=0.0)
    a = ap.parse_args()
    from ctc_pytorch_amd import ops
    dev = torch.device("cuda", 0)
    ops.set_precision(1)
    T, B, I, GH2 = 1200, 64, 40, 3072
    if a.dirty > 0:
        g = torch.Generator(device="cpu").manual_seed(3)
        junk, total = [], 0
        while total < a.dirty * 2 ** 30:
            n = int(torch.randint(1 << 18, 1 << 27, (1,), generator=g))
            junk.append(torch.full((n,), float("nan"), device=dev))
            total += 4 * n
        del junk[::2]
        keep = junk            # every second block stays allocated: the pool is fragmented for the run
    torch.manual_seed(1)
    x0 = torch.randn(T * B, I, device=dev)
    w = torch.randn(GH2, I, device=dev) * 0.1
    gates = torch.empty(T * B, GH2, device=dev)
    tile = torch.empty(1 << 17, device=dev)
    rows = B
    bad_last = bad_first = 0
    worst = 0.0
    t0 = time.perf_counter()
    for it in range(a.iters):
        if a.lag > 0:
            time.sleep(a.lag)
        x = x0 * (1.0 + 0.01 * (it % 97))
        if a.order == 0:
            ops.gemm(0, 1, T * B, GH2, I, x, I, w, I, gates, GH2)
        else:
            h = (T // 2) * B
            ops.gemm(0, 1, T * B - h, GH2, I, x[h:], I, w, I, gates[h:], GH2)
            ops.gemm(0, 1, h, GH2, I, x[:h], I, w, I, gates[:h], GH2)
        tile.zero_()                                           # (the memset in front of the persistent launch)
        early_last = gates[-rows:].clone()
        early_first = gates[:rows].clone()
        torch.cuda.synchronize()
        late_last, late_first = gates[-rows:].clone(), gates[:rows].clone()
        nl = int((early_last != late_last).sum())
        nf = int((early_first != late_first).sum())
        if nl or nf:
            worst = max(worst, float((early_last - late_last).abs().max()), float((early_first - late_first).abs().max()))
            print("iteration %d: %d stale elements in the last rows, %d in the first rows" % (it, nl, nf), flush=True)
        bad_last += nl > 0
        bad_first += nf > 0
    dt = time.perf_counter() - t0
    print("order %d lag %.3f s dirty %.1f GB: %d iterations in %.1f s; iterations with stale LAST rows %d, with stale FIRST rows %d (max |difference| %.3g)" % (
        a.order, a.lag, a.dirty, a.iters, dt, bad_last, bad_first, worst), flush=True)


if __name__ == "__main__":
    main()
