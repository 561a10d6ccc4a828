"""Does any kernel of a training step read memory it (or an earlier kernel of the step) did not write?  The caching allocator hands out
blocks with whatever their last owner left in them; here its pool is filled with a poison pattern before each run -- NaN, then 1e-3, then
-7.5 -- and the loss trajectories of the runs must be bit-identical and finite.  usage: uninit_probe.py <workload> <steps>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import squat_stress
wl, steps = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda", 0)


def poison(value):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    blocks = []
    for sz in (1 << 30, 256 << 20, 64 << 20, 16 << 20, 2 << 20, 1 << 20, 256 << 10, 32 << 10, 4 << 10):
        n = 24 if sz >= (256 << 20) else 48
        for _ in range(n):
            if sum(b.numel() * 4 for b in blocks) + sz > 0.5 * free:
                break
            blocks.append(torch.full((sz // 4,), value, dtype=torch.float32, device=dev))
    got = sum(b.numel() * 4 for b in blocks)
    del blocks
    torch.cuda.synchronize()
    return got


ref = None
for value in (float("nan"), 1e-3, -7.5, float("nan")):
    nbytes = poison(value)
    r = squat_stress.run(wl, steps, squat=False, dev=dev)
    ok = all(x == x for x in r["losses"])
    if ref is None:
        ref = r["losses"]
    print("%s poisoned %.1f GB with %r: finite %s, equal to the first run %s  (last loss %r)" % (wl, nbytes / 2**30, value, ok, r["losses"] == ref, r["losses"][-1]), flush=True)
