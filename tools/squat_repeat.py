"""tools/squat_stress.py's comparison many times in one process: the undisturbed loss trajectory of a workload against `n` runs with squatter
kernels launched at random points (a different seed each); prints the runs that differ.  usage: squat_repeat.py <workload> <steps> <n>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import squat_stress
from ctc_pytorch_amd import ops
wl, steps, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
base = squat_stress.run(wl, steps, squat=False, dev=dev)
again = squat_stress.run(wl, steps, squat=False, dev=dev)
print("xcd_interleave", ops.get_option("xcd_interleave"), "undisturbed twice equal:", base["losses"] == again["losses"], flush=True)
bad = 0
for seed in range(1, n + 1):
    hit = squat_stress.run(wl, steps, squat=True, seed=seed, dev=dev)
    if hit["losses"] != base["losses"]:
        bad += 1
        first = [i for i, (a, b) in enumerate(zip(hit["losses"], base["losses"])) if a != b][0]
        print("seed %d: differs from step %d on (%r vs %r), %d squats" % (seed, first, hit["losses"][first], base["losses"][first], hit["squats"]), flush=True)
    ops.check_health()
print("%s: %d of %d squatted runs differ from the undisturbed one" % (wl, bad, n))
