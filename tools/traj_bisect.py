"""Where does a training trajectory stop being reproducible?  `n` traced runs of tools/squat_stress.run (same seeds, same dropout stream) in ONE
process; every run that differs from run 0 is reported with the FIRST traced tensor that differs: (step, name) in execution order -- forward
outputs of the recurrent layers and the head, the loss, gradients reaching each layer (backward order), the flat gradient, the flat
parameters after Adam.  usage: traj_bisect.py <workload> <steps> <n> [squat] [prelude-workload prelude-steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import squat_stress
from ctc_pytorch_amd import ops
wl, steps, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
squat = len(sys.argv) > 4 and sys.argv[4] == "squat"
dev = torch.device("cuda", 0)
if len(sys.argv) > 6:
    squat_stress.run(sys.argv[5], int(sys.argv[6]), squat=True, seed=3, dev=dev)
    print("prelude: %s x %s steps with squatters" % (sys.argv[5], sys.argv[6]), flush=True)


def first_difference(a, b):
    step = 0
    for (na, va), (nb, vb) in zip(a, b):
        assert na == nb
        if na == "end-of-step":
            step += 1
        elif va != vb:
            return step, na
    return None


base = squat_stress.run(wl, steps, squat=False, dev=dev, trace=True)
bad = 0
for i in range(1, n + 1):
    r = squat_stress.run(wl, steps, squat=squat and i % 2 == 1, seed=i, dev=dev, trace=True)
    d = first_difference(r["trace"], base["trace"])
    if d is not None or r["losses"] != base["losses"]:
        bad += 1
        names = [nm for nm, _ in base["trace"]]
        print("run %d (squat %s): first difference at step %s in %r; losses differ from step %s" % (
            i, squat and i % 2 == 1, d[0] if d else None, d[1] if d else None,
            next((k for k, (x, y) in enumerate(zip(r["losses"], base["losses"])) if x != y), None)), flush=True)
    ops.check_health()
print("%s x %d steps: %d of %d runs differ from run 0 (kernels %r)" % (wl, steps, bad, n, base["kernels"]), flush=True)
