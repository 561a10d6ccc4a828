"""Does a cfg4 trajectory depend on what persistent launch of the SAME geometry (bidirectional, B = 64, H = 512: every XCD) ran earlier in the process?
(Round 6: the suite's subsets that deviate afterwards all contain such launches at T = 30.)  A prelude of `n` forward + backward passes of one
recurrent layer (gru, T, 64, 24, 512), then 30 traced 12-step cfg4 runs compared with their majority.
usage: prelude_ab.py <T> [n_prelude=6] [kind=gru] [runs=30]"""
import collections, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import squat_stress
from ctc_pytorch_amd import ops

T = int(sys.argv[1]); n_pre = int(sys.argv[2]) if len(sys.argv) > 2 else 6; kind = sys.argv[3] if len(sys.argv) > 3 else "gru"; runs_n = int(sys.argv[4]) if len(sys.argv) > 4 else 30
dev = torch.device("cuda", 0)
ops.set_precision(1)
if T > 0:
    B, I, H = 64, 24, 512
    G = {"lstm": 4, "gru": 3}[kind]
    torch.manual_seed(11)
    x = torch.randn(T, B, I, device=dev)
    w = [torch.randn(G * H, I, device=dev) * 0.2, torch.randn(G * H, H, device=dev) * (1.0 / H ** 0.5), torch.randn(G * H, I, device=dev) * 0.2, torch.randn(G * H, H, device=dev) * (1.0 / H ** 0.5)]
    dy = torch.randn(T, B, 2 * H, device=dev)
    for _ in range(n_pre):
        xs = x.clone().requires_grad_(True)
        ws = [t.clone().requires_grad_(True) for t in w]
        y = ops.rnn_layer(xs, ws[0], ws[1], ws[2], ws[3], kind)
        y.backward(dy)
    torch.cuda.synchronize()
    print("prelude: %d x %s layer T=%d B=64 H=512, kernels %r" % (n_pre, kind, T, ops.rnn_last_kernels()), flush=True)
runs = []
for i in range(runs_n):
    r = squat_stress.run("cfg4", 12, squat=False, seed=i + 1, dev=dev, trace=True)
    runs.append(json.dumps(r["trace"]))
ref = collections.Counter(runs).most_common(1)[0][0]
print("prelude T=%d: %d of %d cfg4 runs deviate from the majority" % (T, sum(1 for j in runs if j != ref), runs_n), flush=True)
