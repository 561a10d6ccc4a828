"""Generic beam kernel with 256 / 1 024 threads per utterance at several (V, W) (development aid): kernel time of a 128-utterance batch."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_pytorch_amd import ops
from ctc_pytorch_amd.testing import synth
dev = torch.device("cuda", 0)
T, B = 800, 128
for V, W in ((62, 20), (62, 60), (200, 20), (200, 60), (200, 200), (41, 200)):
    rs = np.random.RandomState(V + W)
    tab = torch.as_tensor(-3.0 * rs.random_sample((V + 1, V + 1)), dtype=torch.float64).to(dev)
    for regime in ("peaky", "flat"):
        x = torch.from_numpy(synth.make_logprobs(seed=7, T=T, B=B, V=V, regime=regime)).to(dev)
        lens = torch.as_tensor(np.random.RandomState(2).randint(400, 801, size=B), dtype=torch.int32).to(dev)
        row = []
        for fast, nt in ((1, 0), (0, 256), (0, 1024)):
            ops.set_option("beam_fast", fast); ops.set_option("beam_generic_threads", nt)
            ops.beam_decode_device(x, lens, tab, 0.1, W); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = ops.beam_decode_device(x, lens, tab, 0.1, W); e1.record(); torch.cuda.synchronize()
            row.append("%s %8.2f ms" % ("default" if fast else "generic/%d" % nt, e0.elapsed_time(e1)))
        print("V=%3d W=%3d %-5s: %s" % (V, W, regime, " | ".join(row)), flush=True)
ops.set_option("beam_fast", 1); ops.set_option("beam_generic_threads", 0)
