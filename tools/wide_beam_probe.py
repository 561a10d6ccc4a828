"""Reference-default beam (W = 200, alpha 0.01) on the cfg5 batch: utterances/s with NS searches in flight for the generic kernel's
thread count (option beam_generic_threads) and candidate-table placement (beam_cand_global).  A 1 024-thread search owns a CU (116 VGPRs:
four waves per SIMD); two 512-thread searches share one when their LDS fits twice (candidate table in L2, LM table in LDS: 77 KB each).
Labellings of every configuration are compared with the shipped one's.     python tools/wide_beam_probe.py [W]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_pytorch_amd import ops
from ctc_pytorch_amd.testing import synth
from ctc_pytorch_amd.utils.NgramLM import LanguageModel

W = int(sys.argv[1]) if len(sys.argv) > 1 else 200
CONFIGS = [tuple(int(v) for v in c.split(":")) for c in os.environ.get("WB_CONFIGS", "0:0,1024:1,512:0,512:1,256:1").split(",")]
NSS = [int(v) for v in os.environ.get("WB_NS", "3,4,6,8").split(",")]
REPS = int(os.environ.get("WB_REPS", "6"))
V, T, B = 62, 800, 128
dev = torch.device("cuda", 0)
i2c = synth.int2char(V)
tab = LanguageModel(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "lm_phone_bg.arpa")).table([i2c[i] for i in range(V)])
tab_dev = torch.as_tensor(tab, dtype=torch.float64).to(dev)
phones = [i2c[i] for i in range(V)]
for regime in ("peaky", "flat"):
    lp = synth.make_logprobs(seed=7, T=T, B=B, V=V, regime=regime)
    lens = list(np.random.RandomState(2).randint(400, 801, size=B))
    x = torch.from_numpy(lp).to(dev)
    lens_dev = torch.as_tensor(lens, dtype=torch.int32).to(dev)
    base = None
    for threads, cg in CONFIGS:
        ops.set_option("beam_generic_threads", threads)
        ops.set_option("beam_cand_global", cg)
        try:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops.beam_decode_device(x, lens_dev, tab_dev, 0.01, W)
            torch.cuda.synchronize()
            e0.record()
            out = ops.beam_decode_device(x, lens_dev, tab_dev, 0.01, W)
            e1.record()
            torch.cuda.synchronize()
            ids = [out[0][k, : int(out[1][k])].cpu().tolist() for k in range(B)]
            if base is None:
                base = ids
            row = "%-5s W=%d threads=%4d cand_global=%d  one batch %7.2f ms  same=%s |" % (regime, W, threads, cg, e0.elapsed_time(e1), ids == base)
            for NS in NSS:
                streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
                nfl = REPS * NS
                warm = []
                for k in range(NS):
                    with torch.cuda.stream(streams[k]):
                        warm.append(ops.beam_decode_async(x, lens_dev, tab_dev, 0.01, W))
                for h in warm:
                    h.result()
                del warm
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                pend = []
                for k in range(nfl):
                    with torch.cuda.stream(streams[k % NS]):
                        pend.append(ops.beam_decode_async(x, lens_dev, tab_dev, 0.01, W))
                    if len(pend) == 2 * NS:
                        pend.pop(0).strings(phones, " ")
                for h in pend:
                    h.strings(phones, " ")
                dt = (time.perf_counter() - t0) / nfl
                row += "  NS=%d %6.0f utt/s" % (NS, B / dt)
            print(row, flush=True)
        finally:
            ops.set_option("beam_generic_threads", 0)
            ops.set_option("beam_cand_global", 0)
