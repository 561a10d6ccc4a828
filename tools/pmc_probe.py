"""Runs the kernels whose roofline is quoted in DESIGN.md a few times each, for rocprofv3 --pmc passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_pytorch_amd import ops
dev = torch.device("cuda:0")
ops.set_precision(int(os.environ.get("CTCN_PRECISION", "1")))
if os.environ.get("PMC_PROBE_SET") == "scatter2":
    # round 3: one BiLSTM layer of the reference's shipped YAML shape (T=200 recurrent steps, B=8, I=1952 -> here 768, H=384): rnn_bwd_scatter2
    T, B, H = 200, 8, 384
    xr = torch.randn(T, B, 2 * H, device=dev, requires_grad=True)
    wr = [(torch.randn(4 * H, 2 * H, device=dev) * 0.05).requires_grad_(True), (torch.randn(4 * H, H, device=dev) * 0.05).requires_grad_(True),
          (torch.randn(4 * H, 2 * H, device=dev) * 0.05).requires_grad_(True), (torch.randn(4 * H, H, device=dev) * 0.05).requires_grad_(True)]
    for _ in range(3):
        yr = ops.rnn_layer(xr, wr[0], wr[1], wr[2], wr[3], "lstm")
        yr.backward(torch.ones_like(yr))
    torch.cuda.synchronize()
    print(ops.rnn_last_kernels())
    sys.exit(0)
if os.environ.get("PMC_PROBE_SET") == "recurrence":
    # bench.py's in-run traffic measurement: ONE layer of the benchmarked workload, forward + backward (the persistent recurrences), nothing else
    T, B, H = int(os.environ.get("PMC_T", "800")), int(os.environ.get("PMC_B", "32")), int(os.environ.get("PMC_H", "320"))
    cell = os.environ.get("PMC_CELL", "lstm")
    G = 4 if cell == "lstm" else 3
    xr = torch.randn(T, B, 2 * H, device=dev, requires_grad=True)
    wr = [(torch.randn(G * H, 2 * H, device=dev) * 0.05).requires_grad_(True), (torch.randn(G * H, H, device=dev) * 0.05).requires_grad_(True),
          (torch.randn(G * H, 2 * H, device=dev) * 0.05).requires_grad_(True), (torch.randn(G * H, H, device=dev) * 0.05).requires_grad_(True)]
    for _ in range(3):
        yr = ops.rnn_layer(xr, wr[0], wr[1], wr[2], wr[3], cell)
        yr.backward(torch.ones_like(yr))
    torch.cuda.synchronize()
    print(ops.rnn_last_kernels())
    sys.exit(0)
M, K, N = 25600, 640, 1280
A, W, C = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
for _ in range(4):
    ops.gemm(0, 1, M, N, K, A, K, W, K, C, N)
x = torch.randn(M, K, device=dev, requires_grad=True)
g, b = torch.ones(K, device=dev, requires_grad=True), torch.zeros(K, device=dev, requires_grad=True)
rm, rv = torch.zeros(K, device=dev), torch.ones(K, device=dev)
for _ in range(3):
    y = ops.batch_norm(x, g, b, rm, rv, M, K, 1, True)
    y.backward(torch.ones_like(y))
    z = ops.dropout(x, 0.1, True)
# one cfg2 BiLSTM layer (T=800, B=32, I=640, H=320), forward + backward: the persistent recurrent kernels
T, B, H = 800, 32, 320
xr = torch.randn(T, B, 2 * H, device=dev, requires_grad=True)
wr = [(torch.randn(4 * H, 2 * H, device=dev) * 0.05).requires_grad_(True), (torch.randn(4 * H, H, device=dev) * 0.05).requires_grad_(True),
      (torch.randn(4 * H, 2 * H, device=dev) * 0.05).requires_grad_(True), (torch.randn(4 * H, H, device=dev) * 0.05).requires_grad_(True)]
for _ in range(3):
    yr = ops.rnn_layer(xr, wr[0], wr[1], wr[2], wr[3], "lstm")
    yr.backward(torch.ones_like(yr))
torch.cuda.synchronize()
# cfg3 front-end, layer 2 (32 -> 32, 3x3, stride 2x2): MFMA implicit-GEMM forward / dgrad / wgrad
xc = torch.randn(32, 32, 800, 20, device=dev, requires_grad=True)
wc = (torch.randn(32, 32, 3, 3, device=dev) / 17.0).requires_grad_(True)
bc = torch.zeros(32, device=dev, requires_grad=True)
for _ in range(2):
    yc = ops.conv2d(xc, wc, bc, (2, 2), (1, 1))
    yc.backward(torch.ones_like(yc))
torch.cuda.synchronize()
# cfg5 beam decode (peaky regime), two launches: beam_prep_kernel + beam_fast_kernel
import numpy as np
from ctc_pytorch_amd.utils.NgramLM import LanguageModel
from ctc_pytorch_amd.testing import synth
Vb, Tb, Bb, Wb = 62, 800, 128, 20
i2c = synth.int2char(Vb)
tab = LanguageModel(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lm_phone_bg.arpa")).table([i2c[i] for i in range(Vb)])
lp = torch.from_numpy(synth.make_logprobs(seed=7, T=Tb, B=Bb, V=Vb, regime="peaky")).to(dev)
lens = list(np.random.RandomState(2).randint(400, 801, size=Bb))
for _ in range(2):
    ops.beam_decode(lp, lens, tab, 0.1, Wb)
torch.cuda.synchronize()
