import os, sys
import numpy as np, torch, torch.nn as tnn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_pytorch_amd import nn, ops
from ctc_pytorch_amd.models.model_ctc import CTC_Model
from oracle import torch_cpu, np_ref as R
from ctc_pytorch_amd.testing import synth
dev = torch.device("cuda:0")
z = np.load("tests/golden/model_cnn_lstm2x16.npz")
cp = {"batch_norm": True, "activate_function": nn.ReLU, "layer": [[(1, 32), (3, 3), (1, 2), (1, 1), None], [(32, 32), (3, 3), (2, 2), (1, 1), None]]}
base = {"rnn_input_size": 40, "bidirectional": True, "batch_norm": True, "rnn_layers": 2, "rnn_hidden_size": 16}
def mk():
    m = CTC_Model(add_cnn=True, cnn_param=cp, rnn_param=dict(base, rnn_type=nn.LSTM), num_class=62, drop_out=0.0)
    vals = synth.fill_state_dict([(k, tuple(v.shape)) for k, v in m.state_dict().items()], seed=int(z["seed_w"]))
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in vals.items()})
    return m.to(dev).train()
ref = torch_cpu.TorchCpuCTCModel(add_cnn=True, cnn_param=dict(cp, activate_function=tnn.ReLU), rnn_param=dict(base, rnn_type=tnn.LSTM), num_class=62, drop_out=0.0)
vals = synth.fill_state_dict([(k, tuple(v.shape)) for k, v in ref.state_dict().items()], seed=int(z["seed_w"]))
ref.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in vals.items()}); ref.train()
x, tg, tl = torch.from_numpy(z["x"]), torch.from_numpy(z["targets"]), torch.from_numpy(z["tgt_len"])
models = {0: mk(), 1: mk()}
opts = {k: torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-4) for k, m in models.items()}
oref = torch.optim.Adam(ref.parameters(), lr=1e-3, weight_decay=5e-4)
for step in range(3):
    lp = ref(x); il = (torch.from_numpy(z["frac"]) * lp.size(0)).long()
    loss = tnn.CTCLoss(reduction="sum")(lp, tg, il, tl) / x.shape[0]
    oref.zero_grad(); loss.backward()
    gref = {k: p.grad.clone() for k, p in ref.named_parameters()}
    oref.step()
    for flag, m in models.items():
        ops.set_rnn_persistent(flag)
        out = m(x.to(dev))
        l = nn.CTCLoss(reduction="sum")(out, tg.to(dev), il.to(dev), tl.to(dev)) / x.shape[0]
        opts[flag].zero_grad(); l.backward()
        ops.check_health(dev)
        worst = sorted(((float((p.grad.cpu() - gref[k]).norm() / (gref[k].norm() + 1e-30)), k) for k, p in m.named_parameters() if not k.endswith("conv.bias")), reverse=True)[:3]
        print("step", step, "persistent", flag, "loss %.6f ref %.6f" % (float(l), float(loss)), "worst grad rel err:", [(round(e, 7), k) for e, k in worst])
        opts[flag].step()
