"""torch-CPU counterpart of the reference's module stack  --  TEST INFRASTRUCTURE ONLY.

The reference's arithmetic lives in torch (unpinned; 2.10.0 in this image).  This file re-issues, on CPU,
exactly the torch calls the reference makes -- so it can serve (a) as the checker at shapes too large for the
NumPy restatement and (b) as bench.py's `cpu_baseline` ("port") on the GPU box, where /root/reference does
not exist.  Call sites restated:
  nn.LSTM/GRU/RNN(input, hidden, bidirectional, bias=False)        timit/models/model_ctc.py:24-25
  BatchNorm1d on x.transpose(-1,-2) (stats over all T*B rows)       :29-32
  Conv2d(bias) -> BatchNorm2d -> ReLU(inplace) -> Dropout           :46-67
  (B,C,T,F) -> transpose(1,2) -> view -> transpose(0,1)             :153-158
  BatchNorm1d + Linear(bias=False), LogSoftmax(dim=-1)              :135-140,165-168
  nn.CTCLoss(reduction='sum')(out, targets, in_len, tgt_len)/B      timit/steps/train_ctc.py:144,47-48
  torch.optim.Adam(lr, weight_decay)                                :145
State-dict keys equal the reference's (conv.N.conv.*, rnns.N.rnn.*, fc.0.* / fc.1.weight), so weights move
between this, the reference and ctc_pytorch_amd.models.CTC_Model with load_state_dict.
Pinned by tests/test_oracle_golden.py::test_torch_cpu_counterpart against tests/golden/model_*.npz.
"""
from collections import OrderedDict

import torch
import torch.nn as tnn


class _RnnBlock(tnn.Module):
    def __init__(self, n_in, hidden, rnn_cls, bidirectional, batch_norm, p):
        super().__init__()
        self.batch_norm = tnn.BatchNorm1d(n_in) if batch_norm else None
        self.rnn = rnn_cls(input_size=n_in, hidden_size=hidden, bidirectional=bidirectional, bias=False)
        self.dropout = tnn.Dropout(p=p)

    def forward(self, x):
        if self.batch_norm is not None:
            x = self.batch_norm(x.transpose(-1, -2)).transpose(-1, -2)
        x, _ = self.rnn(x)
        return self.dropout(x)


class _CnnBlock(tnn.Module):
    def __init__(self, cin, cout, k, s, pad, batch_norm, p, pool=None):
        super().__init__()
        self.conv = tnn.Conv2d(cin, cout, kernel_size=k, stride=s, padding=pad)
        self.batch_norm = tnn.BatchNorm2d(cout) if batch_norm else None
        self.activation = tnn.ReLU(inplace=True)
        self.pooling = tnn.MaxPool2d(pool) if pool is not None else None        # model_ctc.py:52-53
        self.dropout = tnn.Dropout(p=p)

    def forward(self, x):
        x = self.conv(x)
        if self.batch_norm is not None:
            x = self.batch_norm(x)
        x = self.activation(x)
        if self.pooling is not None:
            x = self.pooling(x)
        return self.dropout(x)


class TorchCpuCTCModel(tnn.Module):
    def __init__(self, add_cnn=False, cnn_param=None, rnn_param=None, num_class=39, drop_out=0.1):
        super().__init__()
        self.add_cnn = add_cnn
        feat = rnn_param["rnn_input_size"]
        if add_cnn:
            blocks = []
            cout = 1
            for n, ((cin, cout), k, s, pad, pool) in enumerate(cnn_param["layer"]):
                blocks.append((str(n), _CnnBlock(cin, cout, k, s, pad, cnn_param["batch_norm"], drop_out, pool)))
                feat = (feat + 2 * pad[1] - k[1]) // s[1] + 1
            self.conv = tnn.Sequential(OrderedDict(blocks))
            feat *= cout
        H, bi = rnn_param["rnn_hidden_size"], rnn_param["bidirectional"]
        D = 2 if bi else 1
        cls = rnn_param["rnn_type"]
        blocks = [("0", _RnnBlock(feat, H, cls, bi, False, drop_out))]
        for i in range(1, rnn_param["rnn_layers"]):
            blocks.append((str(i), _RnnBlock(D * H, H, cls, bi, rnn_param["batch_norm"], drop_out)))
        self.rnns = tnn.Sequential(OrderedDict(blocks))
        if rnn_param["batch_norm"]:
            self.fc = tnn.Sequential(tnn.BatchNorm1d(D * H), tnn.Linear(D * H, num_class, bias=False))
        else:
            self.fc = tnn.Linear(D * H, num_class, bias=False)
        self.log_softmax = tnn.LogSoftmax(dim=-1)

    def forward(self, x, visualize=False):
        vis = [x]
        if self.add_cnn:
            c = self.conv(x.unsqueeze(1))
            vis.append(c)
            h = c.transpose(1, 2).contiguous()
            h = h.view(h.size(0), h.size(1), -1).transpose(0, 1).contiguous()
            vis.append(h)
        else:
            h = x.transpose(0, 1)
        h = self.rnns(h)
        T, B, _ = h.shape
        out = self.log_softmax(self.fc(h.reshape(T * B, -1)).view(T, B, -1))
        vis.append(out)
        return (out, vis) if visualize else out


def train_step(model, opt, x, frac, targets, tgt_len):
    """One step of the reference's loop body (train_ctc.py:44-48,61-65): returns the python float loss."""
    out = model(x)
    in_len = (frac * out.size(0)).long()
    loss = tnn.CTCLoss(reduction="sum")(out, targets, in_len, tgt_len) / out.size(1)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return float(loss.item())
