"""CTC acoustic model on the gfx950 HIP kernels -- same class surface as the reference's
timit/models/model_ctc.py (BatchRNN :13-36, LayerCNN :38-68, CTC_Model :70-229), so that
`from models.model_ctc import *` in the reference's train_ctc.py / test_ctc.py resolves here unchanged
(put the ctc_pytorch_amd/ directory on sys.path).

Same constructor arguments, same module tree => identical `state_dict` keys and shapes
(conv.{n}.conv.*, conv.{n}.batch_norm.*, rnns.{l}.batch_norm.*, rnns.{l}.rnn.weight_{ih,hh}_l0[_reverse],
fc.0.* / fc.1.weight or fc.weight), so checkpoints written by either side load in the other.
Every tensor op in forward() runs in libctcn.so (ctc_pytorch_amd.nn / ops); there is no torch fallback.
"""
import math
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:            # drop-in mode: only ctc_pytorch_amd/ itself is on sys.path
    sys.path.insert(0, _ROOT)

from ctc_pytorch_amd import nn, ops  # noqa: E402  (`nn` is re-exported on purpose: it shadows torch.nn in the drivers)

F = nn.functional

__author__ = "ctc_pytorch_amd (MI355X-native rebuild; class surface after Ruchao Fan's CTC_pytorch)"


class BatchRNN(nn.Module):
    """[BatchNorm1d over all T*B rows] -> bias-free (bi)RNN -> dropout   (reference model_ctc.py:13-36)."""

    def __init__(self, input_size, hidden_size, rnn_type=nn.LSTM, bidirectional=False, batch_norm=True, dropout=0.1):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.bidirectional = bidirectional
        self.batch_norm = nn.BatchNorm1d(input_size) if batch_norm else None
        self.rnn = _native_rnn(rnn_type)(input_size=input_size, hidden_size=hidden_size, bidirectional=bidirectional, bias=False)
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, x):                       # x: (T,B,C) time-major
        if self.batch_norm is not None:
            T, B, C = x.shape
            x = ops.batch_norm(ops.contiguous(x), self.batch_norm.weight, self.batch_norm.bias, self.batch_norm.running_mean,
                               self.batch_norm.running_var, T * B, C, 1, self.batch_norm.training,
                               0.1 if self.batch_norm.momentum is None else self.batch_norm.momentum, self.batch_norm.eps,
                               num_batches_tracked=self.batch_norm.num_batches_tracked)
        if isinstance(self.rnn, (nn.LSTM, nn.GRU, nn.RNN)):
            # the dropout rides along with the recurrent layer (same mask, same values as self.dropout(x); the recurrence stores the
            # dropped output itself where its tagged-gather kernel applies)
            x, _ = self.rnn(x, drop_p=float(self.dropout.p) if self.dropout.training else 0.0)
            return x
        x, _ = self.rnn(x)
        return self.dropout(x)


def _native_rnn(rnn_type):
    """Map torch.nn.{LSTM,GRU,RNN} (what an unmodified caller may still pass) onto the HIP-backed classes."""
    import torch.nn as tnn
    table = {tnn.LSTM: nn.LSTM, tnn.GRU: nn.GRU, tnn.RNN: nn.RNN}
    return table.get(rnn_type, rnn_type)


class LayerCNN(nn.Module):
    """Conv2d(bias) -> BatchNorm2d -> activation -> [pool] -> dropout, or the Conv1d / BatchNorm1d / MaxPool1d stack for a one-element
    kernel_size   (reference model_ctc.py:38-68).

    As in the reference only nn.ReLU can be constructed (it passes inplace=True to the activation class,
    which nn.Tanh / nn.Sigmoid reject); BN + ReLU run as one fused apply pass."""

    def __init__(self, in_channel, out_channel, kernel_size, stride, padding, pooling_size=None, activation_function=nn.ReLU,
                 batch_norm=True, dropout=0.1):
        super().__init__()
        if len(kernel_size) == 2:
            self.conv = nn.Conv2d(in_channel, out_channel, kernel_size=kernel_size, stride=stride, padding=padding)
            self.batch_norm = nn.BatchNorm2d(out_channel) if batch_norm else None
        else:           # the reference's one-element kernel_size branch (model_ctc.py:48-50, 54-55): a stand-alone (B,C,L) layer --
            # CTC_Model.forward feeds 4-D tensors, which nn.Conv1d refuses there as here
            self.conv = nn.Conv1d(in_channel, out_channel, kernel_size=kernel_size, stride=stride, padding=padding)
            self.batch_norm = nn.BatchNorm1d(out_channel) if batch_norm else None
        self.activation = _native_act(activation_function)(inplace=True)
        if pooling_size is not None and len(kernel_size) == 2:
            self.pooling = nn.MaxPool2d(pooling_size)
        elif len(kernel_size) == 1:
            self.pooling = nn.MaxPool1d(pooling_size)
        else:
            self.pooling = None
        self.dropout = nn.Dropout(p=dropout)
        self._fused = isinstance(self.activation, nn.ReLU) and self.batch_norm is not None
        if self._fused:
            self.batch_norm.fuse_relu = True

    def forward(self, x):
        x = self.conv(x)
        # BN + ReLU + dropout in one pass (round 5): same values, same random stream.  Only when BOTH modules train and 0 < p < 1 (a
        # frozen BatchNorm in eval mode next to a training dropout, or p = 1, take the separate modules below, as the reference's stack does)
        if self._fused and self.pooling is None and self.dropout.training and self.batch_norm.training and 0.0 < self.dropout.p < 1.0:
            return self.batch_norm(x, drop_p=float(self.dropout.p))
        if self._fused:
            x = self.batch_norm(x)              # BN + ReLU in one pass
        else:
            if self.batch_norm is not None:
                x = self.batch_norm(x)
            x = self.activation(x)
        if self.pooling is not None:
            x = self.pooling(x)
        return self.dropout(x)


def _native_act(act):
    import torch.nn as tnn
    return nn.ReLU if act is tnn.ReLU else act


def _conv_out_features(feat, kernel_size, stride, padding):
    """Frequency-axis size after one Conv2d layer: axis 1 of kernel / stride / padding (model_ctc.py:111)."""
    return int(math.floor((feat + 2 * padding[1] - kernel_size[1]) / stride[1]) + 1)


class CTC_Model(nn.Module):
    def __init__(self, add_cnn=False, cnn_param=None, rnn_param=None, num_class=39, drop_out=0.1):
        """Arguments as the reference (model_ctc.py:71-81):
        cnn_param = {"layer": [[(cin, cout), (kh, kw), (sh, sw), (ph, pw), pool|None], ...], "batch_norm": bool,
                     "activate_function": nn.ReLU}
        rnn_param = {"rnn_input_size", "rnn_hidden_size", "rnn_layers", "rnn_type", "bidirectional", "batch_norm"}"""
        super().__init__()
        if type(rnn_param) != dict:
            raise ValueError("rnn_param need to be a dict to contain all params of rnn!")
        self.add_cnn, self.cnn_param, self.rnn_param = add_cnn, cnn_param, rnn_param
        self.num_class, self.drop_out = num_class, drop_out
        self.num_directions = 1 + int(bool(rnn_param["bidirectional"]))
        width = self.num_directions * rnn_param["rnn_hidden_size"]

        rnn_in = rnn_param["rnn_input_size"]
        if add_cnn:
            self.conv, rnn_in = self._front_end(cnn_param, rnn_in, drop_out)
        self.rnns = self._recurrent_stack(rnn_param, rnn_in, width, drop_out)
        head = nn.Linear(width, num_class, bias=False)
        self.fc = nn.Sequential(nn.BatchNorm1d(width), head) if rnn_param["batch_norm"] else head
        self.log_softmax = nn.LogSoftmax(dim=-1)

    @staticmethod
    def _front_end(cnn_param, feat, drop_out):
        """conv.{n} = LayerCNN per entry of cnn_param['layer']; returns (Sequential, features fed to the first RNN)."""
        blocks, channels = OrderedDict(), 1
        for n, ((cin, channels), kernel_size, stride, padding, pool) in enumerate(cnn_param["layer"]):
            blocks[str(n)] = LayerCNN(cin, channels, kernel_size, stride, padding, pool, activation_function=cnn_param["activate_function"],
                                      batch_norm=cnn_param["batch_norm"], dropout=drop_out)
            if len(kernel_size) == 2:
                feat = _conv_out_features(feat, kernel_size, stride, padding)
        return nn.Sequential(blocks), feat * channels

    @staticmethod
    def _recurrent_stack(rnn_param, first_in, width, drop_out):
        """rnns.{l} = BatchRNN; layer 0 has no BatchNorm, the others normalise their (dirs * H)-wide input."""
        common = dict(hidden_size=rnn_param["rnn_hidden_size"], rnn_type=rnn_param["rnn_type"], bidirectional=rnn_param["bidirectional"],
                      dropout=drop_out)
        blocks = OrderedDict()
        for layer in range(rnn_param["rnn_layers"]):
            blocks[str(layer)] = BatchRNN(input_size=first_in if layer == 0 else width, batch_norm=bool(layer) and rnn_param["batch_norm"], **common)
        return nn.Sequential(blocks)

    def forward(self, x, visualize=False):
        """x: (B, T, F) float32 on a ROCm device -> log-probs (T', B, num_class); with visualize=True also the
        list [x, (conv_out, rnn_in,) out] the reference returns (model_ctc.py:142-185)."""
        visual = [x] if visualize else None
        if self.add_cnn:
            c = self.conv(x.unsqueeze(1))
            if visualize:
                visual.append(c)
            if c.dim() != 4:
                raise NotImplementedError("Conv1d front-end")
            h = ops.bctf_to_tbcf(c)                     # (B,C,T',F') -> (T',B,C*F'), feature = c*F'+f
            if visualize:
                visual.append(h)
        else:
            h = ops.contiguous(x.transpose(0, 1))       # (T,B,F)
        h = self.rnns(h)
        T, B, _ = h.shape
        z = self.fc(h.reshape(T * B, -1))
        out = self.log_softmax(z.view(T, B, -1))
        if visualize:
            visual.append(out)
            return out, visual
        return out

    # ---- training-time greedy error count (reference model_ctc.py:187-202) ---------------------------------
    def compute_wer(self, index, input_sizes, targets, target_sizes):
        """index (B,T) arg-max ids, input_sizes (B) frames, targets (B,Lmax), target_sizes (B) -- numpy arrays or
        tensors.  Collapse (drop blank 0, drop frame-to-frame repeats, first input_sizes[i] frames) and the
        Levenshtein distance run on the device of the model; returns (errs, tokens) python ints."""
        dev = next(self.parameters()).device
        idx = torch.as_tensor(np.asarray(index) if not torch.is_tensor(index) else index).to(dev).to(torch.int32)
        lens = torch.as_tensor(np.asarray(input_sizes) if not torch.is_tensor(input_sizes) else input_sizes).to(dev)
        tg = torch.as_tensor(np.asarray(targets) if not torch.is_tensor(targets) else targets).to(dev)
        tl = torch.as_tensor(np.asarray(target_sizes) if not torch.is_tensor(target_sizes) else target_sizes).to(dev)
        ids, ids_len = ops.greedy_collapse(idx, lens, blank=0, batch_major=True)
        dist = ops.edit_distance(ids, ids_len, tg, tl)
        return int(dist.sum().item()), int(tl.sum().item())

    def add_weights_noise(self):
        # dead code in the reference as well (model_ctc.py:204-207 rebinds a local and changes nothing)
        return None

    @staticmethod
    def save_package(model, optimizer=None, decoder=None, epoch=None, loss_results=None, dev_loss_results=None,
                     dev_cer_results=None):
        """Checkpoint dict with the reference's keys (model_ctc.py:209-229), loadable by either side."""
        package = {"rnn_param": model.rnn_param, "add_cnn": model.add_cnn, "cnn_param": model.cnn_param,
                   "num_class": model.num_class, "_drop_out": model.drop_out, "state_dict": model.state_dict()}
        optional = {"optim_dict": None if optimizer is None else optimizer.state_dict(), "decoder": decoder, "epoch": epoch}
        package.update({k: v for k, v in optional.items() if v is not None})
        if loss_results is not None:
            package.update(loss_results=loss_results, dev_loss_results=dev_loss_results, dev_cer_results=dev_cer_results)
        return package
