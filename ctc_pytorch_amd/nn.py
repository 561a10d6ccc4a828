"""HIP-backed drop-ins for the `torch.nn` names the reference drivers use.

The reference's train/test scripts do `import torch.nn as nn` and then `from models.model_ctc import *`
(timit/steps/train_ctc.py:13,16; test_ctc.py:9,13), so whatever `nn` our models.model_ctc exports shadows
theirs: `nn.CTCLoss(reduction='sum')` (train_ctc.py:144) and `supported_rnn = {'nn.LSTM': nn.LSTM, ...}`
(:20) resolve to the classes below with the drivers textually unchanged.  Each class subclasses the torch
module purely as a *parameter container* (same constructor, same init => same-seed equality, identical
state_dict keys) and replaces `forward` with calls into libctcn.so.  Names not defined here fall through to
torch.nn (module-level __getattr__).
"""
import torch
import torch.nn as _tnn

from . import ops

Module = _tnn.Module
Sequential = _tnn.Sequential
Parameter = _tnn.Parameter


def __getattr__(name):          # everything else (init, utils, functional, ...) is plain torch.nn
    return getattr(_tnn, name)


class _RecurrentMixin:
    _cell = None

    def _check(self):
        if self.num_layers != 1 or self.bias or self.batch_first or getattr(self, "proj_size", 0):
            raise NotImplementedError(
                "ctc_pytorch_amd.nn.%s supports what the reference constructs: num_layers=1, bias=False, "
                "batch_first=False (model_ctc.py:24-25)" % type(self).__name__)
        if float(self.dropout) != 0.0:
            raise NotImplementedError("inter-layer dropout of a 1-layer RNN is a no-op; got dropout=%r" % self.dropout)

    def forward(self, x, hx=None, drop_p=0.0):
        """drop_p: the dropout that FOLLOWS this layer (BatchRNN), applied to the returned output in training mode -- inside the
        recurrent kernel where that is possible (ops.rnn_layer), as a dropout pass otherwise; the same values either way."""
        if hx is not None:
            raise NotImplementedError("initial state is always zero on the reference path (model_ctc.py:33)")
        self._check()
        w1 = (self.weight_ih_l0_reverse, self.weight_hh_l0_reverse) if self.bidirectional else (None, None)
        H = self.hidden_size
        if H % 4 == 0:
            y = ops.rnn_layer(x, self.weight_ih_l0, self.weight_hh_l0, w1[0], w1[1], self._cell, self.training, drop_p if self.training else 0.0)
            return y, None
        # The kernels move rows of W_hh / h in 16-byte pieces (H % 4 == 0).  Any other hidden size (nn.LSTM takes every H,
        # model_ctc.py:24-25) runs as the next multiple of 4 with all-zero weights for the extra units: their pre-activations are 0,
        # so c = h = 0 for every cell type at every step and the real units never see them; the padding / slicing around the
        # kernels is plain tensor plumbing (autograd routes the gradients of the real rows back through it).
        Hp, G = H + (4 - H % 4), {"lstm": 4, "gru": 3, "tanh": 1}[self._cell]

        def pad(w_ih, w_hh):
            wi = _tnn.functional.pad(w_ih.view(G, H, -1), (0, 0, 0, Hp - H)).reshape(G * Hp, -1)
            wh = _tnn.functional.pad(w_hh.view(G, H, H), (0, Hp - H, 0, Hp - H)).reshape(G * Hp, Hp)
            return wi.contiguous(), wh.contiguous()

        p0 = pad(self.weight_ih_l0, self.weight_hh_l0)
        p1 = pad(*w1) if self.bidirectional else (None, None)
        yp = ops.rnn_layer(x, p0[0], p0[1], p1[0], p1[1], self._cell, self.training)
        T, B, _ = yp.shape
        D = 2 if self.bidirectional else 1
        y = ops.contiguous(yp.view(T, B, D, Hp)[..., :H]).view(T, B, D * H)               # drop the padded units (strided gather kernel)
        return ops.dropout(y, drop_p, self.training), None

    def flatten_parameters(self):
        return None


class LSTM(_RecurrentMixin, _tnn.LSTM):
    """nn.LSTM(input_size, hidden_size, bidirectional=..., bias=False): gate rows i,f,g,o."""
    _cell = "lstm"


class GRU(_RecurrentMixin, _tnn.GRU):
    """nn.GRU(..., bias=False): gate rows r,z,n."""
    _cell = "gru"


class RNN(_RecurrentMixin, _tnn.RNN):
    """nn.RNN(..., bias=False), tanh."""
    _cell = "tanh"

    def _check(self):
        super()._check()
        if self.nonlinearity != "tanh":
            raise NotImplementedError("only nonlinearity='tanh' (the nn.RNN default the reference uses)")


class BatchNorm1d(_tnn.BatchNorm1d):
    """(N,C) or (N,C,L) input, statistics per channel over N*L -- BatchRNN feeds (T,C,B) views (model_ctc.py:29-32)."""

    fuse_relu = False

    def forward(self, x):
        if not (self.affine and self.track_running_stats):
            raise NotImplementedError("affine=True, track_running_stats=True only (nn.BatchNorm1d defaults)")
        C = self.num_features
        training = self.training
        mom = 0.1 if self.momentum is None else self.momentum
        nbt = self.num_batches_tracked              # (+= 1 by the statistics kernel in training: no launch of its own)
        if x.dim() == 2:
            return ops.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, x.shape[0], C, 1, training,
                                  mom, self.eps, self.fuse_relu, nbt)
        if x.dim() != 3 or x.shape[1] != C:
            raise ValueError("BatchNorm1d expects (N,C) or (N,C,L)")
        xt = x.transpose(1, 2)                     # (N,L,C)
        if xt.is_contiguous():                     # the reference's x.transpose(-1,-2) of a (T,B,C) tensor
            N, Lq = xt.shape[0], xt.shape[1]
            y = ops.batch_norm(xt, self.weight, self.bias, self.running_mean, self.running_var, N * Lq, C, 1, training, mom,
                               self.eps, self.fuse_relu, nbt)
            return y.view(N, Lq, C).transpose(1, 2)
        x = ops.contiguous(x)
        return ops.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, x.shape[0], C, x.shape[2], training,
                              mom, self.eps, self.fuse_relu, nbt)


class BatchNorm2d(_tnn.BatchNorm2d):
    """NCHW, statistics per channel over (B,T,F) (model_ctc.py:47,63)."""

    fuse_relu = False

    def forward(self, x, drop_p=0.0):
        """drop_p > 0 (LayerCNN, training): the layer's dropout rides along with the apply pass (ops.batch_norm)."""
        if not (self.affine and self.track_running_stats):
            raise NotImplementedError("affine=True, track_running_stats=True only")
        if x.dim() != 4 or x.shape[1] != self.num_features:
            raise ValueError("BatchNorm2d expects (B,C,H,W)")
        mom = 0.1 if self.momentum is None else self.momentum
        x = ops.contiguous(x)
        return ops.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, x.shape[0], x.shape[1],
                              x.shape[2] * x.shape[3], self.training, mom, self.eps, self.fuse_relu, self.num_batches_tracked,
                              drop_p=float(drop_p) if self.training else 0.0)


class Linear(_tnn.Linear):
    def forward(self, x):
        if self.bias is not None:
            raise NotImplementedError("bias=False only (model_ctc.py:137,139)")
        shp = x.shape
        y = ops.linear(x.reshape(-1, shp[-1]), self.weight)
        return y.view(*shp[:-1], self.out_features)


class Conv2d(_tnn.Conv2d):
    def forward(self, x):
        if self.groups != 1 or tuple(self.dilation) != (1, 1) or self.padding_mode != "zeros" or isinstance(self.padding, str):
            raise NotImplementedError("plain Conv2d (groups=1, dilation=1, zero padding) only (model_ctc.py:46)")
        return ops.conv2d(x, self.weight, self.bias, self.stride, self.padding)


class ReLU(_tnn.Module):
    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def forward(self, x):
        return ops.relu(x)


class Dropout(_tnn.Dropout):
    def forward(self, x):
        return ops.dropout(x, self.p, self.training)


class LogSoftmax(_tnn.Module):
    def __init__(self, dim=None):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        if self.dim not in (-1, x.dim() - 1):
            raise NotImplementedError("LogSoftmax over the last dim only (model_ctc.py:140)")
        return ops.log_softmax(x)


class CTCLoss(_tnn.Module):
    """nn.CTCLoss(blank=0, reduction='sum', zero_infinity=False) as called at train_ctc.py:144,47."""

    def __init__(self, blank=0, reduction="mean", zero_infinity=False):
        super().__init__()
        if blank != 0:
            raise NotImplementedError("blank index 0 only (data_loader.py:16)")
        if zero_infinity:
            raise NotImplementedError("zero_infinity=False only (train_ctc.py:144)")
        if reduction not in ("sum", "none"):
            raise NotImplementedError("reduction='sum' (train_ctc.py:144) or 'none'")
        self.reduction = reduction

    def forward(self, log_probs, targets, input_lengths, target_lengths):
        return ops.ctc_loss(log_probs, targets, torch.as_tensor(input_lengths), torch.as_tensor(target_lengths), self.reduction)


class MaxPool2d(_tnn.Module):
    """nn.MaxPool2d(pooling_size) of LayerCNN (model_ctc.py:52-53): stride = kernel, no padding, floor."""

    def __init__(self, kernel_size, stride=None, padding=0, dilation=1, return_indices=False, ceil_mode=False):
        super().__init__()
        ks = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        st = ks if stride is None else ((stride, stride) if isinstance(stride, int) else tuple(stride))
        if st != ks or padding not in (0, (0, 0)) or dilation not in (1, (1, 1)) or return_indices or ceil_mode:
            raise NotImplementedError("MaxPool2d(kernel_size) with its defaults only (model_ctc.py:53)")
        self.kernel_size = ks

    def forward(self, x):
        return ops.max_pool2d(x, self.kernel_size)

    def extra_repr(self):
        return "kernel_size=%s" % (self.kernel_size,)


class Conv1d(_tnn.Conv1d):
    """nn.Conv1d of LayerCNN's one-element kernel_size branch (model_ctc.py:48-50): (B,Ci,L) -> (B,Co,L').  Runs on the Conv2d kernels as
    an image of height 1 (kernel (1,k), stride (1,s), padding (0,p)): the fma chain of an output element is the same one, and the parameter
    keeps nn.Conv1d's (Co,Ci,k) shape, so state_dicts are interchangeable with the reference's."""

    def forward(self, x):
        if self.groups != 1 or tuple(self.dilation) != (1,) or self.padding_mode != "zeros" or isinstance(self.padding, str):
            raise NotImplementedError("plain Conv1d (groups=1, dilation=1, zero padding) only (model_ctc.py:49)")
        if x.dim() != 3 or x.shape[1] != self.in_channels:
            raise ValueError("Conv1d expects (B,C,L)")          # (a 4-D input -- what CTC_Model.forward feeds -- is refused by torch as well)
        y = ops.conv2d(ops.contiguous(x).unsqueeze(2), self.weight.unsqueeze(2), self.bias, (1, self.stride[0]), (0, self.padding[0]))
        return y.squeeze(2)


class MaxPool1d(_tnn.Module):
    """nn.MaxPool1d(pooling_size) of LayerCNN (model_ctc.py:54-55): stride = kernel, no padding, floor; (B,C,L) -> (B,C,L // k)."""

    def __init__(self, kernel_size, stride=None, padding=0, dilation=1, return_indices=False, ceil_mode=False):
        super().__init__()
        if isinstance(kernel_size, (tuple, list)):
            (kernel_size,) = kernel_size
        if (stride not in (None, kernel_size, (kernel_size,))) or padding not in (0, (0,)) or dilation not in (1, (1,)) or return_indices or ceil_mode:
            raise NotImplementedError("MaxPool1d(kernel_size) with its defaults only (model_ctc.py:55)")
        self.kernel_size = kernel_size          # (None constructs, as in torch -- LayerCNN builds MaxPool1d(None) for pooling_size=None -- and fails in forward)

    def forward(self, x):
        if self.kernel_size is None:
            raise TypeError("MaxPool1d: kernel_size is None (LayerCNN's one-element kernel_size branch needs a pooling_size, model_ctc.py:54-55)")
        if x.dim() != 3:
            raise ValueError("MaxPool1d expects (B,C,L)")
        return ops.max_pool2d(ops.contiguous(x).unsqueeze(2), (1, int(self.kernel_size))).squeeze(2)

    def extra_repr(self):
        return "kernel_size=%s" % (self.kernel_size,)
