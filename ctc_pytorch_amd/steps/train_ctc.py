"""Training driver counterpart of the reference's timit/steps/train_ctc.py on the HIP path.

`run_epoch` keeps the reference's signature and semantics (train_ctc.py:26-69): per minibatch forward ->
length conversion floor(float32(len/Tmax)*T_out) -> CTC loss (sum)/B -> greedy error count BEFORE the update ->
zero_grad/backward/step; returns (1 - errs/tokens, mean loss).  Differences are only in where work runs:
arg-max, path collapse and edit distance stay on the GPU (one small D2H of two integers per step instead of
a (B,T) index matrix + python loops), and the optimiser may be optim.FlatAdam (fused, flat buffers).

`main(conf)` accepts the same YAML keys as timit/conf/ctc_config.yaml and reproduces the dev-loss driven
LR-halving / best-state rollback / 8-halvings stop rule of train_ctc.py:160-249 without
visdom (stdout + JSONL).
"""
import argparse
import ast
import copy
import json
import os
import sys
import time

import numpy as np
import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from ctc_pytorch_amd import _lib, nn, ops, parallel  # noqa: E402
from ctc_pytorch_amd.models.model_ctc import CTC_Model  # noqa: E402
from ctc_pytorch_amd.optim import FlatAdam  # noqa: E402

supported_rnn = {"nn.LSTM": nn.LSTM, "nn.GRU": nn.GRU, "nn.RNN": nn.RNN}
supported_activate = {"relu": nn.ReLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid}


def frames_from_fraction(input_sizes, out_len):
    """(input_sizes * out_len).long() of train_ctc.py:46 / test_ctc.py:82-83: float32 product, truncation.
    Host logic on the (B,) CPU vector the loader produced; identical integers to the torch expression."""
    frac = np.asarray(input_sizes.detach().cpu().numpy() if torch.is_tensor(input_sizes) else input_sizes, dtype=np.float32)
    return (frac * np.float32(out_len)).astype(np.int64)


def run_epoch(epoch_id, model, data_iter, loss_fn, device, optimizer=None, print_every=20, is_training=True,
              global_batch=None, log=print):
    model.train() if is_training else model.eval()
    total_loss = 0.0
    total_tokens = 0
    total_errs = 0
    cur_loss = 0.0
    n_batches = 0
    # Step statistics travel one step behind the launches: the (loss, errors, tokens, health) vector of step i is copied into pinned
    # host memory asynchronously and read while step i+1 is already queued, so the host never drains the device inside the loop
    # (a blocking .cpu() per step left the GPU idle for the whole host-side enqueue of the next step: 39 vs 16 ms per cfg2 step).
    on_gpu = torch.device(device).type == "cuda"
    host_stats = [torch.empty(4, dtype=torch.float64).pin_memory() if on_gpu else torch.empty(4, dtype=torch.float64) for _ in range(2)]
    acc = dict(total_loss=0.0, cur_loss=0.0, total_errs=0, total_tokens=0, n_batches=0)

    def consume(pending):
        buf, ready, step = pending
        if ready is not None:
            ready.synchronize()
        if int(buf[3]) != 0:
            raise RuntimeError("ctc_pytorch_amd: a persistent recurrent kernel gave up waiting for a hand-off (status %d); "
                               "results of this step are poisoned -- set CTCN_RNN_PERSISTENT=0 to run one launch per timestep" % int(buf[3]))
        lv = float(buf[0])
        acc["cur_loss"] += lv
        acc["total_loss"] += lv
        acc["total_errs"] += int(buf[1])
        acc["total_tokens"] += int(buf[2])
        acc["n_batches"] = step + 1
        if (step + 1) % print_every == 0 and is_training:
            log("Epoch = %d, step = %d, cur_loss = %.4f, total_loss = %.4f, total_wer = %.4f" % (
                epoch_id, step + 1, acc["cur_loss"] / print_every, acc["total_loss"] / (step + 1), acc["total_errs"] / acc["total_tokens"]))
            acc["cur_loss"] = 0.0

    pending = None
    for i, data in enumerate(data_iter):
        inputs, input_sizes, targets, target_sizes, utt_list = data[:5]
        # data parallel (parallel.ShardedBatches): a 6th entry carries the utterance count of the GLOBAL minibatch; the loss
        # is sum_shard nll / B_global, so that the SUM all-reduce of the gradients yields the single-process gradient
        step_global = data[5] if len(data) > 5 else global_batch
        parallel.set_batch_split(step_global, int(inputs.shape[0]) if step_global else None, count=is_training)
        inputs = inputs.to(device, non_blocking=True)
        targets_d = targets.to(device, non_blocking=True)
        target_sizes_d = target_sizes.to(device, non_blocking=True)
        with torch.set_grad_enabled(is_training):
            out = model(inputs)
            out_len, batch_size, _ = out.size()
            if torch.is_tensor(input_sizes) and input_sizes.is_cuda:
                # DevicePrefetcher staged the fractions: the float32 product and truncation of train_ctc.py:46 on the device
                # (same IEEE multiply, same integers), no blocking pageable H2D in the middle of the step
                in_len = (input_sizes.float() * float(out_len)).long()
            else:
                in_len = torch.from_numpy(frames_from_fraction(input_sizes, out_len)).to(device)
            loss = loss_fn(out, targets_d, in_len, target_sizes_d)
            loss = loss / (step_global or batch_size)
        # greedy error count on the pre-update model, all on device
        idx = ops.argmax_last(out)                                        # (T,B) int32
        ids, ids_len = ops.greedy_collapse(idx, in_len, blank=0)
        dist = ops.edit_distance(ids, ids_len, targets_d, target_sizes_d)
        if is_training:
            optimizer.zero_grad()
            loss.backward()
            if isinstance(optimizer, FlatAdam):
                parallel.allreduce_grads(optimizer.grad)
            optimizer.step()
        # loss, error count, token count and the sticky hand-off status word of the persistent kernels
        if inputs.is_cuda:
            stats = ops.step_stats(loss, dist, target_sizes_d)               # one launch
        else:
            stats = torch.cat([torch.stack([loss.detach().double(), dist.sum().double(), target_sizes_d.sum().double()]), torch.zeros(1, dtype=torch.float64)])
        if step_global:
            stats = parallel.allreduce_stats(stats)          # global loss / error / token counts (and any rank's bad health)
        buf = host_stats[i % 2]
        buf.copy_(stats, non_blocking=True)
        ready = None
        if stats.is_cuda:
            ready = torch.cuda.Event()
            ready.record()
        if pending is not None:
            consume(pending)
        pending = (buf, ready, i)
    if pending is not None:
        consume(pending)
    total_loss, total_errs, total_tokens, n_batches = acc["total_loss"], acc["total_errs"], acc["total_tokens"], acc["n_batches"]
    parallel.set_batch_split(None, None)
    average_loss = total_loss / max(n_batches, 1)
    log("Epoch %d %s done, total_loss: %.4f, total_wer: %.4f" % (epoch_id, "Train" if is_training else "Valid", average_loss,
                                                                total_errs / max(total_tokens, 1)))
    return 1 - total_errs / max(total_tokens, 1), average_loss


class Config(object):
    batch_size = 4
    dropout = 0.1


def build_model_from_opts(opts, num_class):
    rnn_param = {"rnn_input_size": opts.rnn_input_size, "rnn_hidden_size": opts.rnn_hidden_size, "rnn_layers": opts.rnn_layers,
                 "rnn_type": supported_rnn[opts.rnn_type], "bidirectional": opts.bidirectional, "batch_norm": opts.batch_norm}
    cnn_param = {}
    lit = lambda v: ast.literal_eval(v) if isinstance(v, str) else v       # the reference eval()s these YAML strings
    channel, kernel_size, stride, padding = lit(opts.channel), lit(opts.kernel_size), lit(opts.stride), lit(opts.padding)
    pooling = lit(opts.pooling)
    cnn_param["batch_norm"] = opts.batch_norm
    cnn_param["activate_function"] = supported_activate[opts.activation_function]
    cnn_param["layer"] = []
    for layer in range(opts.layers):
        cnn_param["layer"].append([channel[layer], kernel_size[layer], stride[layer], padding[layer],
                                   pooling[layer] if pooling is not None else None])
    return CTC_Model(add_cnn=opts.add_cnn, cnn_param=cnn_param, rnn_param=rnn_param, num_class=num_class, drop_out=opts.drop_out)


class LRController:
    """Dev-loss driven LR halving with best-state rollback (train_ctc.py:160-227)."""

    def __init__(self, end_adjust_acc, decay):
        self.delta, self.decay = end_adjust_acc, decay
        self.loss_best = 1000
        self.loss_best_true = 1000
        self.adjust_rate_flag = False
        self.adjust_rate_count = 0
        self.adjust_time = 0
        self.acc_best = 0
        self.stop = False
        self.model_state = self.op_state = self.best_model_state = self.best_op_state = None

    def begin_epoch(self, optimizer):
        if self.adjust_rate_flag:
            self.adjust_rate_flag = False
            for g in optimizer.param_groups:
                g["lr"] *= self.decay

    def end_epoch(self, model, optimizer, acc, dev_loss):
        snap = lambda: (copy.deepcopy(model.state_dict()), copy.deepcopy(optimizer.state_dict()))
        if dev_loss < (self.loss_best - self.delta):
            self.loss_best = dev_loss
            self.loss_best_true = dev_loss
            self.adjust_rate_count = 0
            self.model_state, self.op_state = snap()
        elif dev_loss < self.loss_best + self.delta:
            self.adjust_rate_count += 1
            if dev_loss < self.loss_best and dev_loss < self.loss_best_true:
                self.loss_best_true = dev_loss
                self.model_state, self.op_state = snap()
        else:
            self.adjust_rate_count = 10
        if acc > self.acc_best:
            self.acc_best = acc
            self.best_model_state, self.best_op_state = snap()
        if self.adjust_rate_count == 10:
            self.adjust_rate_flag = True
            self.adjust_time += 1
            self.adjust_rate_count = 0
            if self.loss_best > self.loss_best_true:
                self.loss_best = self.loss_best_true
            if self.model_state is not None:
                model.load_state_dict(self.model_state)
                optimizer.load_state_dict(self.op_state)
        if self.adjust_time == 8:
            self.stop = True


def main(conf, train_loader=None, dev_loader=None, num_class=None, log=print):
    opts = Config()
    for k, v in conf.items():
        setattr(opts, k, v)
    rank, world, local = parallel.init_from_env()
    device = torch.device("cuda", local)
    # The host side of a step is collation and a 4 MB staging copy: a torch CPU op that fans out to every core (128 OpenMP
    # threads on the GPU box) between two steps stalls the launch thread for 8-10 ms (tools/epoch_probe.py).  CTCN_HOST_THREADS
    # overrides; one process per GPU shares the host anyway.
    threads_before = torch.get_num_threads()
    torch.set_num_threads(max(1, int(os.environ.get("CTCN_HOST_THREADS", "4"))))       # (restored when main returns)
    torch.manual_seed(opts.seed)
    np.random.seed(opts.seed)
    if train_loader is None:
        from ctc_pytorch_amd.utils.data_loader import Vocab, SpeechDataset, SpeechDataLoader
        vocab = Vocab(opts.vocab_file)
        num_class = vocab.n_words
        train_loader = SpeechDataLoader(SpeechDataset(vocab, opts.train_scp_path, opts.train_lab_path, opts),
                                        batch_size=opts.batch_size, shuffle=opts.shuffle_train, num_workers=opts.num_workers)
        dev_loader = SpeechDataLoader(SpeechDataset(vocab, opts.valid_scp_path, opts.valid_lab_path, opts),
                                      batch_size=opts.batch_size, shuffle=False, num_workers=opts.num_workers)
    if world > 1:
        # data parallel: `batch_size` stays the GLOBAL minibatch (the optimisation trajectory of the 1-GPU run); every rank
        # walks the same loader (same seed / shuffle) and keeps its contiguous shard of each batch, padded to the global T_max
        train_loader = parallel.ShardedBatches(train_loader, rank, world, log=log)
        dev_loader = parallel.ShardedBatches(dev_loader, rank, world, log=log)
        parallel.enable_sync_bn(bool(getattr(opts, "sync_bn", False)))
        parallel.enable_overlap(True)                     # run_epoch does one backward pass per optimiser step
    if rank != 0:
        log = lambda *_a, **_k: None                      # one log stream: every rank holds the same all-reduced statistics
    if device.type == "cuda":
        from ctc_pytorch_amd.utils.data_loader import DevicePrefetcher
        train_loader, dev_loader = DevicePrefetcher(train_loader, device), DevicePrefetcher(dev_loader, device)   # async double-buffered H2D
    model = build_model_from_opts(opts, num_class).to(device)
    log("Number of parameters %d" % sum(p.numel() for p in model.parameters()))
    loss_fn = nn.CTCLoss(reduction="sum")
    optimizer = FlatAdam(model, lr=opts.init_lr, weight_decay=opts.weight_decay)
    parallel.broadcast_params(optimizer.flat)
    ctl = LRController(opts.end_adjust_acc, opts.lr_decay)
    loss_results, dev_loss_results, dev_cer_results = [], [], []
    count = 0
    # the long-lived objects built so far (module tree, optimizer, loaders) leave the interpreter's cyclic collector: its generation-2
    # passes over them took ~50 ms each -- three or four whole training steps -- and the loop only stays two steps ahead of the device
    # (both are undone when the loop ends, however it ends: main is also a library entry point -- tests)
    import gc
    gc.collect()
    gc.freeze()
    start = time.time()
    try:
        while not ctl.stop and count < opts.num_epoches:
            count += 1
            ctl.begin_epoch(optimizer)
            log("Start training epoch: %d, learning_rate: %.5f" % (count, optimizer.param_groups[0]["lr"]))
            _, loss = run_epoch(count, model, train_loader, loss_fn, device, optimizer=optimizer, print_every=opts.verbose_step,
                                is_training=True, log=log)
            # per-shard BatchNorm: one set of running statistics for the evaluation below and for rank 0's checkpoint
            parallel.sync_bn_buffers(model)
            acc, dev_loss = run_epoch(count, model, dev_loader, loss_fn, device, optimizer=None, print_every=opts.verbose_step,
                                      is_training=False, log=log)
            loss_results.append(loss)
            dev_loss_results.append(dev_loss)
            dev_cer_results.append(acc)
            ctl.end_epoch(model, optimizer, acc, dev_loss)
            log(json.dumps(dict(epoch=count, train_loss=loss, dev_loss=dev_loss, dev_acc=acc, adjust_time=ctl.adjust_time,
                                minutes=(time.time() - start) / 60)))
    finally:
        gc.unfreeze()
        torch.set_num_threads(threads_before)
    if ctl.best_model_state is not None:
        model.load_state_dict(ctl.best_model_state)
        optimizer.load_state_dict(ctl.best_op_state)
    if rank == 0 and getattr(opts, "checkpoint_dir", None):
        save_dir = os.path.join(opts.checkpoint_dir, opts.exp_name)
        os.makedirs(save_dir, exist_ok=True)
        params = {"num_epoches": opts.num_epoches, "end_adjust_acc": opts.end_adjust_acc, "seed": opts.seed, "decay": opts.lr_decay,
                  "learning_rate": opts.init_lr, "weight_decay": opts.weight_decay, "batch_size": opts.batch_size,
                  "feature_type": getattr(opts, "feature_type", "fbank"), "n_feats": getattr(opts, "feature_dim", 40), "epoch": count}
        torch.save(CTC_Model.save_package(model, optimizer=optimizer, epoch=params, loss_results=loss_results,
                                          dev_loss_results=dev_loss_results, dev_cer_results=dev_cer_results),
                   os.path.join(save_dir, "ctc_best_model.pkl"))
    return model, dict(loss=loss_results, dev_loss=dev_loss_results, dev_acc=dev_cer_results)


if __name__ == "__main__":
    import yaml
    ap = argparse.ArgumentParser(description="cnn_lstm_ctc on MI355X")
    ap.add_argument("--conf", default="conf/ctc_config.yaml")
    a = ap.parse_args()
    main(yaml.safe_load(open(a.conf, "r")))
