"""Worker of the data-parallel GPU tests (tests/test_gpu_kernels.py): one process per rank, several ranks on ONE MI355X
(gloo carries the collectives: RCCL refuses two ranks per device).  Not a test module itself.

    python dp_worker.py equiv <out.json>          2-rank DP step == single-process step on the same global batch
    python dp_worker.py main  <corpus_dir> <out>  steps/train_ctc.main on a toy corpus; writes history + parameter checksum
    python dp_worker.py overlap <out.json>        early all-reduce of gradient slices: even shards reduce early, uneven ones never diverge
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ctc_pytorch_amd import nn, ops, parallel  # noqa: E402
from ctc_pytorch_amd.models.model_ctc import CTC_Model  # noqa: E402
from ctc_pytorch_amd.optim import FlatAdam  # noqa: E402
from ctc_pytorch_amd.testing import synth  # noqa: E402  (synthetic inputs only)


def build(dev, cnn, H=32):
    rp = {"rnn_input_size": 40, "rnn_hidden_size": H, "rnn_layers": 2, "rnn_type": nn.LSTM, "bidirectional": True, "batch_norm": True}
    if cnn:
        cp = {"batch_norm": True, "activate_function": nn.ReLU, "layer": [[(1, 32), (3, 3), (1, 2), (1, 1), None], [(32, 32), (3, 3), (2, 2), (1, 1), None]]}
        m = CTC_Model(add_cnn=True, cnn_param=cp, rnn_param=rp, num_class=30, drop_out=0.0)
    else:
        m = CTC_Model(rnn_param=rp, num_class=30, drop_out=0.0)
    vals = synth.fill_state_dict([(k, tuple(v.shape)) for k, v in m.state_dict().items()], seed=17)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in vals.items()})
    return m.to(dev).train()


def one_step(model, opt, batch, lo, hi, global_b, dev):
    """forward / CTC / backward of rows [lo, hi) of the global batch, loss = sum_shard nll / B_global."""
    x = torch.from_numpy(batch["x"][lo:hi]).to(dev)
    tg, tl = torch.from_numpy(batch["targets"][lo:hi]).to(dev), torch.from_numpy(batch["tgt_len"][lo:hi]).to(dev)
    parallel.set_batch_split(global_b, hi - lo)
    out = model(x)
    frac = batch["frac"][lo:hi].astype(np.float32)
    in_len = torch.from_numpy((frac * np.float32(out.size(0))).astype(np.int64)).to(dev)
    loss = nn.CTCLoss(reduction="sum")(out, tg, in_len, tl) / global_b
    opt.zero_grad()
    loss.backward()
    ops.join_side_stream()
    return loss.detach().double(), out.detach()


def equiv(out_path):
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    # CTCN_TEST_H / CTCN_TEST_PRECISION (round 5): the same comparison on the BENCHMARKED kernels -- persistent recurrences (the caller leaves
    # CTCN_RNN_PERSISTENT at 1) and bf16x3 GEMMs -- at a hidden size whose grids (dirs x 1 batch tile x H / 16 workgroups per rank) fit
    # the chip twice, so that the two ranks' launches can be co-resident
    H = int(os.environ.get("CTCN_TEST_H", "32"))
    ops.set_precision(int(os.environ.get("CTCN_TEST_PRECISION", "0")))
    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ["RANK"])
    results = {}
    for cnn in (False, True):
        B = 7                                           # uneven shards: 4 + 3 utterances
        batch = synth.make_batch(seed=5, B=B, T=48, F=40, V=30, lab_lo=3, lab_hi=6)
        ref = None
        if rank == 0:                                   # single process, whole batch, plain BatchNorm, no collective
            m = build(dev, cnn, H)
            opt = FlatAdam(m, lr=1e-3)
            loss, lp = one_step(m, opt, batch, 0, B, B, dev)
            ref = (float(loss), opt.grad.clone(), lp.clone(),
                   {k: v.clone() for k, v in m.state_dict().items() if "running" in k})
        if not torch.distributed.is_initialized():
            parallel.init_from_env(backend="gloo")
        torch.distributed.barrier()
        parallel.enable_sync_bn(True)
        m = build(dev, cnn, H)
        opt = FlatAdam(m, lr=1e-3)
        parallel.broadcast_params(opt.flat)
        lo, hi = parallel.shard_range(B, rank, world)
        loss, lp = one_step(m, opt, batch, lo, hi, B, dev)
        kernels = ops.rnn_last_kernels()
        parallel.allreduce_grads(opt.grad)
        tot = parallel.allreduce_stats(loss.reshape(1).clone())
        torch.cuda.synchronize()
        ops.check_health()
        parallel.enable_sync_bn(False)
        parallel.set_batch_split(None, None)
        if rank == 0:
            g_ref, g = ref[1].double(), opt.grad.double()
            stats = {k: float((m.state_dict()[k] - v).abs().max()) for k, v in ref[3].items()}
            results["cnn" if cnn else "rnn"] = dict(
                loss_ref=ref[0], loss_dp=float(tot[0]), loss_rel=abs(float(tot[0]) - ref[0]) / abs(ref[0]),
                grad_rel_l2=float((g - g_ref).norm() / g_ref.norm()), grad_norm=float(g_ref.norm()),
                lp_shard_maxabs=float((lp - ref[2][:, lo:hi]).abs().max()), running_stats_maxabs=max(stats.values()),
                kernels=list(kernels), precision=ops.get_precision(), H=H)
        torch.distributed.barrier()
    if rank == 0:
        json.dump(results, open(out_path, "w"))


def overlap(out_path):
    """Early (overlapped) all-reduce of a layer's gradient slice under 2 ranks.  `even`: 16 + 16 utterances, every rank takes the same
    decision and the top layer's slice is reduced behind its weight GEMMs; `uneven`: 17 + 16 with the side-stream threshold placed
    BETWEEN the two shard sizes -- rank 0 alone would issue the early collective: parallel.overlap_is_rank_invariant() must
    keep both ranks on the step-end all-reduce.  Both must equal the single-process gradient."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ops.set_precision(0)
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    results = {}
    T, H = 48, 32
    for name, B, min_items in (("even", 32, 1), ("uneven", 33, T * 16 * H + 1)):
        batch = synth.make_batch(seed=9, B=B, T=T, F=40, V=30, lab_lo=3, lab_hi=6, full_length=True)
        ops.set_side_stream(True, min_items=1 << 30)        # reference: everything inline
        parallel.enable_overlap(False)
        ref = None
        if rank == 0:
            m = build(dev, False)
            opt = FlatAdam(m, lr=1e-3)
            loss, _ = one_step(m, opt, batch, 0, B, B, dev)
            ref = (float(loss), opt.grad.clone())
        if not torch.distributed.is_initialized():
            parallel.init_from_env(backend="gloo")
        torch.distributed.barrier()
        parallel.enable_sync_bn(True)
        ops.set_side_stream(True, min_items=min_items)
        parallel.enable_overlap(True)
        m = build(dev, False)
        opt = FlatAdam(m, lr=1e-3)
        parallel.broadcast_params(opt.flat)
        lo, hi = parallel.shard_range(B, rank, world)
        loss, _ = one_step(m, opt, batch, lo, hi, B, dev)
        early = len(parallel._overlap["done"])
        parallel.allreduce_grads(opt.grad)
        tot = parallel.allreduce_stats(loss.reshape(1).clone())
        early_all = parallel.allreduce_stats(torch.tensor([float(early)], dtype=torch.float64, device=dev))
        torch.cuda.synchronize()
        ops.check_health()
        parallel.enable_sync_bn(False)
        parallel.enable_overlap(False)
        parallel.set_batch_split(None, None)
        if rank == 0:
            g_ref, g = ref[1].double(), opt.grad.double()
            results[name] = dict(loss_rel=abs(float(tot[0]) - ref[0]) / abs(ref[0]), grad_rel_l2=float((g - g_ref).norm() / g_ref.norm()),
                                 early_slices_rank0=early, early_slices_all_ranks=float(early_all[0]))
        torch.distributed.barrier()
    if rank == 0:
        json.dump(results, open(out_path, "w"))


def main_mode(corpus, out_path):
    from ctc_pytorch_amd.steps import train_ctc as TR
    rank = int(os.environ.get("RANK", "0"))
    d = corpus
    conf = dict(vocab_file=d + "/vocab", train_scp_path=d + "/feats.scp", train_lab_path=d + "/text", valid_scp_path=d + "/feats.scp",
                valid_lab_path=d + "/text", left_ctx=0, right_ctx=0, n_skip_frame=1, n_downsample=1, batch_size=8, shuffle_train=True,
                num_workers=0, rnn_input_size=40, rnn_hidden_size=32, rnn_layers=2, rnn_type="nn.LSTM", bidirectional=True,
                batch_norm=True, add_cnn=False, layers=2, channel="[(1,32),(32,32)]", kernel_size="[(3,3),(3,3)]", stride="[(1,2),(2,2)]",
                padding="[(1,1),(1,1)]", pooling="None", activation_function="relu", drop_out=0.0, init_lr=1e-2, weight_decay=0.0,
                end_adjust_acc=2.0, lr_decay=0.5, num_epoches=3, verbose_step=100, seed=1, sync_bn=True,
                checkpoint_dir=d + "/ckpt_w%s_r%d" % (os.environ.get("WORLD_SIZE", "1"), rank), exp_name="toy")   # per rank: only rank 0 may write
    ops.set_precision(0)
    lines = []
    model, hist = TR.main(conf, log=lines.append)
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).double()
    json.dump(dict(hist=hist, param_sum=float(flat.sum()), param_norm=float(flat.norm()), lines=len(lines),
                   ckpt=os.path.exists(os.path.join(conf["checkpoint_dir"], "toy", "ctc_best_model.pkl"))),
              open("%s.rank%d" % (out_path, rank), "w"))


if __name__ == "__main__":
    if sys.argv[1] == "equiv":
        equiv(sys.argv[2])
    elif sys.argv[1] == "overlap":
        overlap(sys.argv[2])
    else:
        main_mode(sys.argv[2], sys.argv[3])
