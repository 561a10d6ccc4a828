"""Utterance-sharded data parallelism: one process per GPU, one RCCL all-reduce of the flat gradient buffer.

The reference is single-device; this is the north-star's DP requirement.  Each rank holds a
contiguous shard of the global minibatch padded to the GLOBAL T_max / L_max, computes
loss_rank = sum_shard nll / B_global, and the flat fp32 gradient (33.4 MB for 4x320 BiLSTM) is summed with a
single torch.distributed all_reduce (backend 'nccl' == RCCL over xGMI; 'gloo' in the CPU tests).
BatchNorm batch statistics are local to the shard unless sync_bn is enabled (documented deviation).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or _FORCE) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


# CTCN_FORCE_COLLECTIVES=1: create the process group and issue every collective even for a single rank, so that the
# RCCL code path (communicator set-up, stream ordering against the side stream) can be exercised on a 1-GPU box.
_FORCE = os.environ.get("CTCN_FORCE_COLLECTIVES", "0") == "1"


def _collectives_on():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for `rank` (first n_items % world ranks get one extra)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_grads(flat_grad):
    """SUM all-reduce of the flat gradient buffer (loss is already divided by the GLOBAL batch size)."""
    if flat_grad.is_cuda:
        from . import ops
        ops.join_side_stream()            # weight gradients issued on the side stream must have landed
    if _collectives_on():
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


def _sync_bn_reduce(sums, local_count):
    """All-reduce of the (C, 2) float64 per-channel BatchNorm sums; returns the global element count per channel.
    Every rank holds the same padded shard shape (global T_max, equal utterances per rank), so the count is
    local_count * world_size and needs no second collective."""
    if _collectives_on():
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    return float(local_count) * world_size()


def enable_sync_bn(flag=True):
    """BatchNorm statistics over the GLOBAL batch: N-GPU math equals single-process math on the same
    global batch, at the price of two tiny (2*C doubles) all-reduces per BatchNorm layer per pass.  Default off:
    per-shard statistics, the fast documented deviation."""
    from . import ops
    ops.set_sync_bn(_sync_bn_reduce if flag else None)


def broadcast_params(flat_params, src=0):
    if _collectives_on():
        dist.broadcast(flat_params, src=src)
    return flat_params


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if _collectives_on():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
