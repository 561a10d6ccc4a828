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
        if backend is None:      # CTCN_DIST_BACKEND=gloo: several ranks on ONE GPU (tests; RCCL refuses two ranks per device)
            backend = os.environ.get("CTCN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
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


# Utterances of the current minibatch: (global, in this rank's shard).  Every shard is padded to the GLOBAL T_max, so a
# BatchNorm layer's global element count is local_count * B_global / B_local -- known on the host without a collective
# (and without a device sync), also when the shards are uneven.  None: equal shards (count = local_count * world).
_batch = {"global": None, "local": None}


def set_batch_split(global_b=None, local_b=None):
    """Tell the synchronised BatchNorm how the current minibatch is split (run_epoch calls it per step)."""
    _batch["global"], _batch["local"] = global_b, local_b


def _sync_bn_reduce(sums, local_count):
    """All-reduce of the (C, 2) float64 per-channel BatchNorm sums; returns the global element count per channel."""
    if _collectives_on():
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    if _batch["global"] is not None and _batch["local"]:
        q, r = divmod(int(local_count) * int(_batch["global"]), int(_batch["local"]))
        if r != 0:
            raise RuntimeError("sync BatchNorm: %d elements per channel is not a multiple of the %d local utterances" % (local_count, _batch["local"]))
        return float(q)
    return float(local_count) * world_size()


class ShardedBatches(object):
    """Data-parallel view of a loader of GLOBAL minibatches: every rank iterates the same loader (same seed, same
    shuffle), and takes the contiguous shard [lo, hi) = shard_range(B_global, rank, world) of each collated batch.  The
    collate (utils/data_loader.create_input, reference data_loader.py:119-140) has already padded the batch to the GLOBAL
    T_max / L_max and expressed the lengths as fractions of the global T_max, so slicing rows keeps both: N-GPU math equals
    single-process math on the same global batch (with sync BatchNorm).  Yields (inputs, input_sizes, targets, target_sizes,
    utt_list, B_global).  A global batch with fewer utterances than ranks is dropped on every rank (logged once)."""

    def __init__(self, loader, rank, world, log=print):
        self.loader, self.rank, self.world, self.log = loader, rank, world, log
        self._warned = False

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for inputs, input_sizes, targets, target_sizes, utt_list in self.loader:
            n = int(inputs.shape[0])
            if n < self.world:
                if not self._warned and self.rank == 0:
                    self.log("ShardedBatches: dropping a minibatch of %d utterances (< %d ranks)" % (n, self.world))
                self._warned = True
                continue
            lo, hi = shard_range(n, self.rank, self.world)
            yield inputs[lo:hi], input_sizes[lo:hi], targets[lo:hi], target_sizes[lo:hi], utt_list[lo:hi], n


def allreduce_stats(t):
    """SUM over ranks of a small statistics tensor (loss / error / token counts of a step), in place."""
    if _collectives_on():
        if dist.get_backend() == "nccl" and not t.is_cuda:
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


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
