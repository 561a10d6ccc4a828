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


def dp_safe():
    """CTCN_DP_SAFE=1: the conservative data-parallel arrangement -- no early per-layer slice all-reduce, no weight-gradient side stream, no
    pipelined input projection: every kernel of a step on ONE stream and the ONE gradient all-reduce at the step's end, when nothing of this
    process is resident next to RCCL's kernels.  What the driver's first N > 1 run can A/B the co-residency contract of DESIGN.md section 6
    against (the persistent recurrences next to RCCL workgroups have never met a second device)."""
    return os.environ.get("CTCN_DP_SAFE", "0") == "1"


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).  The rendezvous and every collective
    are bounded by CTCN_DIST_TIMEOUT_S (default 120 s): a peer that never arrives ends in an exception, not in a hang."""
    import datetime
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
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=float(os.environ.get("CTCN_DIST_TIMEOUT_S", "120"))))
    if dp_safe():
        from . import ops
        ops.set_side_stream(False)
        ops.set_fwd_overlap(False)
    return rank, world, local


class RankMonitor(object):
    """First contact with N > 1 ranks must leave evidence, not a hang.  Every rank reports through the rendezvous
    TCPStore -- which needs no collective and lives in rank 0's process -- what phase it is in (`progress`), that it failed (`fail`) or that
    it is done (`finish`, with its hand-off status word and the recurrence kernels it ran).  Rank 0 runs a daemon thread that watches those
    keys: as soon as ANY rank has failed, or when `deadline_s` passes without every rank finishing, it calls `on_trouble(report)` -- bench.py
    prints its one JSON line with `error`, the per-rank records and the RCCL summary from there -- and ends the process with `exit_code`, no
    matter where the main thread is stuck (a collective waiting for the dead peer, a kernel spinning for a hand-off).  The other ranks wait
    (bounded) for rank 0's `monitor/closed` key before they leave with a non-zero code, so that a launcher which tears the job down at the
    first failed worker does not kill rank 0 before the line is out."""

    def __init__(self, rank, world, store=None, deadline_s=None, poll_s=0.5, on_trouble=None, exit_code=3):
        import threading
        import time
        self.rank, self.world, self.on_trouble, self.exit_code = rank, world, on_trouble, exit_code
        self.store = store if store is not None else (dist.distributed_c10d._get_default_store() if dist.is_initialized() else None)
        self.deadline = None if deadline_s is None else time.time() + float(deadline_s)
        self.poll_s, self._stop, self._fired = poll_s, threading.Event(), threading.Event()
        self._thread = None
        if self.store is not None and rank == 0:
            self._thread = threading.Thread(target=self._watch, name="ctcn-rank-monitor", daemon=True)
            self._thread.start()

    # -- every rank ---------------------------------------------------------------------------------------------------------------
    def _set(self, key, obj):
        import json
        if self.store is not None:
            self.store.set("monitor/%s/%d" % (key, self.rank), json.dumps(obj))

    def progress(self, phase, **extra):
        self._set("progress", dict(phase=phase, **extra))

    def finish(self, **record):
        self._set("done", record)

    def fail(self, exc, wait_s=60.0, **record):
        """Report a failure of THIS rank; on ranks other than 0, wait (bounded) until rank 0 has printed."""
        import time
        import traceback
        self._set("failed", dict(error=repr(exc), where=traceback.format_exc(limit=6)[-1200:], **record))
        if self.store is not None and self.rank != 0:
            t_end = time.time() + wait_s
            while time.time() < t_end:
                try:
                    if self.store.check(["monitor/closed"]):
                        break
                except Exception:       # noqa: BLE001 -- rank 0 is gone: nothing left to wait for
                    break
                time.sleep(0.2)

    def close(self):
        """Rank 0, after its line is out (or on the clean path): stop watching and release the waiting ranks."""
        self._stop.set()
        if self.store is not None and self.rank == 0:
            try:
                self.store.set("monitor/closed", "1")
            except Exception:           # noqa: BLE001
                pass

    # -- rank 0 -------------------------------------------------------------------------------------------------------------------
    def collect(self):
        """What every rank has reported so far: {rank: {progress, failed, done}}."""
        import json
        out = {}
        for r in range(self.world):
            rec = {}
            for key in ("progress", "failed", "done"):
                k = "monitor/%s/%d" % (key, r)
                try:
                    if self.store.check([k]):
                        rec[key] = json.loads(self.store.get(k).decode())
                except Exception as e:  # noqa: BLE001
                    rec[key] = {"unreadable": repr(e)}
            out[str(r)] = rec
        return out

    def _watch(self):
        import os as _os
        import sys
        import time
        while not self._stop.wait(self.poll_s):
            try:
                ranks = self.collect()
            except Exception:           # noqa: BLE001
                continue
            failed = [r for r, rec in ranks.items() if "failed" in rec]
            late = self.deadline is not None and time.time() > self.deadline and any("done" not in rec for rec in ranks.values())
            if not failed and not late:
                continue
            if self._stop.is_set():
                return
            self._fired.set()
            why = ("rank %s failed: %s" % (failed[0], ranks[failed[0]]["failed"].get("error")) if failed else
                   "deadline passed; unfinished ranks: %s" % [r for r, rec in ranks.items() if "done" not in rec])
            try:
                if self.on_trouble is not None:
                    self.on_trouble(dict(error=why, ranks=ranks))
            finally:
                sys.stdout.flush()
                self.close()
                time.sleep(0.5)         # (the waiting ranks poll monitor/closed every 0.2 s)
                _os._exit(self.exit_code)


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


class CtcnComm(object):
    """The RCCL communicator behind the C ABI (ctcn_comm_*): what a host without torch.distributed would bind.  Selected for the
    gradient all-reduce with CTCN_COMM=1; the 128-byte unique id travels over the torch.distributed process group (any backend)."""

    def __init__(self, rank, world):
        import ctypes
        from . import _lib
        self._lib, self._ct = _lib, ctypes
        L = _lib.lib()
        blob = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (ctypes.c_char * 128)()
            _lib.check(L.ctcn_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)), "comm_unique_id")
            blob = torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8).clone()
        if world > 1:
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
            blob = blob.to(dev)
            dist.broadcast(blob, src=0)
            blob = blob.cpu()
        raw = (ctypes.c_char * 128).from_buffer_copy(bytes(blob.numpy().tobytes()))
        self.handle = ctypes.c_void_p()
        _lib.check(L.ctcn_comm_init(ctypes.cast(raw, ctypes.c_void_p), rank, world, ctypes.byref(self.handle)), "comm_init")

    def all_reduce_sum_(self, t):
        """in place, float32, enqueued on the current stream"""
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise TypeError("CtcnComm.all_reduce_sum_: contiguous float32 device tensor expected")
        self._lib.check(self._lib.lib().ctcn_comm_allreduce_sum_f32(self.handle, self._ct.c_void_p(t.data_ptr()), t.numel(), self._lib.stream_ptr()),
                        "comm_allreduce_sum_f32")
        return t

    def destroy(self):
        if self.handle:
            self._lib.check(self._lib.lib().ctcn_comm_destroy(self.handle), "comm_destroy")
            self.handle = None


_ctcn_comm = {"comm": None}


def _comm():
    """The C-ABI communicator when CTCN_COMM=1 (created on first use), else None (torch.distributed carries the collective)."""
    if os.environ.get("CTCN_COMM", "0") != "1":
        return None
    if _ctcn_comm["comm"] is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
        _ctcn_comm["comm"] = CtcnComm(rank, world_size())
    return _ctcn_comm["comm"]


# ---- overlap of the gradient all-reduce with the backward pass -------------------------------------------------------------
# optim.FlatAdam lays the four weight matrices of a recurrent layer out adjacently, and ops.py produces their gradients on the
# weight-gradient side stream while the recurrence of the layer below runs.  With the overlap enabled the all-reduce of such a slice
# (9.8 MB per 4x320 layer) is issued right behind those GEMMs -- on RCCL's own stream -- and the step-end call only waits for
# them and reduces what is left (BatchNorm / fc / bottom-layer gradients).  One backward pass per optimiser step is assumed
# (gradient accumulation over several backward passes would reduce a slice twice): opt in with enable_overlap().
_overlap = {"works": [], "done": [], "events": []}


def enable_overlap(flag=True):
    from . import ops
    ops.set_grad_ready_hook(_slice_ready if (flag and not dp_safe()) else None)      # (CTCN_DP_SAFE=1: one all-reduce at the step's end, nothing early)
    _overlap["works"], _overlap["done"], _overlap["events"] = [], [], []


def overlap_is_rank_invariant():
    """The early all-reduce of a layer's gradient slice is a COLLECTIVE: every rank must take the same decision for the same layer, or
    the RCCL call sequences diverge (hang / silent corruption).  ops.py decides per layer from (T, B_local, H, dirs): T, H and dirs are
    global, so the decision is rank-invariant exactly when every rank holds the same number of utterances.  With uneven shards
    (B_global % world != 0, e.g. the last minibatch of an epoch) no slice is reduced early: everything goes with the step-end call."""
    g = _batch["global"]
    return g is None or g % world_size() == 0


def _slice_ready(tensors):
    if not _collectives_on() or not tensors or not overlap_is_rank_invariant():
        return
    ts = sorted(tensors, key=lambda t: t.data_ptr())
    lo, n = ts[0].data_ptr(), sum(t.numel() for t in ts)
    if any(not t.is_contiguous() or t.dtype != torch.float32 for t in ts) or ts[-1].data_ptr() + ts[-1].numel() * 4 - lo != n * 4:
        return                                   # not one contiguous slice of the flat buffer: left to the step-end all-reduce
    view = torch.as_strided(ts[0], (n,), (1,))
    c = _comm()
    if c is not None:                            # C-ABI communicator: in stream order on the side stream itself
        c.all_reduce_sum_(view)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(view.device))
        _overlap["events"].append(ev)
    else:
        _overlap["works"].append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))
    _overlap["done"].append((lo, lo + n * 4))


def allreduce_grads(flat_grad):
    """SUM all-reduce of the flat gradient buffer (loss is already divided by the GLOBAL batch size); with enable_overlap() the
    slices reduced during the backward pass are only waited for and the remainder is reduced here."""
    if flat_grad.is_cuda:
        from . import ops
        ops.join_side_stream()            # weight gradients issued on the side stream must have landed
    if not _collectives_on():
        return flat_grad
    c = _comm() if flat_grad.is_cuda else None
    reduce = (lambda t: c.all_reduce_sum_(t)) if c is not None else (lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM))
    done = sorted(_overlap["done"])
    for w in _overlap["works"]:
        w.wait()                          # the current stream waits for the collective
    for ev in _overlap["events"]:
        torch.cuda.current_stream(flat_grad.device).wait_event(ev)
    _overlap["works"], _overlap["done"], _overlap["events"] = [], [], []
    if not done:
        reduce(flat_grad)
        return flat_grad
    base, end = flat_grad.data_ptr(), flat_grad.data_ptr() + flat_grad.numel() * 4
    cur = base
    for lo, hi in done + [(end, end)]:
        if lo > cur:
            reduce(flat_grad[(cur - base) // 4:(lo - base) // 4])
        cur = max(cur, hi)
    return flat_grad


# Utterances of the current minibatch: (global, in this rank's shard).  Every shard is padded to the GLOBAL T_max, so a
# BatchNorm layer's global element count is local_count * B_global / B_local -- known on the host without a collective
# (and without a device sync), also when the shards are uneven.  None: equal shards (count = local_count * world).
_batch = {"global": None, "local": None, "seen": 0}


def set_batch_split(global_b=None, local_b=None, count=True):
    """Tell the synchronised BatchNorm how the current minibatch is split (run_epoch calls it per step); the utterances this rank has TRAINED on
    since the last sync_bn_buffers are counted on the way (its pooling weights).  `count=False` for a validation pass: the running statistics
    do not move there, so its utterances must not weigh in (from epoch 2 on the weights were train + previous dev utterances)."""
    _batch["global"], _batch["local"] = global_b, local_b
    if local_b and count:
        _batch["seen"] = _batch.get("seen", 0) + int(local_b)


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


def sync_bn_buffers(model, weight=None):
    """Per-shard BatchNorm (the default, sync_bn off) leaves every rank with running_mean / running_var of its OWN shards; evaluation and
    the checkpoint (written by rank 0) would then depend on the rank.  Called at the end of a training epoch: the buffers of all ranks are
    merged the way two populations are pooled (law of total variance), weighted by the utterances every rank has seen since the last
    merge (`weight`; default: the count set_batch_split accumulated, 1 if none -- equal shards):

        running_mean = sum_r p_r rm_r                     running_var = sum_r p_r rv_r + sum_r p_r (rm_r - running_mean)^2

    (round 4 averaged the variances and dropped the second term).  What this recovers and what it cannot: with per-shard statistics the
    single-process running_var is the moving average of [E_r var_{r,k} + Var_r mu_{r,k}] over the steps k; the buffers hold the moving
    averages of var_{r,k} and of mu_{r,k} SEPARATELY, so the between-shard term is available only as the variance of the AVERAGED means:
    exact for differences between the shards that persist over the steps, while step-to-step sampling noise of the shard means is
    averaged away before it can be squared (the cross-rank product mu_{r,k} mu_{s,k} of one step is not in any rank's buffer).  The
    remainder is of the order of var / (independent frames per shard) and goes to zero as the shards grow; `enable_sync_bn()` is the exact
    mode.  num_batches_tracked is identical everywhere.  A no-op without collectives or with sync_bn on.  Returns the
    number of buffers merged."""
    if not _collectives_on() or world_size() == 1:
        return 0
    from . import ops
    if ops._sync_bn["reduce"] is not None:
        return 0
    mods = [m for m in model.modules() if getattr(m, "running_mean", None) is not None and getattr(m, "running_var", None) is not None]
    if not mods:
        return 0
    # the weights must mean the same thing on every rank: utterance counts where EVERY rank has one (or was given one), else equal weights
    mine = float(weight) if weight is not None else float(_batch.get("seen") or 0)
    _batch["seen"] = 0
    have = torch.tensor([1.0 if mine > 0 else 0.0], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        have = have.cuda()
    dist.all_reduce(have, op=dist.ReduceOp.SUM)
    w = mine if int(have.item()) == world_size() else 1.0
    means = torch.cat([m.running_mean.detach().reshape(-1).double() for m in mods])
    vars_ = torch.cat([m.running_var.detach().reshape(-1).double() for m in mods])
    n = means.numel()
    # one collective: [w, w * mean, w * (var + mean^2)] -- the pooled second moment minus the pooled mean squared is the formula above
    pack = torch.cat([torch.full((1,), w, dtype=torch.float64, device=means.device), w * means, w * (vars_ + means * means)])
    if dist.get_backend() == "nccl" and not pack.is_cuda:
        pack = pack.cuda()
    dist.all_reduce(pack, op=dist.ReduceOp.SUM)
    pack = pack.to(means.device)
    tot = pack[0]
    mean = pack[1:1 + n] / tot
    var = (pack[1 + n:] / tot - mean * mean).clamp_(min=0.0)
    off = 0
    with torch.no_grad():
        for m in mods:
            c = m.running_mean.numel()
            m.running_mean.copy_(mean[off:off + c].view_as(m.running_mean))
            m.running_var.copy_(var[off:off + c].view_as(m.running_var))
            off += c
    return 2 * len(mods)


def broadcast_params(flat_params, src=0):
    if _collectives_on():
        dist.broadcast(flat_params, src=src)
    return flat_params


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if _collectives_on():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value, device):
    """[value of rank 0, value of rank 1, ...] on every rank (one small all-gather); [value] without collectives."""
    if not _collectives_on():
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]
