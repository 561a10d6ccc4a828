"""Flat parameter / gradient storage and fused Adam for the HIP path.

replaces: torch.optim.Adam(model.parameters(), lr, weight_decay) as used at the reference's
timit/steps/train_ctc.py:145,62-65 (L2-coupled weight decay, betas (0.9,0.999), eps 1e-8).

All parameters of the model are re-homed into ONE contiguous float32 buffer and all gradients into a second
one (288 GB of HBM: no reason to scatter 25 tensors).  The backward kernels of ops.py accumulate weight
gradients straight into views of the flat gradient buffer (`param._ctcn_grad`), so a training step needs
  1 memset (zero_grad) + 1 RCCL all-reduce over the flat gradient (data parallel) + 1 fused Adam launch.
BatchNorm running statistics stay ordinary buffers.
"""
import torch

from . import ops


def placement_order(names):
    """Indices of `names` (model.named_parameters() order) in flat-buffer placement order: unchanged, except that inside each
    module the recurrent weights come as weight_ih_l0, weight_ih_l0_reverse, weight_hh_l0, weight_hh_l0_reverse."""
    rank = {"weight_ih_l0": 0, "weight_ih_l0_reverse": 1, "weight_hh_l0": 2, "weight_hh_l0_reverse": 3}
    first = {}
    for i, n in enumerate(names):
        first.setdefault(n.rsplit(".", 1)[0], i)
    return sorted(range(len(names)), key=lambda i: (first[names[i].rsplit(".", 1)[0]], rank.get(names[i].rsplit(".", 1)[-1], 4), i))


class FlatAdam:
    """Adam over the flattened parameters of `model`; same public surface as torch.optim.Optimizer where the
    reference touches it: zero_grad(), step(), state_dict(), load_state_dict(), param_groups[i]['lr']."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.model = model
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError("no parameters")
        # Placement order in the flat buffers: model order, except that the input-projection weights of the two directions of a
        # recurrent layer sit next to each other (weight_ih_l0, weight_ih_l0_reverse, then the two weight_hh): ctcn_rnn_fwd /
        # ctcn_rnn_bwd then see [W_ih_fwd ; W_ih_rev] as ONE (2*G*H, I) matrix and run the input projection / dx as a single
        # product without stacking copies.  `self.params` keeps the model's own order.
        placed = placement_order([n for n, _ in named])
        self.layout = [named[i][0] for i in placed]
        params = [named[i][1] for i in placed]
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam: move the model to the ROCm device first (no CPU path)")
        self.params = [p for _, p in named]
        sizes = [p.numel() for p in params]
        total = sum(sizes)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p, n in zip(params, sizes):
                self.flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat[off:off + n].view(p.shape)
                g = self.grad[off:off + n].view(p.shape)
                p._ctcn_grad = g          # ops.py backward kernels accumulate here and return None to autograd
                p.grad = g
                off += n
        self.step_count = 0
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, params=self.params)]

    def zero_grad(self, set_to_none=False):
        ops.join_side_stream()            # no-op unless weight gradients are still in flight on the side stream
        self.grad.zero_()

    def step(self):
        ops.join_side_stream()
        g = self.param_groups[0]
        self.step_count += 1
        ops.adam_step(self.flat, self.grad, self.m, self.v, g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"],
                      self.step_count)

    # ---- checkpoint surface: the torch.optim.Adam layout, so that 'optim_dict' of a package written by either side loads in
    # the other (train_ctc.py:198,223,246 snapshot / roll back / save optimizer.state_dict()) ---------------------------------
    def _slices(self):
        """(index in model.parameters() order, offset, numel, shape) of every parameter inside the flat buffers."""
        order = {id(p): i for i, p in enumerate(self.params)}
        by_name = dict((n, p) for n, p in self.model.named_parameters() if p.requires_grad)
        out, off = [], 0
        for name in self.layout:
            p = by_name[name]
            out.append((order[id(p)], off, p.numel(), tuple(p.shape)))
            off += p.numel()
        return out

    def state_dict(self):
        """{'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]} exactly as torch.optim.Adam writes it
        (parameter i = i-th of model.parameters()); the moments are per-parameter copies of the flat buffers."""
        g = self.param_groups[0]
        template = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"],
                                    weight_decay=g["weight_decay"]).state_dict()["param_groups"][0]      # key set of this torch version
        group = dict(template, params=list(range(len(self.params))))
        state = {}
        if self.step_count > 0:
            for i, off, n, shape in self._slices():
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.m[off:off + n].view(shape).clone(),
                            "exp_avg_sq": self.v[off:off + n].view(shape).clone()}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        if "state" not in sd and "m" in sd:            # round-1 packages: flat moment buffers
            if "layout" in sd and list(sd["layout"]) != list(self.layout):
                raise ValueError("FlatAdam.load_state_dict: the moment buffers were saved with a different parameter placement")
            self.step_count = int(sd["step"])
            self.m.copy_(sd["m"])
            self.v.copy_(sd["v"])
        else:
            groups = sd["param_groups"]
            if len(groups) != 1 or len(groups[0]["params"]) != len(self.params):
                raise ValueError("FlatAdam.load_state_dict: expected one parameter group over %d parameters" % len(self.params))
            state, steps = sd["state"], set()
            self.m.zero_()
            self.v.zero_()
            for i, off, n, shape in self._slices():
                st = state.get(i, state.get(str(i)))
                if st is None:
                    continue
                if tuple(st["exp_avg"].shape) != shape:
                    raise ValueError("FlatAdam.load_state_dict: parameter %d has shape %s, the checkpoint %s" % (i, shape, tuple(st["exp_avg"].shape)))
                self.m[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
            if len(steps) > 1:
                raise ValueError("FlatAdam.load_state_dict: per-parameter step counts differ (%s); one fused step serves all" % sorted(steps))
            self.step_count = steps.pop() if steps else 0
        for k, v in sd["param_groups"][0].items():
            if k in ("lr", "betas", "eps", "weight_decay"):
                self.param_groups[0][k] = v
