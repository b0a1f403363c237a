"""Flat-arena runtime for the optimizer side of the step.

All trainable parameters of the model are re-pointed into ONE contiguous fp32 device buffer
(`params`), with matching `grads`, Adam `exp_avg` and `exp_avg_sq` arenas.  Layout:

  * live parameters first, grouped into SEGMENTS in the order in which the backward pass finishes
    them (DepthDecoder -> BEV heads -> LayoutEncoder -> pose networks -> DepthEncoder layer4 -> rest); a segment is
    one contiguous slice, so the data-parallel hook can all-reduce it the moment the tape reports it
    final (`ops.grad_ready`) while the rest of the backward still runs;
  * parameters that never receive a gradient in the reference (ResNet `fc`, CCT `res_conv`, the
    head the config's `type` does not train) at the tail, so the reduced / updated
    range is a single prefix.

Consequences: gradient zeroing is one memset, the global norm is one (deterministic, two-stage)
reduction, clip+Adam is one kernel, and the all-reduce works on slices of one buffer instead of
466 tensors (the reference flattens them every step, mono/core/utils/dist_utils.py:17-31).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ._lib import call, lib as _jplib

_NO_GRAD_SUFFIXES = (".fc.weight", ".fc.bias", ".res_conv.weight", ".res_conv.bias")

# gradient-ready order of the top-level modules (= reverse order of the backward pass, jperceiver_amd/model/net.py)
# DepthEncoder is the LAST thing the backward finishes (main stream): its layer4 (8.4 M of its 11.2 M parameters) is
# reported separately so that only the small remainder's all-reduce is exposed after the final backward kernel.
SEGMENT_ORDER = ("DepthDecoder", "heads", "LayoutEncoder", "Pose", "DepthEncoder.l4", "DepthEncoder.lo")
_HEAD_PREFIXES = ("CycledViewProjection", "CrossViewTransformer", "LayoutDecoder", "LayoutTransformDecoder")


def segment_of(name: str) -> str:
    top = name.split(".")[0]
    if top.startswith(_HEAD_PREFIXES):
        return "heads"
    if top in ("PoseEncoder", "PoseDecoder"):
        return "Pose"
    if top == "DepthEncoder":
        return "DepthEncoder.l4" if ".layer4." in name else "DepthEncoder.lo"
    return top if top in SEGMENT_ORDER else "other"


def _align(n, a=64):
    return (n + a - 1) // a * a


class FlatArena:
    def __init__(self, model: nn.Module, skip_prefixes=()):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        live, dead = [], []
        for n, p in named:
            is_dead = n.endswith(_NO_GRAD_SUFFIXES) or any(n.startswith(s) for s in skip_prefixes)
            (dead if is_dead else live).append((n, p))
        order = {s: i for i, s in enumerate(SEGMENT_ORDER + ("other",))}
        live.sort(key=lambda np_: order[segment_of(np_[0])])          # stable: named_parameters order inside a segment
        self.entries = []          # (name, param, offset, numel)
        self.segments = {}         # segment -> (offset, length) in floats; contiguous, 256-B aligned
        off = 0
        for n, p in live:
            seg = segment_of(n)
            if seg not in self.segments:
                self.segments[seg] = [off, 0]
            self.entries.append((n, p, off, p.numel()))
            off += _align(p.numel())   # 256-B aligned slices: float4 kernels and RCCL-friendly
            self.segments[seg][1] = off - self.segments[seg][0]
        self.segments = {k: tuple(v) for k, v in self.segments.items()}
        self.n_live_entries = len(live)
        self.live_numel = off
        for n, p in dead:
            self.entries.append((n, p, off, p.numel()))
            off += _align(p.numel())
        self.total = off
        dev = named[0][1].device
        self.params = torch.zeros(self.total, device=dev, dtype=torch.float32)
        self.grads = torch.zeros(self.total, device=dev, dtype=torch.float32)
        for n, p, o, k in self.entries:
            self.params[o:o + k].copy_(p.data.reshape(-1))      # one-time setup (plumbing)
            p.data = self.params[o:o + k].view(p.shape)
            p.grad = self.grads[o:o + k].view(p.shape)
        self.exp_avg = None
        self.exp_avg_sq = None
        self.normsq = torch.zeros(1, device=dev, dtype=torch.float64)
        self._nblk = int(_jplib().fn["jp_sumsq_blocks"]())
        self._partials = torch.zeros(64 * self._nblk, device=dev, dtype=torch.float64)   # grown on demand (small buckets)
        self._n_partials = 0
        self.step_count = 0
        self.dev_state = None      # 3-float device tensor while a captured step owns the optimizer scalars

    def zero_grad(self):
        # memset node on the current stream; keeps p.grad views alive (no reallocation)
        call("jp_fill", self.grads, self.live_numel, 0.0)
        self._n_partials = 0

    # ---- global norm, two-phase so that buckets can be folded in as their all-reduce lands
    def add_norm_partial(self, off: int, n: int):
        """Stage 1 over grads[off:off+n] (a reduced bucket): block partials into the next slot."""
        if (self._n_partials + 1) * self._nblk > self._partials.numel():
            # any bucket_size_mb is legal (the reference accepts any): grow the partial-sum table, keeping what landed
            grown = torch.zeros(2 * self._partials.numel(), device=self._partials.device, dtype=torch.float64)
            grown[:self._partials.numel()].copy_(self._partials)
            self._partials = grown
        slot = self._partials[self._n_partials * self._nblk:(self._n_partials + 1) * self._nblk]
        call("jp_grad_sumsq_partials", self.grads[off:off + n], slot, n)
        self._n_partials += 1

    def grad_norm_sq(self):
        """Stage 2: squared global L2 norm of the live gradients -> self.normsq (device double)."""
        if self._n_partials == 0:
            self.add_norm_partial(0, self.live_numel)
        call("jp_sum_doubles", self._partials, self.normsq, self._n_partials * self._nblk)
        self._n_partials = 0
        return self.normsq

    def adam_step(self, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, max_norm=None, grad_scale=1.0):
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.params)
            self.exp_avg_sq = torch.zeros_like(self.params)
        self.step_count += 1
        normsq = None
        if max_norm is not None and max_norm > 0:
            normsq = self.grad_norm_sq()
        self._n_partials = 0
        if self.dev_state is not None:
            # captured step (apis/trainer.py CapturedStep): lr and the bias corrections come from device memory, refreshed by
            # the host before every replay (`step_state`) -- nothing step-dependent is a kernel argument
            call("jp_adam_clip_step_dev", self.params, self.grads, self.exp_avg, self.exp_avg_sq, self.live_numel, normsq,
                 float(grad_scale), float(max_norm or 0.0), float(betas[0]), float(betas[1]), float(eps), self.dev_state)
            return
        call("jp_adam_clip_step", self.params, self.grads, self.exp_avg, self.exp_avg_sq, self.live_numel, normsq,
             float(grad_scale), float(max_norm or 0.0), float(lr), float(betas[0]), float(betas[1]), float(eps),
             self.step_count)

    @staticmethod
    def step_state(lr, betas, step):
        """[lr, 1 - beta1^step, 1 - beta2^step] as fp32: the scalars jp_adam_clip_step derives on the host for step `step`"""
        import numpy as np
        return np.array([lr, 1.0 - betas[0] ** step, 1.0 - betas[1] ** step], dtype=np.float64).astype(np.float32)

    def layout(self):
        return [(n, o, k) for n, _, o, k in self.entries]


class FlatAdam:
    """torch.optim.Adam-shaped facade over the arena (what build_optimizer returns, trainer.py:76-143).
    `state_dict()` / `load_state_dict()` speak torch.optim.Adam's own format — {'state': {i: {'step', 'exp_avg',
    'exp_avg_sq'}}, 'param_groups': [{..., 'params': [0..n-1]}]} with i indexing `model.parameters()` in order, which
    is how the reference builds its optimizer (trainer.py:99-100) — so an mmcv checkpoint's 'optimizer' entry written by
    the reference resumes here and vice versa.  Parameters that never get a gradient carry no state there either."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, skip_prefixes=()):
        if weight_decay:
            raise NotImplementedError("the north-star configs use weight_decay=0")
        self.arena = FlatArena(model, skip_prefixes)
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False)
        self._model_order = [n for n, p in model.named_parameters() if p.requires_grad]
        self.param_groups = [dict(params=[p for _, p, _, _ in self.arena.entries], **self.defaults)]
        self.max_norm = None          # set by DistOptimizerHook (grad_clip)
        self.grad_scale = 1.0         # 1/world_size folded into the Adam pass after an all-reduce(SUM)

    def zero_grad(self):
        self.arena.zero_grad()

    def step(self):
        g = self.param_groups[0]
        self.arena.adam_step(g["lr"], g["betas"], g["eps"], self.max_norm, self.grad_scale)
        from . import ops
        ops.weights_changed()        # the kernel rewrote parameter memory: packed conv weights are stale

    def state_dict(self):
        a = self.arena
        idx = {n: i for i, n in enumerate(self._model_order)}
        state = {}
        if a.exp_avg is not None:
            for n, p, o, k in a.entries[:a.n_live_entries]:
                state[idx[n]] = dict(step=torch.tensor(float(a.step_count)),
                                     exp_avg=a.exp_avg[o:o + k].view(p.shape).clone(),
                                     exp_avg_sq=a.exp_avg_sq[o:o + k].view(p.shape).clone())
        g = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        g["params"] = list(range(len(self._model_order)))
        return dict(state=state, param_groups=[g],
                    jp_arena=dict(step=a.step_count, layout=a.layout(), total=a.total, live_numel=a.live_numel))

    def load_state_dict(self, sd):
        a = self.arena
        groups = sd.get("param_groups") or [{}]
        if len(groups) != 1:
            raise ValueError("FlatAdam holds one parameter group (the north-star configs have no paramwise_options)")
        n_saved = len(groups[0].get("params", self._model_order))
        if n_saved != len(self._model_order):
            raise ValueError(f"optimizer state covers {n_saved} parameters, the model has {len(self._model_order)}")
        extra = sd.get("jp_arena")
        if extra is not None and [tuple(x) for x in extra["layout"]] != a.layout():
            raise ValueError("flat-arena layout of the checkpoint differs from this model's (different `type` / "
                             "parameter set?): refusing to mis-assign Adam moments")
        for k in ("lr", "betas", "eps"):
            if k in groups[0]:
                self.param_groups[0][k] = tuple(groups[0][k]) if k == "betas" else groups[0][k]
        state = sd.get("state", {})
        if a.exp_avg is None:
            a.exp_avg = torch.zeros_like(a.params)
            a.exp_avg_sq = torch.zeros_like(a.params)
        else:
            a.exp_avg.zero_()
            a.exp_avg_sq.zero_()
        where = {n: (o, k) for n, _, o, k in a.entries}
        live = {n for n, _, _, _ in a.entries[:a.n_live_entries]}
        steps = set()
        for i, st in state.items():
            n = self._model_order[int(i)]
            o, k = where[n]
            if st["exp_avg"].numel() != k:
                raise ValueError(f"Adam state of {n}: {st['exp_avg'].numel()} elements, parameter has {k}")
            if n not in live:
                if float(st["exp_avg_sq"].abs().max()) != 0.0:
                    raise ValueError(f"checkpoint holds Adam state for {n}, which this configuration never trains")
                continue
            a.exp_avg[o:o + k].copy_(st["exp_avg"].reshape(-1))
            a.exp_avg_sq[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): not a single-group Adam run")
        a.step_count = steps.pop() if steps else int(extra["step"]) if extra else 0
